"""Race hunt 2: the first two training steps of a FRESH SPLIT-GMVAE fp32 model (as tests/test_gpu_gm.py: forward surface call, then
two steps with lr 1e-4, an idle gap before each), repeated; the step-2 gradients of every repetition against the first one's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_gm import _inputs, _params, BETA, ALPHA, K, TAU, H
from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
from split_vae_amd.optimizer import Adam

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gap = float(os.environ.get("GAP", "0.2"))
B = 4
images, nz = _inputs(B)
params = _params()
img = torch.from_numpy(images).cuda()
cu = lambda a: torch.from_numpy(a).cuda()
eps = (cu(nz["eps_x"]), cu(nz["eps_h"]))
noise = (cu(nz["u"]), cu(nz["keep1"]), cu(nz["keep5"]))
ref, bad, wdiff, bad_px = None, [], [], []
for it in range(N):
    model = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="f32", device="cuda", seed=1)
    model.beta, model.alpha = BETA, ALPHA
    model.set_weights(params)
    opt = Adam(learning_rate=1e-4)
    model(img, training=True, eps=eps, noise=noise)
    gs = []
    for t in range(2):
        torch.cuda.synchronize(); time.sleep(gap)
        if t == 1 and os.environ.get("PIN_W"):          # same weights for every repetition's second step
            if it == 0:
                snap = (model.flat.clone(), model.gm_flat.clone())
            else:
                dw = float((model.flat - snap[0]).abs().max()), float((model.gm_flat - snap[1]).abs().max())
                if max(dw) > 0:
                    wdiff.append((it, dw, int((model.flat != snap[0]).sum()), int((model.gm_flat != snap[1]).sum())))
                model.flat.copy_(snap[0]); model.gm_flat.copy_(snap[1])
        train_step_lg_gm_vae(model, img, opt, eps=eps, noise=noise)
        gs.append([g.clone() for g in model.gradients])
        if t == 1:
            pl = model.plan(B)
            probe = (pl.buffer("out6_x", torch.float32, (B, H, H, 6)).clone(), pl.buffer("g5_x", torch.float32, (B, H, H, 8)).clone())
        model.get_weights()
    if ref is None:
        ref = gs
        ref_probe = probe
        names = model.keras_names()
        continue
    d5 = (probe[1] - ref_probe[1]).abs()
    if float(d5.max()) > 1e-3 * float(ref_probe[1].abs().max()) and not bad_px:
        idx = [int(v) for v in torch.nonzero(d5 == d5.max())[0]]
        b_, y_, x_, c_ = idx
        bad_px.append({"rep": it, "pixel": idx, "g5_now": probe[1][b_, y_, x_].tolist(), "g5_ref": ref_probe[1][b_, y_, x_].tolist(),
                       "out6_now": probe[0][b_, y_, x_].tolist(), "out6_ref": ref_probe[0][b_, y_, x_].tolist(),
                       "image_x": img[b_, y_, x_, :3].tolist(), "n_pixels_differing": int((d5.amax(dim=-1) > 1e-3 * float(ref_probe[1].abs().max())).sum())})
    for t in range(2):
        for n, a, b in zip(names, gs[t], ref[t]):
            s = float(b.abs().max()) + 1e-12
            e = float((a - b).abs().max()) / s
            if e > 1e-3:
                bad.append((it, t + 1, n, round(e, 4), int(((a - b).abs() > 1e-3 * s).sum())))
    del model
print("repetitions", N, "glitches:", [b for b in bad if b[2] == "decoder_x/d1/kernel:0"] or "none")
print("weights after step 1 differing from repetition 0:", wdiff[:12], len(wdiff))
print("loss-gradient probe at the first glitch:", bad_px or "none")
