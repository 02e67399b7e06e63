"""Race hunt: the SPLIT-GMVAE fp32 step (B = 4, as tests/test_gpu_gm.py) repeated with the SAME weights (lr = 0) and pinned noise;
every gradient is compared with the first step's.  fp32 atomics reorder sums at the 1e-6 level; anything above 1e-3 of a tensor's
scale is a glitch.  usage: python scripts/stress_gm_race.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_gm import _inputs, _params, BETA, ALPHA, K, TAU, H
from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
from split_vae_amd.optimizer import Adam

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = 4
images, nz = _inputs(B)
model = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype=os.environ.get("GM_DTYPE", "f32"), device="cuda", seed=1)
model.beta, model.alpha = BETA, ALPHA
model.set_weights(_params())
opt = Adam(learning_rate=0.0)
img = torch.from_numpy(images).cuda()
cu = lambda a: torch.from_numpy(a).cuda()
eps = (cu(nz["eps_x"]), cu(nz["eps_h"]))
noise = (cu(nz["u"]), cu(nz["keep1"]), cu(nz["keep5"]))
names = model.keras_names()
# two different sets of draws, alternated: a stale buffer from the previous step then shows (identical steps would hide it)
images2, nz2 = _inputs(B, seed=7)
sets = [(img, eps, noise), (torch.from_numpy(images2).cuda(), (cu(nz2["eps_x"]), cu(nz2["eps_h"])), (cu(nz2["u"]), cu(nz2["keep1"]), cu(nz2["keep5"])))]
refs = [None, None]
bad = {}
for i in range(N):
    im, ep, no = sets[i & 1]
    train_step_lg_gm_vae(model, im, opt, eps=ep, noise=no)
    g = [t.clone() for t in model.gradients]
    if refs[i & 1] is None:
        refs[i & 1] = g
        scales = [float(t.abs().max()) + 1e-12 for t in g]
        continue
    ref = refs[i & 1]
    for n, a, b, s in zip(names, g, ref, scales):
        e = float((a - b).abs().max()) / s
        if e > 1e-3:
            bad.setdefault(n, []).append((i, round(e, 4), int(((a - b).abs() > 1e-3 * s).sum())))
print("steps", N, "glitches:", {k: v[:4] + ([("...", len(v))] if len(v) > 4 else []) for k, v in bad.items()} or "none")
