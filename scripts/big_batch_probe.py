"""Large batches: a 2048-image (and 4096-image bf16) gradient evaluation against the mean of its 512-image quarters -- per-image terms bit for bit where the arithmetic is per image,
gradients to accumulation order.  Hunts 32-bit offset overflows: at 2048 images the 64 x 64 x 32-channel fp32 tensors pass 2^31 bytes, at 4096 the element counts pass 2^30."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from split_vae_amd import data, ops
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
H, PATCH, BETA = 64, 8, 120.0

def images(batch, off):
    x = data.synthetic_images(batch, H, H, seed=0, device="cuda", sample_offset=off)
    return Augmentator("scramble", size=PATCH, seed=1).augment(x, sample_offset=off)

def run(plan, P, img, off):
    n = img.shape[0]
    G = torch.zeros_like(P)
    plan.step(PHASE_ALL & ~PHASE_ADAM, params=P, grads=G, images6=img, seed=5, step=3, sample_offset=off, t=1)
    torch.cuda.synchronize()
    per = {k: plan.buffer(k, torch.float32, (n,)).clone() for k in ("nll_x", "nll_xh", "kl_x", "kl_xh")}
    return per, G, plan.buffer("losses", torch.float32, (8,)).clone()

bad = 0
for dt, tdt, BIG in (("f32", torch.float32, 2048), ("bf16", torch.bfloat16, 4096)):
    model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dt, device=torch.device("cuda"), seed=3)
    P = model.flat
    big = ops.LGVaePlan(BIG, H, H, beta=BETA, dtype=tdt)
    pf, Gf, Lf = run(big, P, images(BIG, 0), 0)
    del big
    torch.cuda.empty_cache()
    q = ops.LGVaePlan(512, H, H, beta=BETA, dtype=tdt)
    parts, Gs = [], []
    for i in range(BIG // 512):
        p, G, _ = run(q, P, images(512, 512 * i), 512 * i)
        parts.append(p); Gs.append(G)
    Gm = sum(Gs) / len(Gs)
    for k in pf:
        got = torch.cat([p[k] for p in parts])
        err = float((got - pf[k]).abs().max() / pf[k].abs().max())
        worst = int((got - pf[k]).abs().argmax())
        ok = err <= 2e-4
        bad += not ok
        print("%s B=%d %-7s per-image max rel diff vs quarters %.2e (image %d) %s" % (dt, BIG, k, err, worst, "ok" if ok else "BAD"))
    rel = float((Gm - Gf).norm() / Gf.norm())
    ok = rel <= 2e-3 and bool(torch.isfinite(Lf).all())
    bad += not ok
    print("%s B=%d gradient vs mean of quarters: relative L2 %.2e, losses finite %s %s" % (dt, BIG, rel, bool(torch.isfinite(Lf).all()), "ok" if ok else "BAD"))
sys.exit(1 if bad else 0)
