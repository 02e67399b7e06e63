"""Generates tests/golden/lgspair_b2.npz and lgspair_hard_b2.npz from the fp64 oracle restatement of SPLIT-SPAIR (oracle/spair_model_ref.py).

Like make_golden.py these are NOT TensorFlow outputs (TF-2.0 cannot be installed here): they pin the oracle against regressions
and give the GPU test a committed target.  Config 5 of the reference README (:93): lg_spair -split_z_l -concat_z_what
-dense_local -dense_bg, latent 64 / bg 4 / local 4, patch 8, z_bg_beta 10; 48x48 canvases, batch 2, step 41 of the annealing
schedule -- and (lgspair_hard_b2.npz) BASELINE config 5 as the README names it (:107, Multi-Bird-Hard): latent 64 / bg 64 / local 64,
z_bg_beta 1, z_what_beta 0.5, same switches.  The 31.9 M variables and the random draws are regenerated from their seeds; the fixture stores the inputs, every loss
term, small tensors whole, evenly spaced samples + norms of the large ones and of every variable's gradient.

Run from the repo root:  python tests/golden/make_golden_spair.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spair_model_ref as R  # noqa: E402

B, STEP, SEED_W, SEED_N = 2, 41, 5, 7
CONFIG = dict(model="lg_spair", latent_size=64, bg_latent_size=4, local_latent_size=4, patch_size=8, z_bg_beta=10.0, split_z_l=True,
              concat_z_what=True, dense_local=True, dense_bg=True)                                         # README.md:93 (Multi-Bird-Easy)
CONFIG_HARD = dict(model="lg_spair", latent_size=64, bg_latent_size=64, local_latent_size=64, patch_size=8, z_bg_beta=1.0, z_what_beta=0.5,
                   split_z_l=True, concat_z_what=True, dense_local=True, dense_bg=True)                    # README.md:107 (Multi-Bird-Hard)
FIXTURES = {"lgspair_b2.npz": CONFIG, "lgspair_hard_b2.npz": CONFIG_HARD}
WHOLE = ["z_where", "z_where_mean", "z_where_sigma", "z_depth", "z_pres", "z_pres_logits", "z_pres_pre_sigmoid", "z_bg", "z_l", "obj_bbox_mask"]
SAMPLED = ["x_recon", "x_hat_recon", "z_what", "all_glimpses", "obj_full_recon_unnorm", "obj_recon_alpha"]


def sample_idx(n, k=64):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def images():
    g = torch.Generator().manual_seed(11)
    return torch.rand(B, 48, 48, 6, generator=g)


def compute(config=None):
    cfg = R.default_config(**(config or CONFIG))
    p = R.init_params(cfg, seed=SEED_W)
    noise = R.draw_noise(cfg, B, seed=SEED_N)
    for v in p.values():
        v.requires_grad_(True)
    img = images()
    o = R.forward(p, cfg, img.double(), noise, training=True)
    total, losses = R.losses(cfg, img.double(), o, STEP)
    grads = torch.autograd.grad(total, list(p.values()))
    out = {"images": img.numpy(), "total_loss": np.float64(total.item()), "losses": np.array([float(l) for l in losses])}
    for k in WHOLE:
        out["out/" + k] = o[k].detach().numpy()
    for k in SAMPLED:
        f = o[k].detach().numpy().reshape(-1)
        out["sample/" + k] = f[sample_idx(f.size)]
        out["norm/" + k] = np.float64(np.linalg.norm(f))
    out["grad_norms"] = np.array([float(g.norm()) for g in grads])
    # the same graph evaluated in fp32 on the CPU: how far fp32 rounding alone moves each gradient's norm (the canvas cross-entropy divides by
    # predictions near 1e-8, some sums cancel by 1e4 and more: tests/test_gpu_spair_model.py bounds the device's norms by 3x this, floor 5e-3)
    p32 = {k: v.detach().float().requires_grad_(True) for k, v in p.items()}
    o32 = R.forward(p32, cfg, img, {k: v.float() for k, v in noise.items()}, training=True)
    g32 = torch.autograd.grad(R.losses(cfg, img, o32, STEP)[0], list(p32.values()))
    out["grad_norms_f32"] = np.array([float(g.double().norm()) for g in g32])
    # ... and each gradient's relative L2 distance from the fp64 one (the same noise scale tests/test_gpu_spair_model.py::test_spair_step_matches_oracle uses)
    out["grad_err_f32"] = np.array([float((a.double() - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(g32, grads)])
    out["grad_samples"] = np.stack([g.detach().numpy().reshape(-1)[sample_idx(g.numel(), 8)] if g.numel() >= 8 else
                                    np.resize(g.detach().numpy().reshape(-1), 8) for g in grads])
    return out


if __name__ == "__main__":
    for name, config in FIXTURES.items():
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), **compute(config))
        print("wrote tests/golden/" + name)
