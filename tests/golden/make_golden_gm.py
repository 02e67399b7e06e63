"""Generates tests/golden/lggmvae_svhn32_b2.npz from the fp64 oracle restatement of SPLIT-GMVAE (oracle/gm_ref.py).

Like make_golden.py these are NOT TensorFlow outputs (TF-2.0 cannot be installed here): they pin the oracle
against regressions and give the GPU test a committed target.  Config 3 of the reference README (:62): SVHN-32,
y_size 30, tau 0.4, beta 40, alpha 40, patch 4.  Weights are regenerated from the seed (6.8 M parameters do not
belong in a fixture); gradients / updated weights are stored as evenly spaced samples plus norms.

Run from the repo root:  python tests/golden/make_golden_gm.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gm_ref, np_ref  # noqa: E402

B, H, PATCH, BETA, ALPHA, K, TAU, SEED_W = 2, 32, 4, 40.0, 40.0, 30, 0.4, 5
NAMES14 = ["x_mean", "x_log_scale", "z_x", "z_mean_x", "z_sig_x", "z_x_hat", "x_hat_mean", "x_hat_log_scale", "z_mean_x_hat",
           "z_sig_x_hat", "y", "y_logits", "z_prior_mean", "z_prior_sig"]


def sample_idx(n, k=48):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def golden_params():
    params = gm_ref.gm_glorot_init(H, H, seed=SEED_W, y_size=K)
    rngb = np.random.Generator(np.random.PCG64(98))
    names = [n for n, _ in gm_ref.gm_param_shapes(H, H, y_size=K)]
    for i, n in enumerate(names):
        if n.endswith("bias"):
            params[i] = (params[i] + rngb.standard_normal(params[i].shape) * 0.05).astype(np.float32)
    return params


def inputs():
    rng = np.random.Generator(np.random.PCG64(4321))
    x = (rng.integers(0, 256, size=(B, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
    perm = np.stack([rng.permutation((H // PATCH) ** 2) for _ in range(B)]).astype(np.int32)
    F_ = (H // 8) ** 2 * 128
    return dict(x=x, perm=perm, eps_x=rng.standard_normal((B, 128)).astype(np.float32),
                eps_h=rng.standard_normal((B, 128)).astype(np.float32), u=rng.uniform(0.02, 0.98, (B, K)).astype(np.float32),
                keep1=(rng.uniform(size=(B, 1024)) > 0.2).astype(np.float32), keep5=(rng.uniform(size=(B, F_)) > 0.2).astype(np.float32))


def main():
    inp = inputs()
    params = golden_params()
    images = np_ref.scramble_batch(inp["x"], inp["perm"], PATCH).astype(np.float32)
    ref = gm_ref.GMRefTrainer(params, BETA, ALPHA, y_size=K, tau=TAU, dtype=torch.float64)
    args = (images, inp["eps_x"], inp["eps_h"], inp["u"], inp["keep1"], inp["keep5"])
    fwd, losses, grads = ref.grads(*args)
    out = dict(inp, images=images, beta=np.float64(BETA), alpha=np.float64(ALPHA), tau=np.float64(TAU), y_size=np.int32(K),
               patch=np.int32(PATCH), weight_seed=np.int32(SEED_W),
               weight_checksum=np.float64(sum(float(np.abs(p.astype(np.float64)).sum()) for p in params)))
    for n, t in zip(NAMES14, fwd):
        out["fwd_" + n] = t.detach().numpy().astype(np.float32 if n.startswith("x_") else np.float64)
    for k, v in losses.items():
        out["loss_" + k] = np.float64(float(v))
    for i, g in enumerate(grads):
        gn = g.numpy().astype(np.float64).ravel()
        out["grad_norm_%02d" % i] = np.float64(np.linalg.norm(gn))
        out["grad_max_%02d" % i] = np.float64(np.abs(gn).max())
        out["grad_samp_%02d" % i] = gn[sample_idx(gn.size)]
    for step in range(1, 3):
        l, _ = ref.train_step(*args)
        out["step%d_total_loss" % step] = np.float64(l["total_loss"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lggmvae_svhn32_b2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
