"""Generates tests/golden/lgvae_svhn32_b2.npz from the oracle restatement (fp64).

The reference cannot run here (TensorFlow 2.0 is not installable), so these vectors are
NOT TensorFlow outputs: they are produced by oracle/torch_ref.py in float64 and cross-checked in
this script against the independent NumPy restatement oracle/np_ref.py (forward + losses to
1e-12, sampled gradients against central finite differences).  They pin the oracle against
regressions and give the GPU tests a committed target.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_ref, torch_ref  # noqa: E402

B, H, PATCH, BETA, SEED_W = 2, 32, 4, 40.0, 3
SAMPLE = 64   # sampled entries per tensor for gradients / updated weights


def inputs():
    rng = np.random.Generator(np.random.PCG64(1234))
    x = (rng.integers(0, 256, size=(B, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
    x[0, 0, :4] = -1.0          # edge bins (vae/trainer.py:37)
    x[0, 1, :4] = 1.0
    perm = np.stack([rng.permutation((H // PATCH) ** 2) for _ in range(B)]).astype(np.int32)
    eps = rng.standard_normal((2, B, 128)).astype(np.float32)
    return x, perm, eps


def sample_idx(n, k=SAMPLE):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def main():
    x, perm, eps = inputs()
    params = np_ref.glorot_init(H, H, seed=SEED_W)
    rngb = np.random.Generator(np.random.PCG64(99))
    for i in range(1, len(params), 2):   # non-zero biases
        params[i] = (rngb.standard_normal(params[i].shape) * 0.05).astype(np.float32)
    images = np_ref.scramble_batch(x, perm, PATCH).astype(np.float32)
    assert np.array_equal(images, torch_ref.scramble_batch(x, perm, PATCH).numpy())
    ref = torch_ref.RefTrainer(params, BETA, dtype=torch.float64)
    fwd, losses, grads = ref.grads(torch.from_numpy(images).double(), eps[0], eps[1])
    # cross-check with the independent NumPy restatement
    fwd_np = np_ref.lgvae_forward(images, params, eps[0], eps[1])
    for a, b in zip(fwd_np, fwd):
        assert np.abs(a - b.detach().numpy()).max() < 1e-12
    l_np = np_ref.lgvae_losses(images, fwd_np, BETA)
    for k in l_np:
        assert abs(l_np[k] - float(losses[k])) < 1e-9 * max(1.0, abs(l_np[k])), k
    for which, idx in [(0, 5), (8, 77), (20, 33), (28, 10), (38, 3), (39, 2)]:
        fd = np_ref.fd_grad(images, params, eps[0], eps[1], BETA, which, idx)
        ad = float(grads[which].flatten()[idx])
        assert abs(fd - ad) < 1e-5 * max(1.0, abs(ad)), (which, idx, fd, ad)
    out = dict(x=x, perm=perm, eps=eps, images=images, beta=np.float64(BETA), patch=np.int32(PATCH),
               weight_seed=np.int32(SEED_W),
               weight_checksum=np.float64(sum(float(np.abs(p.astype(np.float64)).sum()) for p in params)))
    names = ["x_mean", "x_log_scale", "z_x", "z_mean_x", "z_sig_x", "z_x_hat", "x_hat_mean", "x_hat_log_scale",
             "z_mean_x_hat", "z_sig_x_hat"]
    for n, t in zip(names, fwd):
        out["fwd_" + n] = t.detach().numpy().astype(np.float64)
    for k, v in losses.items():
        out["loss_" + k] = np.float64(float(v))
    for i, g in enumerate(grads):
        gn = g.numpy().astype(np.float64).ravel()
        out["grad_norm_%02d" % i] = np.float64(np.linalg.norm(gn))
        out["grad_max_%02d" % i] = np.float64(np.abs(gn).max())
        out["grad_samp_%02d" % i] = gn[sample_idx(gn.size)]
    # three Adam steps on the same batch
    for step in range(1, 4):
        l, _ = ref.train_step(torch.from_numpy(images).double(), eps[0], eps[1])
        out["step%d_total_loss" % step] = np.float64(l["total_loss"])
        if step in (1, 3):
            for i, p in enumerate(ref.params):
                pn = p.detach().numpy().ravel()
                out["w%d_samp_%02d" % (step, i)] = pn[sample_idx(pn.size)]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lgvae_svhn32_b2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
