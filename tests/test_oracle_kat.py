"""CPU: pins for the oracle restatement.  The reference ships no tests or vectors for this path
 and TensorFlow cannot run here, so the oracle is pinned by (i) analytic known-answer
tests, (ii) agreement of two independent restatements (NumPy f64 loops vs torch functional +
autograd), (iii) finite differences, (iv) the committed golden fixture."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ref, torch_ref


def test_same_padding_table():
    # TF SAME: k4s1 -> 1/2, k6s1 -> 2/3, k6s2 -> 2/2, k4s2 -> 1/1
    assert np_ref.same_pads(32, 4, 1) == (32, 1, 2)
    assert np_ref.same_pads(32, 6, 1) == (32, 2, 3)
    assert np_ref.same_pads(64, 6, 2) == (32, 2, 2)
    assert np_ref.same_pads(16, 4, 2) == (8, 1, 1)
    # torch's own padding='same' (stride 1) puts the extra pixel on the same side
    x = torch.randn(1, 3, 8, 8, dtype=torch.float64)
    w = torch.randn(5, 3, 4, 4, dtype=torch.float64)
    a = F.conv2d(x, w, padding='same')
    b = torch_ref.conv2d_same(x.permute(0, 2, 3, 1), w.permute(2, 3, 1, 0), torch.zeros(5, dtype=torch.float64), 1)
    assert torch.allclose(a, b.permute(0, 3, 1, 2), atol=1e-12)


def test_bilinear_stencil():
    v = (np.arange(8, dtype=np.float64) ** 2)
    x = np.broadcast_to(v[None, :, None, None], (1, 8, 8, 1)).copy()
    up = np_ref.resize_bilinear_2x(x)[0, :, 0, 0]
    want = [0, .25, .75, 1.75, 3.25, 5.25, 7.75, 10.75, 14.25, 18.25, 22.75, 27.75, 33.25, 39.25, 45.75, 49.0]
    assert np.allclose(up, want, atol=1e-12)
    xt = torch.from_numpy(np.random.default_rng(0).standard_normal((2, 5, 7, 3)))
    assert np.allclose(np_ref.resize_bilinear_2x(xt.numpy()), torch_ref.resize_bilinear_2x(xt).numpy(), atol=1e-12)


@pytest.mark.parametrize("m,ls", [(0.1, -2.0), (-0.7, -5.0), (0.9, -1.0), (0.0, -3.5), (0.3, -6.0)])
def test_discretised_logistic_sums_to_one(m, ls):
    ks = -1 + 2 * np.arange(256) / 255.0
    nll = np_ref.discretised_logistic_loss(ks, np.full(256, m), np.full(256, ls))
    assert abs(np.exp(-nll).sum() - 1.0) < 1e-5


def test_discretised_logistic_branches_match_torch():
    rng = np.random.default_rng(1)
    x = (rng.integers(0, 256, 4000) / 255.0 * 2 - 1)
    x[:50] = -1.0
    x[50:100] = 1.0
    m = rng.standard_normal(4000)
    ls = rng.uniform(-7, 9, 4000)      # sharp to very wide: exercises all four branches
    a = np_ref.discretised_logistic_loss(x, m, ls)
    b = torch_ref.discretised_logistic_loss(torch.from_numpy(x), torch.from_numpy(m), torch.from_numpy(ls)).numpy()
    # torch's softplus switches to the identity above 20 (error <= e^-20 = 2e-9); nothing else differs
    assert np.allclose(a, b, rtol=1e-9, atol=5e-9)


def test_kl_kats():
    z = np.zeros((3, 16)); one = np.ones((3, 16))
    assert abs(np_ref.kl_divergence(z, one)) < 1e-12
    mu = np.random.default_rng(2).standard_normal((3, 16))
    assert abs(np_ref.kl_divergence(mu, one) - 0.5 * np.mean(np.sum(mu ** 2, 1))) < 1e-12
    sg = np.random.default_rng(3).uniform(0.2, 2.0, (3, 16))
    assert abs(np_ref.kl_divergence(z, sg) - np.mean(-0.5 * np.sum(1 + 2 * np.log(sg) - sg ** 2, 1))) < 1e-12
    # KL(concat) = KL_x + KL_xhat ; kl_two_gauss(mu, sig, 0, 1) == kl_divergence(mu, sig)
    both = np_ref.kl_divergence(np.concatenate([mu, z], 1), np.concatenate([sg, one], 1))
    assert abs(both - np_ref.kl_divergence(mu, sg) - np_ref.kl_divergence(z, one)) < 1e-12
    assert abs(np_ref.kl_divergence_two_gauss(mu, sg, 0.0, 1.0) - np_ref.kl_divergence(mu, sg)) < 1e-12
    assert abs(float(torch_ref.kl_divergence(torch.from_numpy(mu), torch.from_numpy(sg))) - np_ref.kl_divergence(mu, sg)) < 1e-12


def test_scramble_properties():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3, 16, 16, 3))
    ident = np.tile(np.arange(16), (3, 1))
    out = np_ref.scramble_batch(x, ident, 4)
    assert np.array_equal(out[..., :3], x) and np.array_equal(out[..., 3:], x)       # identity perm
    assert np.array_equal(np_ref.scramble_batch(x, np.zeros((3, 1), int), 16)[..., 3:], x)   # s = H
    perm = np.stack([rng.permutation(64) for _ in range(3)])
    out = np_ref.scramble_batch(x, perm, 2)
    for b in range(3):
        for c in range(3):
            assert np.array_equal(np.sort(out[b, :, :, 3 + c].ravel()), np.sort(x[b, :, :, c].ravel()))
    assert np.array_equal(out, torch_ref.scramble_batch(x, perm, 2).numpy())
    # explicit index rule
    s, G = 2, 8
    r, c, i, j, b = 3, 5, 1, 0, 2
    pr, pc = divmod(int(perm[b, r * G + c]), G)
    assert np.array_equal(out[b, r * s + i, c * s + j, 3:], x[b, pr * s + i, pc * s + j])


def test_keras_adam_first_step_and_eps_placement():
    g = np.array([1.0, -1.0, 5.0, 1e-3, -1e-9])
    p, m, v = np_ref.keras_adam_step([np.zeros(5)], [g], [np.zeros(5)], [np.zeros(5)], 1, lr=1e-4)
    want = -1e-4 * math.sqrt(1 - 0.999) * g / (math.sqrt(1 - 0.999) * np.abs(g) + 1e-7) / (1 - 0.9) * (1 - 0.9)
    assert np.allclose(p[0], want, rtol=1e-12)
    assert np.allclose(p[0][:3], -1e-4 * np.sign(g[:3]), rtol=1e-4)      # ~ -lr*sign(g)
    # eps OUTSIDE the bias correction: differs from torch.optim.Adam for tiny gradients
    tp = torch.zeros(5, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1e-4, eps=1e-7)
    tp.grad = torch.from_numpy(g.copy())
    opt.step()
    assert abs(float(tp[4]) - p[0][4]) > 1e-7 * 1e-2     # visibly different on the 1e-9 gradient
    assert np.allclose(tp.detach().numpy()[:3], p[0][:3], rtol=1e-4)


def test_zero_weights_closed_form():
    H, B = 32, 2
    params = [np.zeros(s, np.float64) for _, s in np_ref.param_shapes(H, H)]
    rng = np.random.default_rng(5)
    img = (rng.integers(0, 256, (B, H, H, 6)) / 255.0 * 2 - 1)
    fwd = np_ref.lgvae_forward(img, params, np.zeros((B, 128)), np.zeros((B, 128)))
    assert np.all(fwd[0] == 0) and np.all(fwd[1] == 0)
    assert np.allclose(fwd[4], math.log(2.0))                 # sigma = softplus(0)
    l = np_ref.lgvae_losses(img, fwd, 40.0)
    sg = math.log(2.0)
    kl = -0.5 * 128 * (1 + math.log(sg * sg) - sg * sg)
    assert abs(l["x_kl_loss"] - kl) < 1e-9 and abs(l["total_kl_loss"] - 80 * kl) < 1e-7
    want = np.sum(np_ref.discretised_logistic_loss(img[..., :3], 0.0, 0.0), axis=(1, 2, 3)).mean()
    assert abs(l["x_recon_loss"] - want) < 1e-9


def test_two_restatements_agree_and_fd_gradients():
    H, B = 32, 2
    rng = np.random.default_rng(6)
    x = (rng.integers(0, 256, (B, H, H, 3)) / 255.0 * 2 - 1)
    perm = np.stack([rng.permutation(1024) for _ in range(B)])
    img = np_ref.scramble_batch(x, perm, 1)
    params = np_ref.glorot_init(H, H, dtype=np.float64)
    eps = rng.standard_normal((2, B, 128))
    fw = np_ref.lgvae_forward(img, params, eps[0], eps[1])
    tr = torch_ref.RefTrainer(params, 40.0, dtype=torch.float64)
    fwt, lt, g = tr.grads(img, eps[0], eps[1])
    for a, b in zip(fw, fwt):
        assert np.abs(a - b.detach().numpy()).max() < 1e-12
    ln = np_ref.lgvae_losses(img, fw, 40.0)
    for k in ln:
        assert abs(ln[k] - float(lt[k].detach())) < 1e-9 * max(1.0, abs(ln[k]))
    for which, idx in [(0, 5), (4, 100), (8, 77), (20, 33), (28, 10), (38, 3), (39, 2)]:
        fd = np_ref.fd_grad(img, params, eps[0], eps[1], 40.0, which, idx)
        assert abs(fd - float(g[which].flatten()[idx])) < 2e-5 * max(1.0, abs(fd))


def test_param_table_and_work_per_image():
    # BASELINE.md section 2 denominators follow from the layer definitions
    shapes = np_ref.param_shapes(64, 64)
    assert len(shapes) == 40 and sum(int(np.prod(s)) for _, s in shapes) == 8722060
    assert sum(int(np.prod(s)) for _, s in np_ref.param_shapes(32, 32)) == 3204748
    assert [n.split('/')[0] for n, _ in shapes[::10]] == ["encoder_x", "encoder_x_hat", "decoder_x", "decoder_x_hat"]


# ------------------------------------------------------------------ SPLIT-GMVAE restatement (oracle/gm_ref.py)
def test_gm_oracle_kats():
    """Analytic pins of the LGGMVae loss pieces (SURVEY 8c pins (2) + the A9 additions)."""
    import torch
    from oracle import gm_ref, torch_ref
    g = torch.Generator().manual_seed(0)
    mu = torch.randn(5, 16, generator=g, dtype=torch.float64)
    sig = torch.rand(5, 16, generator=g, dtype=torch.float64) + 0.3
    # kl_two_gauss(mu, sig, 0, 1) == kl_divergence(mu, sig)   (vae/trainer.py:17-18 vs :11-15)
    assert abs(float(gm_ref.kl_divergence_two_gauss(mu, sig, 0.0, 1.0)) - float(torch_ref.kl_divergence(mu, sig))) < 1e-12
    # KL(q || q) = 0 ; KL >= 0
    assert abs(float(gm_ref.kl_divergence_two_gauss(mu, sig, mu, sig))) < 1e-12
    assert float(gm_ref.kl_divergence_two_gauss(mu, sig, mu + 1, sig * 2)) > 0
    # categorical term: zero at uniform logits, -> log K at a one-hot limit (vae/trainer.py:161-162)
    K = 30
    assert abs(float(gm_ref.categorical_kl(torch.zeros(4, K, dtype=torch.float64), K))) < 1e-6
    hot = torch.full((4, K), -50.0, dtype=torch.float64); hot[:, 3] = 50.0
    assert abs(float(gm_ref.categorical_kl(hot, K)) - np.log(K)) < 1e-6
    # Gumbel-softmax: rows sum to one; tau -> 0 picks argmax(logits + gumbel)
    logits = torch.randn(6, K, generator=g, dtype=torch.float64)
    u = torch.rand(6, K, generator=g, dtype=torch.float64).clamp(1e-6, 1 - 1e-6)
    y = gm_ref.gumbel_softmax(logits, u, 0.4)
    assert torch.allclose(y.sum(1), torch.ones(6, dtype=torch.float64))
    y0 = gm_ref.gumbel_softmax(logits, u, 1e-3)
    assert torch.equal(y0.argmax(1), (logits - torch.log(-torch.log(u))).argmax(1)) and float(y0.max(1).values.min()) > 0.99


def test_gm_oracle_shapes_counts_and_fd_gradient():
    import torch
    from oracle import gm_ref
    H, B, K = 32, 2, 30
    shapes = gm_ref.gm_param_shapes(H, H, y_size=K)
    assert len(shapes) == 54 and sum(int(np.prod(s)) for _, s in shapes) == 6775370     # 6.78 M params
    params = gm_ref.gm_glorot_init(H, H, seed=2, y_size=K, dtype=np.float64)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (B, H, H, 6)) / 255.0 * 2 - 1
    F_ = (H // 8) ** 2 * 128
    a = (rng.standard_normal((B, 128)), rng.standard_normal((B, 128)), rng.uniform(0.05, 0.95, (B, K)),
         (rng.uniform(size=(B, 1024)) > 0.2) * 1.0, (rng.uniform(size=(B, F_)) > 0.2) * 1.0)
    tr = gm_ref.GMRefTrainer(params, 40.0, 40.0, y_size=K, dtype=torch.float64)
    _, losses, g = tr.grads(img, *a)
    assert all(np.isfinite(float(v)) for v in losses.values())
    # central finite differences on a few coordinates of representative tensors (conv, dense, prior head, bias)
    for ti, idx in [(0, (1, 2, 0, 5)), (6, (100, 7)), (10, (3, 4)), (16, (2, 9)), (17, (4,)), (22, (11, 3)), (34, (7, 50))]:
        h = 1e-5
        base = tr.params[ti].detach().clone()
        vals = []
        for sgn in (+1, -1):
            with torch.no_grad():
                tr.params[ti].copy_(base); tr.params[ti][idx] += sgn * h
            vals.append(float(tr.forward_losses(img, *a)[1]["total_loss"]))
        with torch.no_grad():
            tr.params[ti].copy_(base)
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - float(g[ti][idx])) <= 1e-5 * max(1.0, abs(fd)), (ti, idx, fd, float(g[ti][idx]))
