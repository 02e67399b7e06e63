"""GPU parity: every HIP kernel family, called through the C ABI, against the oracle on the same
seeded inputs.  Bit-exact for the index bookkeeping (scramble); stated fp32 / bf16 tolerances
for floating point."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ref, torch_ref

pytestmark = pytest.mark.gpu

# fp32 path: exact-f32 MFMA / fp32 pointwise vs the fp32 oracle -- only summation order differs
F32_RTOL, F32_ATOL = 1e-4, 1e-5
# bf16 path: operands rounded to 8 significant bits, fp32 accumulate
BF16_RTOL, BF16_ATOL = 3e-2, 3e-2


@pytest.fixture(scope="module")
def ops(lib_built):
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from split_vae_amd import ops as o
    return o


def levels(rng, shape):
    """the reference's data domain: x/255*2-1 (vae/data.py:52)"""
    return (rng.integers(0, 256, size=shape) / 255.0 * 2 - 1).astype(np.float32)


# ------------------------------------------------------------------ scramble: bit exact
@pytest.mark.parametrize("H,patch", [(32, 1), (32, 4), (64, 8), (64, 64), (32, 2)])
def test_scramble_bit_exact(ops, H, patch):
    rng = np.random.default_rng(0)
    B = 5
    x = levels(rng, (B, H, H, 3))
    G2 = (H // patch) ** 2
    perm = np.stack([rng.permutation(G2) for _ in range(B)]).astype(np.int32)
    want = np_ref.scramble_batch(x, perm, patch)
    got = ops.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), patch).cpu().numpy()
    assert got.dtype == np.float32
    assert np.array_equal(got, want.astype(np.float32))          # every bit
    assert np.array_equal(got[..., :3], x)                        # channels 0-2 are the input


def test_scramble_identity_and_multiset(ops):
    rng = np.random.default_rng(1)
    x = torch.from_numpy(levels(rng, (3, 32, 32, 3))).cuda()
    ident = torch.arange(64, dtype=torch.int32).repeat(3, 1).cuda()
    out = ops.scramble_gather(x, ident, 4)
    assert torch.equal(out[..., 3:], x)
    perm = ops.random_perm(3, 64, seed=7)
    out = ops.scramble_gather(x, perm, 4)
    for b in range(3):
        for c in range(3):
            assert torch.equal(out[b, :, :, 3 + c].flatten().sort().values, x[b, :, :, c].flatten().sort().values)


def test_random_perm_properties(ops):
    p = ops.random_perm(64, 1024, seed=123, step=5).cpu().numpy()
    for row in p:
        assert np.array_equal(np.sort(row), np.arange(1024))
    assert len({tuple(r) for r in p}) == 64
    p2 = ops.random_perm(64, 1024, seed=123, step=5).cpu().numpy()
    assert np.array_equal(p, p2)                                   # counter-based: reproducible
    # data-parallel invariance: rank r's rows == rows [off, off+n) of the single-process draw
    shard = ops.random_perm(16, 1024, seed=123, step=5, sample_offset=32).cpu().numpy()
    assert np.array_equal(shard, p[32:48])
    assert not np.array_equal(ops.random_perm(64, 1024, seed=123, step=6).cpu().numpy(), p)


# ------------------------------------------------------------------ discretised logistic NLL
def _dll_inputs(rng, B, H):
    x6 = levels(rng, (B, H, H, 6))
    x6[0, 0, :, :] = -1.0           # edge bins (vae/trainer.py:37: x < -0.999 / x > 0.999)
    x6[0, 1, :, :] = 1.0
    out6 = rng.standard_normal((B, H, H, 6)).astype(np.float32)
    out6[..., 3:] = rng.uniform(-6.0, 1.0, size=(B, H, H, 3))    # log-scales incl. very sharp ones
    out6[1, :4, :, 3:] = 3.0                                     # wide: cdf_delta <= 1e-5 branch needs tiny inv_stdv
    out6[1, 4:8, :, 3:] = 9.0
    return x6, out6


@pytest.mark.parametrize("ch_off", [0, 3])
def test_dlogistic_nll_and_grad(ops, ch_off):
    rng = np.random.default_rng(2)
    B, H = 4, 32
    x6, out6 = _dll_inputs(rng, B, H)
    xt = torch.from_numpy(x6).double()
    ot = torch.from_numpy(out6).double().requires_grad_(True)
    nll_ref = torch_ref.discretised_logistic_loss(xt[..., ch_off:ch_off + 3], ot[..., :3], ot[..., 3:]).sum(dim=(1, 2, 3))
    (nll_ref.mean()).backward()
    nll, grad = ops.dlogistic_nll(torch.from_numpy(x6).cuda(), ch_off, torch.from_numpy(out6).cuda(),
                                  grad_dtype=torch.float32, grad_scale=1.0 / B)
    torch.testing.assert_close(nll.cpu().double(), nll_ref.detach(), rtol=2e-5, atol=1e-3)
    g = grad.cpu().double()
    assert torch.count_nonzero(g[..., 6:]) == 0
    torch.testing.assert_close(g[..., :6], ot.grad, rtol=2e-4, atol=2e-6)
    # bf16 gradient output = rounded fp32 gradient
    _, gb = ops.dlogistic_nll(torch.from_numpy(x6).cuda(), ch_off, torch.from_numpy(out6).cuda(),
                              grad_dtype=torch.bfloat16, grad_scale=1.0 / B)
    torch.testing.assert_close(gb.float().cpu(), grad.cpu().bfloat16().float(), rtol=1e-2, atol=1e-7)


def test_dlogistic_sums_to_one(ops):
    """KAT: sum over the 256 bins of exp(-nll) == 1 for any (m, log_scale).
    One probed element per image (B = 256 images, one per bin); every other element of the image
    is parked at x=-1, m=-3, log_scale=-5 where its nll is exactly 0 in fp32."""
    ks = (np.arange(256) / 255.0 * 2 - 1).astype(np.float32)
    for m, ls in [(0.1, -2.0), (-0.7, -5.0), (0.9, -1.0), (0.0, -3.5), (0.3, -6.0)]:
        x6 = np.full((256, 8, 8, 6), -1.0, np.float32)
        out6 = np.zeros((256, 8, 8, 6), np.float32)
        out6[..., :3] = -3.0
        out6[..., 3:] = -5.0
        x6[:, 0, 0, 0] = ks
        out6[:, 0, 0, 0] = m
        out6[:, 0, 0, 3] = ls
        nll, _ = ops.dlogistic_nll(torch.from_numpy(x6).cuda(), 0, torch.from_numpy(out6).cuda())
        per = nll.cpu().double().numpy()
        assert abs(np.exp(-per).sum() - 1.0) < 1e-4, (m, ls, np.exp(-per).sum())


def test_dlogistic_bitwise_deterministic(ops):
    """the ELBO reduction uses a fixed-order shuffle + LDS tree: same bits every launch."""
    rng = np.random.default_rng(12)
    x6, out6 = _dll_inputs(rng, 8, 64)
    a = ops.dlogistic_nll(torch.from_numpy(x6).cuda(), 0, torch.from_numpy(out6).cuda(), grad_dtype=torch.bfloat16)
    b = ops.dlogistic_nll(torch.from_numpy(x6).cuda(), 0, torch.from_numpy(out6).cuda(), grad_dtype=torch.bfloat16)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


# ------------------------------------------------------------------ reparam + KL
def test_reparam_kl_fwd_bwd(ops):
    rng = np.random.default_rng(3)
    B, L = 6, 128
    pre = rng.standard_normal((B, 2 * L)).astype(np.float32) * 2
    bias = rng.standard_normal((2 * L,)).astype(np.float32) * 0.1
    eps = rng.standard_normal((B, L)).astype(np.float32)
    z_mean, z_sig, z, z_lp, kl, eps_out = ops.reparam_kl_fwd(torch.from_numpy(pre).cuda(), torch.from_numpy(bias).cuda(),
                                                             torch.from_numpy(eps).cuda(), z_dtype=torch.float32)
    pt = torch.from_numpy(pre).double().requires_grad_(True)
    bt = torch.from_numpy(bias).double()
    mu = pt[:, :L] + bt[:L]
    sg = F.softplus(pt[:, L:] + bt[L:])
    zz = mu + sg * torch.from_numpy(eps).double()
    kl_ref = -0.5 * torch.sum(1 + torch.log(sg ** 2) - mu ** 2 - torch.exp(torch.log(sg ** 2)), dim=1)
    torch.testing.assert_close(z_mean.cpu().double(), mu.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(z_sig.cpu().double(), sg.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(z.cpu().double(), zz.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(z_lp.cpu().double(), zz.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(kl.cpu().double(), kl_ref.detach(), rtol=1e-4, atol=1e-3)
    assert torch.equal(eps_out.cpu(), torch.from_numpy(eps))
    # KAT: kl(0,1) == 0 ; kl(mu,1) = 0.5|mu|^2
    pre0 = np.zeros((2, 2 * L), np.float32)
    pre0[:, L:] = math.log(math.e - 1)            # softplus^-1(1)
    pre0[1, :L] = 0.5
    _, s1, _, _, kl0, _ = ops.reparam_kl_fwd(torch.from_numpy(pre0).cuda(), torch.zeros(2 * L).cuda(),
                                             torch.zeros(2, L).cuda(), z_dtype=torch.float32)
    assert abs(float(kl0[0])) < 1e-4 and abs(float(kl0[1]) - 0.5 * L * 0.25) < 1e-3
    # backward
    dz = rng.standard_normal((B, L)).astype(np.float32)
    dz2 = rng.standard_normal((B, L)).astype(np.float32)
    beta = 40.0
    loss = (zz * torch.from_numpy(dz + dz2).double()).sum() + beta * kl_ref.mean()
    loss.backward()
    g = ops.reparam_kl_bwd(torch.from_numpy(dz).cuda(), z_mean, z_sig, eps_out, beta / B, g_dtype=torch.float32,
                           dz2=torch.from_numpy(dz2).cuda())
    torch.testing.assert_close(g.cpu().double(), pt.grad, rtol=2e-4, atol=2e-5)


def test_reparam_philox_normal(ops):
    B, L = 512, 128
    pre = torch.zeros((B, 2 * L), device="cuda")
    bias = torch.zeros((2 * L,), device="cuda")
    *_, e1 = ops.reparam_kl_fwd(pre, bias, None, z_dtype=torch.float32, seed=11, step=3)
    *_, e2 = ops.reparam_kl_fwd(pre, bias, None, z_dtype=torch.float32, seed=11, step=3)
    assert torch.equal(e1, e2)
    e = e1.cpu().double()
    assert abs(float(e.mean())) < 0.02 and abs(float(e.std()) - 1.0) < 0.02
    assert abs(float((e ** 4).mean()) - 3.0) < 0.15
    *_, sh = ops.reparam_kl_fwd(pre[:64], bias, None, z_dtype=torch.float32, seed=11, step=3, sample_offset=128)
    assert torch.equal(sh, e1[128:192])                            # 1-GPU == N-GPU samples
    *_, e3 = ops.reparam_kl_fwd(pre, bias, None, z_dtype=torch.float32, seed=11, step=3, stream_id=1)
    assert not torch.equal(e3, e1)


# ------------------------------------------------------------------ Keras Adam
def test_adam_matches_keras_formula(ops):
    rng = np.random.default_rng(4)
    n = 10007 * 4
    p0 = rng.standard_normal(n).astype(np.float32)
    p = torch.from_numpy(p0.copy()).cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    rp = [torch.from_numpy(p0.copy()).double()]
    rm, rv = [torch.zeros(n).double()], [torch.zeros(n).double()]
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32) * 10 ** rng.uniform(-6, 1)
        ops.adam_step(p, torch.from_numpy(g).cuda(), m, v, t, lr=1e-4)
        torch_ref.keras_adam_(rp, [torch.from_numpy(g).double()], rm, rv, t, lr=1e-4)
        torch.testing.assert_close(p.cpu().double(), rp[0], rtol=1e-6, atol=1e-7)
    # KAT: first step from zero state ~ -lr*sign(g)
    p = torch.zeros(8, device="cuda"); m = torch.zeros(8, device="cuda"); v = torch.zeros(8, device="cuda")
    g = torch.tensor([1., -1., 5., -5., 1e-2, -1e-2, 3., -3.], device="cuda")
    ops.adam_step(p, g, m, v, 1, lr=1e-4)
    torch.testing.assert_close(p.cpu(), -1e-4 * torch.sign(g.cpu()), rtol=1e-3, atol=0)


# ------------------------------------------------------------------ bilinear 2x
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_upsample_fwd_bwd(ops, dtype, tol):
    rng = np.random.default_rng(5)
    B, H, C = 2, 8, 32
    x = torch.from_numpy(rng.standard_normal((B, H, H, C)).astype(np.float32)).to(dtype)
    xr = x.double().requires_grad_(True)
    up = torch_ref.resize_bilinear_2x(xr)
    got = ops.upsample2x_fwd(x.cuda())
    torch.testing.assert_close(got.cpu().double(), up.detach(), rtol=tol, atol=tol)
    ghi = torch.from_numpy(rng.standard_normal((B, 2 * H, 2 * H, C)).astype(np.float32)).to(dtype)
    up.backward(ghi.double())
    mask = torch.from_numpy(rng.standard_normal((B, H, H, C)).astype(np.float32)).to(dtype)
    glo = ops.upsample2x_bwd(ghi.cuda(), mask.cuda())
    want = xr.grad * (mask.double() > 0)
    torch.testing.assert_close(glo.cpu().double(), want, rtol=tol, atol=4 * tol)
    glo2 = ops.upsample2x_bwd(ghi.cuda(), None)
    torch.testing.assert_close(glo2.cpu().double(), xr.grad, rtol=tol, atol=4 * tol)


def test_fp32_resize_adjoint_row_window_is_bitwise_the_per_output_kernel(lib_built, tmp_path):
    """upsample2x_bwd_rows_kernel (a thread walks eight low-res rows with the 4 x 4 hi-res window in registers) against the one-thread-per-output
    kernel (SV_UPS_BWD_PLAIN=1, latched at first use: separate processes): same products in the same order, bit-identical; and against the fp64
    adjoint of the resize.  Shapes: the fp32 step's three launches at a small batch (band edges, image edges, with and without the ReLU mask)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "g = torch.Generator().manual_seed(11)\n"
        "outs = []\n"
        "for (B, H, C) in ((70, 32, 32), (130, 16, 64), (260, 8, 128)):\n"
        "    ghi = torch.randn(B, 2 * H, 2 * H, C, generator=g).cuda()\n"
        "    mask = torch.randn(B, H, H, C, generator=g).cuda()\n"
        "    outs += [ops.upsample2x_bwd(ghi, mask).cpu().numpy(), ops.upsample2x_bwd(ghi, None).cpu().numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env in (("rows", {}), ("plain", {"SV_UPS_BWD_PLAIN": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    for key in res[0].files:
        assert np.array_equal(res[0][key], res[1][key]), key
        assert np.abs(res[0][key]).max() > 0
    # the fp64 adjoint on the first shape (no mask)
    g = torch.Generator().manual_seed(11)
    ghi = torch.randn(70, 64, 64, 32, generator=g)
    xr = torch.zeros(70, 32, 32, 32, dtype=torch.float64, requires_grad=True)
    torch_ref.resize_bilinear_2x(xr).backward(ghi.double())
    torch.testing.assert_close(torch.from_numpy(res[0]["arr_1"]).double(), xr.grad, rtol=1e-6, atol=4e-6)


def test_upsample_stencil_kat(ops):
    """tf.image.resize half-pixel stencil on i^2: [0, .25, .75, 1.75, ...]."""
    H = 8
    v = (torch.arange(H, dtype=torch.float32) ** 2)
    x = v[None, :, None, None].expand(1, H, H, 4).contiguous().cuda()
    up = ops.upsample2x_fwd(x)[0, :, 0, 0].cpu()
    want = torch.tensor([0, .25, .75, 1.75, 3.25, 5.25, 7.75, 10.75, 14.25, 18.25, 22.75, 27.75, 33.25, 39.25, 45.75, 49.0])
    torch.testing.assert_close(up, want, rtol=0, atol=1e-5)


# ------------------------------------------------------------------ conv layers: every geometry of the model
LAYERS = [  # name, H, Cin, Cout, k, stride, act, y_f32
    ("e1", 32, 3, 32, 6, 2, "relu", False),
    ("e2", 16, 32, 64, 6, 2, "relu", False),
    ("e3", 8, 64, 128, 4, 2, "relu", False),
    ("d2", 4, 128, 128, 4, 1, "relu", False),
    ("d3", 8, 128, 64, 4, 1, "relu", False),
    ("d4", 16, 64, 32, 6, 1, "relu", False),
    ("d5", 32, 32, 6, 6, 1, None, True),
    ("e1_64", 64, 3, 32, 6, 2, "relu", False),
    ("d5_64", 64, 32, 6, 6, 1, None, True),
    ("d4_64", 32, 64, 32, 6, 1, "relu", False),      # two 16-pixel column strips per image (wgrad_roll.hip)
    ("d4_128", 64, 64, 32, 6, 1, "relu", False),     # four strips: first / inner / last strip variants of the rolling-window kernel
    ("e2_64", 32, 32, 64, 6, 2, "relu", False),      # 16 x 16 class grid: the merged parity classes of its input gradient on the row-ring kernel
    ("d2_64", 8, 128, 128, 4, 1, "relu", False),     # 8 x 8 grid: image pairs per strip on the row-ring kernel (B = 3: a half pair)
    ("e3_64", 16, 64, 128, 4, 2, "relu", False),     # k 4 stride 2 onto the 8 x 8 grid: shifted space-to-depth blocks on image pairs
]


def _pad_c(t, c):
    if t.shape[-1] == c:
        return t
    out = torch.zeros(t.shape[:-1] + (c,), dtype=t.dtype)
    out[..., :t.shape[-1]] = t
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_fwd_dgrad_wgrad(ops, layer, dtype):
    name, H, Cin, Cout, k, s, act, yf32 = layer
    rng = np.random.default_rng(sum(map(ord, name)))
    B = 3
    rtol, atol = (F32_RTOL, F32_ATOL) if dtype == torch.float32 else (BF16_RTOL, BF16_ATOL)
    x = torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32)).to(dtype)
    fan = k * k * (Cin + Cout)
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / fan)
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=act, dtype=dtype, y_f32=yf32)
    conv.prep(w.cuda())
    xg = _pad_c(x, conv.desc.ldx).cuda()
    y = conv.fwd(xg, b.cuda())
    # reference in fp64 from the SAME (rounded) operands
    wr = w.to(dtype).double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = torch_ref.conv2d_same(xr, wr, br, s, act)
    scale = float(yr.abs().max())
    torch.testing.assert_close(y[..., :Cout].double().cpu(), yr.detach(), rtol=rtol, atol=atol * scale)
    # backward: dy is the gradient w.r.t. the pre-activation (ReLU mask already applied upstream)
    OH = yr.shape[1]
    dy = torch.from_numpy(rng.standard_normal((B, OH, OH, Cout)).astype(np.float32)).to(dtype)
    pre = torch_ref.conv2d_same(xr, wr, br, s, None)
    pre.backward(dy.double())
    dyg = _pad_c(dy, (Cout + 7) // 8 * 8).cuda()
    dw, db = conv.wgrad(xg, dyg)
    torch.testing.assert_close(dw.double().cpu(), wr.grad, rtol=rtol, atol=atol * float(wr.grad.abs().max()))
    torch.testing.assert_close(db.double().cpu(), br.grad, rtol=rtol, atol=atol * float(br.grad.abs().max()))
    # two-stage (slab + fixed-order reduce) flush: same numbers, and bitwise reproducible
    dw2, db2 = conv.wgrad(xg, dyg, workspace=True)
    torch.testing.assert_close(dw2.double().cpu(), wr.grad, rtol=rtol, atol=atol * float(wr.grad.abs().max()))
    dw3, _ = conv.wgrad(xg, dyg, workspace=True)
    assert torch.equal(dw2, dw3)                             # tile kernel + slab flush on every conv layer, at both precisions (wgrad_tile_f32.hip)
    if name.startswith("e1"):
        return   # first conv: no data gradient in the model
    mask = torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32)).to(dtype)
    dx = conv.dgrad(dyg, relu_mask=_pad_c(mask, conv.desc.ldx).cuda())
    want = xr.grad * (mask.double() > 0)
    torch.testing.assert_close(dx[..., :Cin].double().cpu(), want, rtol=rtol, atol=atol * float(xr.grad.abs().max()))
    dx2 = conv.dgrad(dyg, f32_atomic=True)
    torch.testing.assert_close(dx2[..., :Cin].double().cpu(), xr.grad, rtol=rtol, atol=atol * float(xr.grad.abs().max()))


@pytest.mark.parametrize("layer", [l for l in LAYERS if l[0] in ("d3", "d4", "d4_64", "d4_128", "d5", "d5_64")], ids=lambda l: l[0])
def test_conv_fused_upsample_equals_materialised(ops, layer):
    """ups_in: the tile staging interpolates from the low-res tensor.  Same blend order as the
    stand-alone resize kernel -> the forward output is BITWISE the unfused result; the weight
    gradient agrees to accumulation order."""
    name, H, Cin, Cout, k, s, act, yf32 = layer
    rng = np.random.default_rng(sum(map(ord, name)) + 1)
    B = 3
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).bfloat16().cuda()
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)).cuda() * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)).cuda() * 0.1
    plain = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=act, dtype=torch.bfloat16, y_f32=yf32)
    fused = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=act, dtype=torch.bfloat16, y_f32=yf32, ups_in=True)
    plain.prep(w); fused.prep(w)
    x_hi = ops.upsample2x_fwd(x_lo)
    # the resize itself against the oracle (tf.image.resize semantics)
    ref_hi = torch_ref.resize_bilinear_2x(x_lo.float().cpu().double())
    torch.testing.assert_close(x_hi.float().cpu().double(), ref_hi, rtol=1e-2, atol=1e-2)
    y0 = plain.fwd(x_hi, b)
    y1 = fused.fwd(x_lo, b)
    if name.startswith("d5"):
        # the head runs in POLYPHASE form (conv_geom.h: svg_poly): composite weights round once to bf16 instead of the
        # upsampled activations, so it agrees with the materialised path to bf16 accuracy -- on the border ring (where
        # poly_fix.hip removes the taps that leave the image) as well as inside
        a, r = y1.double().cpu(), y0.double().cpu()
        assert float((a - r).norm() / r.norm()) < 4e-3
        ring = torch.ones(H, H, dtype=torch.bool)
        ring[3:H - 3, 3:H - 3] = False
        assert float((a - r)[:, ring].norm() / r[:, ring].norm()) < 4e-3
        assert float((a - r).abs().max()) < 2e-2 * float(r.abs().max())
    elif name in ("d4_64", "d4_128"):
        # 32 x 32 grid (and larger): the fused form runs on the row-ring kernel, the materialised one on the tile kernel (other summation order)
        torch.testing.assert_close(y1.float(), y0.float(), rtol=2e-2, atol=2e-2 * float(y0.float().abs().max()))
    else:
        assert torch.equal(y0, y1)
    dy = torch.from_numpy(rng.standard_normal((B, H, H, (Cout + 7) // 8 * 8)).astype(np.float32)).bfloat16().cuda()
    if Cout % 8:
        dy[..., Cout:] = 0
    dw0, db0 = plain.wgrad(x_hi, dy, workspace=True)
    dw1, db1 = fused.wgrad(x_lo, dy, workspace=True)
    torch.testing.assert_close(dw1, dw0, rtol=1e-4, atol=1e-5 * float(dw0.abs().max()))
    torch.testing.assert_close(db1, db0, rtol=1e-4, atol=1e-5 * float(db0.abs().max()))


@pytest.mark.parametrize("layer", [("d4_64", 32, 64, 32, 6, 515), ("d3_64", 16, 128, 64, 4, 259)], ids=["d4", "d3"])
def test_forward_with_the_resize_on_the_matrix_pipe(ops, layer):
    """From 512 (d4) / 256 (d3) images per launch (whole images per unit of work: bands == 1) the d4 / d3 forward (UpSampling2D(bilinear) ->
    Conv2D(32, 6) / Conv2D(64, 4), vae/model.py:154-155,:163-165) stages its input by LDS-DMA of the raw low-res rows and blends them with MFMAs against constant weight
    operands (row_conv.hip: RowCfg::MB) instead of the VALU blend.  Against the fp64 composition resize -> conv on the same bf16 inputs
    (every row and column of a sample of images: first / last units of the walk, image edges) and against the small-batch (VALU blend)
    form on the same images: the MFMA blend rounds the upsampled activation once instead of per lerp stage, so the two agree to a bf16
    ulp of the activations, not bitwise."""
    name, H, Cin, Cout, k, B = layer                          # B: a few units more than workgroup slots (512 / 256): some walk two images
    rng = np.random.default_rng(77 + H)
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).bfloat16()
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act="relu", dtype=torch.bfloat16, ups_in=True)
    conv.prep(w.cuda())
    y = conv.fwd(x_lo.cuda(), b.cuda())
    torch.cuda.synchronize()
    sel = [0, 1, 2, 255, 256, 257] + ([510, 511, 512, 513, 514] if B > 514 else [B - 2, B - 1])
    xs = x_lo[sel].double()
    ref = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(xs), w.bfloat16().double(), b.double(), 1, "relu")
    got = y[sel][..., :Cout].double().cpu()
    assert got.shape == ref.shape
    err = got - ref
    assert float(err.norm() / ref.norm()) < 4e-3, float(err.norm() / ref.norm())
    assert float(err.abs().max()) < 2e-2 * float(ref.abs().max())
    for sl in ((slice(0, 3), slice(None)), (slice(H - 3, H), slice(None)), (slice(None), slice(0, 3)), (slice(None), slice(H - 3, H))):
        e, r = err[:, sl[0], sl[1]], ref[:, sl[0], sl[1]]
        assert float(e.norm() / r.norm()) < 4e-3            # SAME padding rows / columns and the edge-clamped resize
    # the same images through the small-batch form (row bands, VALU blend)
    small = ops.Conv2D(len(sel), H, H, Cin, Cout, k, 1, act="relu", dtype=torch.bfloat16, ups_in=True)
    small.prep(w.cuda())
    y_s = small.fwd(x_lo[sel].contiguous().cuda(), b.cuda())[..., :Cout].double().cpu()
    d = got - y_s
    assert float(d.norm() / ref.norm()) < 3e-3 and float(d.abs().max()) < 2e-2 * float(ref.abs().max())
    # deterministic: a second launch gives the same bits
    y2 = conv.fwd(x_lo.cuda(), b.cuda())
    assert torch.equal(y, y2)


@pytest.mark.parametrize("B", [1, 5])
@pytest.mark.parametrize("H", [16, 32, 64])
def test_polyphase_head_against_upsample_then_conv_fp64(ops, H, B):
    """2x bilinear upsample -> 6x6 SAME conv (vae/model.py:163-167, d5) as ONE conv over the low-res tensor + the border
    fix, against the fp64 composition resize -> zero-padded conv on the same bf16 inputs: every pixel, with the border
    ring (rows / columns 0, 1, H-3..H-1, whose taps leave the image) checked on its own, and with a bias."""
    rng = np.random.default_rng(H * 10 + B)
    Cin, Cout, k = 32, 6, 6
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).bfloat16()
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
    conv.prep(w.cuda())
    y = conv.fwd(x_lo.cuda(), b.cuda()).double().cpu()                         # border terms through the workspace
    y_at = conv.fwd(x_lo.cuda(), b.cuda(), workspace=False).double().cpu()     # ... added with atomics after the conv
    ref = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(x_lo.double()), w.double(), b.double(), 1, None)
    assert y.shape == ref.shape
    torch.testing.assert_close(y_at, y, rtol=0, atol=1e-5 * float(ref.abs().max()))   # fp32 addition order only
    err = y - ref
    assert float(err.norm() / ref.norm()) < 4e-3
    for sl in ((slice(0, 2), slice(None)), (slice(H - 3, H), slice(None)), (slice(None), slice(0, 2)), (slice(None), slice(H - 3, H))):
        e, r = err[:, sl[0], sl[1]], ref[:, sl[0], sl[1]]
        assert float(e.norm() / r.norm()) < 4e-3, sl
    assert float(err.abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("K,N", [(2048, 256), (256, 2048), (128, 8192), (8192, 256)])
def test_dense_as_conv(ops, K, N, dtype):
    """Dense layers (vae/model.py:41-42,:152) are the 1x1 / H=W=1 instance of the same kernels."""
    rng = np.random.default_rng(K + N)
    B = 37     # ragged: not a multiple of the 128-row tile
    rtol, atol = (F32_RTOL, F32_ATOL) if dtype == torch.float32 else (BF16_RTOL, BF16_ATOL)
    x = torch.from_numpy(rng.standard_normal((B, 1, 1, K)).astype(np.float32)).to(dtype)
    w = (torch.from_numpy(rng.uniform(-1, 1, (1, 1, K, N)).astype(np.float32)) * math.sqrt(6.0 / (K + N)))
    b = torch.from_numpy(rng.standard_normal((N,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, 1, 1, K, N, 1, 1, act="relu", dtype=dtype)
    conv.prep(w.cuda())
    y = conv.fwd(x.cuda(), b.cuda())
    wr = w.to(dtype).double()[0, 0].requires_grad_(True)
    xr = x.double()[:, 0, 0].requires_grad_(True)
    pre = xr @ wr + b.double()
    torch.testing.assert_close(y[:, 0, 0].double().cpu(), F.relu(pre).detach(), rtol=rtol, atol=atol * float(pre.abs().max()))
    dy = torch.from_numpy(rng.standard_normal((B, 1, 1, N)).astype(np.float32)).to(dtype)
    pre.backward(dy.double()[:, 0, 0])
    dw, db = conv.wgrad(x.cuda(), dy.cuda())
    torch.testing.assert_close(dw[0, 0].double().cpu(), wr.grad, rtol=rtol, atol=atol * float(wr.grad.abs().max()))
    dx = conv.dgrad(dy.cuda(), f32_atomic=True)
    torch.testing.assert_close(dx[:, 0, 0].double().cpu(), xr.grad, rtol=rtol, atol=atol * float(xr.grad.abs().max()))


# ------------------------------------------------------------------ row-ring kernel (row_conv.hip): the CelebA-64 decoder geometries
ROW_LAYERS = [  # name, H (conv input = output size), Cin, Cout, k
    ("d3_64", 16, 128, 64, 4),
    ("d4_64", 32, 64, 32, 6),
]


@pytest.mark.parametrize("B", [3, 70])      # 3: every image cut into row bands; 70: several images per workgroup + the twin-net launch shape
@pytest.mark.parametrize("layer", ROW_LAYERS, ids=[l[0] for l in ROW_LAYERS])
def test_row_ring_conv_forward_and_dgrad(ops, layer, B, monkeypatch):
    """Weights in registers, image rows rolling through the LDS ring (fused 2x bilinear upsample on the forward): forward
    against fp64 conv(resize(x)) from the same bf16 operands, input gradient against fp64 autograd; and against the LDS-tile
    kernel (SV_NO_ROWCONV) to accumulation order."""
    name, H, Cin, Cout, k = layer
    rng = np.random.default_rng(sum(map(ord, name)) + B)
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).bfloat16()
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act="relu", dtype=torch.bfloat16, ups_in=True)
    conv.prep(w.cuda())
    y = conv.fwd(x_lo.cuda(), b.cuda())
    # the kernel blends in fp32 and rounds the hi-res pixels to bf16 (as upsample2x_fwd does): reference from those operands
    x_hi = ops.upsample2x_fwd(x_lo.cuda()).float().cpu().double()
    torch.testing.assert_close(x_hi, torch_ref.resize_bilinear_2x(x_lo.double()), rtol=1e-2, atol=1e-2)
    xr = x_hi.clone().requires_grad_(True)
    wr = w.bfloat16().double()
    pre = torch_ref.conv2d_same(xr, wr, b.double(), 1, None)
    yr = torch.relu(pre).detach()
    torch.testing.assert_close(y.double().cpu(), yr, rtol=BF16_RTOL, atol=1e-2 * float(yr.abs().max()))
    assert float((y.double().cpu() - yr).norm() / yr.norm()) < 4e-3          # bf16 output rounding only
    # input gradient (at the virtual hi-res input), no mask
    dy = torch.from_numpy(rng.standard_normal((B, H, H, Cout)).astype(np.float32)).bfloat16()
    pre.backward(dy.double())
    dx = conv.dgrad(dy.cuda())
    torch.testing.assert_close(dx.double().cpu(), xr.grad, rtol=BF16_RTOL, atol=1e-2 * float(xr.grad.abs().max()))
    assert float((dx.double().cpu() - xr.grad).norm() / xr.grad.norm()) < 4e-3


def test_row_ring_matches_tile_kernel_in_a_subprocess(lib_built, tmp_path):
    """A/B inside the product: the same forward + dgrad with SV_NO_ROWCONV=1 (LDS-tile kernel) and without, in separate
    processes (the knob is latched at first use); fp32 accumulation order is the only difference."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "g = torch.Generator().manual_seed(5)\n"
        "outs = []\n"
        "for (H, Cin, Cout, k) in ((16, 128, 64, 4), (32, 64, 32, 6)):\n"
        "    B = 9\n"
        "    x = torch.randn(B, H // 2, H // 2, Cin, generator=g).bfloat16().cuda()\n"
        "    w = (torch.randn(k, k, Cin, Cout, generator=g) * 0.03).cuda()\n"
        "    dy = torch.randn(B, H, H, Cout, generator=g).bfloat16().cuda()\n"
        "    c = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act='relu', dtype=torch.bfloat16, ups_in=True); c.prep(w)\n"
        "    outs += [c.fwd(x, torch.zeros(Cout, device='cuda')).float().cpu().numpy(), c.dgrad(dy).float().cpu().numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env in (("row", {}), ("tile", {"SV_NO_ROWCONV": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    for key in res[0].files:
        a, b = res[0][key].astype(np.float64), res[1][key].astype(np.float64)
        assert np.linalg.norm(a - b) <= 4e-3 * np.linalg.norm(b), key      # both rounded to bf16 from differently ordered fp32 sums
        assert not np.array_equal(a, np.zeros_like(a))


def test_merged_parity_classes_match_one_problem_per_class(lib_built, tmp_path):
    """e2's input gradient (k 6, stride 2: the four parity classes read the same dY window) as ONE 128-column problem
    with the class -> sub-pixel store, against one problem per class (SV_NO_CLS_MERGE=1), masked and unmasked, at a batch
    with a ragged last tile: the same products in the same K order, so bit-exact."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "g = torch.Generator().manual_seed(11)\n"
        "outs = []\n"
        "for B in (1, 7):\n"
        "    w = (torch.randn(6, 6, 32, 64, generator=g) * 0.03).cuda()\n"
        "    dy = torch.randn(B, 8, 8, 64, generator=g).bfloat16().cuda()\n"
        "    mask = torch.randn(B, 16, 16, 32, generator=g).bfloat16().cuda()\n"
        "    c = ops.Conv2D(B, 16, 16, 32, 64, 6, 2, act='relu', dtype=torch.bfloat16); c.prep(w)\n"
        "    outs += [c.dgrad(dy).float().cpu().numpy(), c.dgrad(dy, relu_mask=mask).float().cpu().numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env in (("merged", {}), ("classes", {"SV_NO_CLS_MERGE": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    for key in res[0].files:
        assert np.array_equal(res[0][key], res[1][key]), key
        assert np.count_nonzero(res[0][key]) > res[0][key].size // 4


def test_merged_parity_classes_on_the_row_ring_kernel(lib_built, tmp_path):
    """The same input gradient at CelebA-64's size (32 x 32 input, 16 x 16 class grid), where the merged problem runs on the weight-stationary
    row-ring kernel (row_conv.hip, RC_e2g: class -> sub-pixel store + ReLU mask in its epilogue), against the tile kernel (SV_RC_NO_CLS=1): other
    K order, so bf16 roundings of differently ordered fp32 sums (4e-3 of the norm); masked zeros in the same places."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "g = torch.Generator().manual_seed(12)\n"
        "outs = []\n"
        "for B in (1, 5, 64):\n"
        "    w = (torch.randn(6, 6, 32, 64, generator=g) * 0.03).cuda()\n"
        "    dy = torch.randn(B, 16, 16, 64, generator=g).bfloat16().cuda()\n"
        "    mask = torch.randn(B, 32, 32, 32, generator=g).bfloat16().cuda()\n"
        "    c = ops.Conv2D(B, 32, 32, 32, 64, 6, 2, act='relu', dtype=torch.bfloat16); c.prep(w)\n"
        "    outs += [c.dgrad(dy).float().cpu().numpy(), c.dgrad(dy, relu_mask=mask).float().cpu().numpy(), (mask > 0).cpu().numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env in (("row", {}), ("tile", {"SV_RC_NO_CLS": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    keys = res[0].files
    for i in range(0, len(keys), 3):
        for k in keys[i:i + 2]:
            a, b = res[0][k].astype(np.float64), res[1][k].astype(np.float64)
            assert np.linalg.norm(a - b) <= 4e-3 * np.linalg.norm(b), k
        gate = res[0][keys[i + 2]]
        assert not np.any(res[0][keys[i + 1]][~gate]) and np.count_nonzero(res[0][keys[i + 1]]) > gate.sum() // 2


def test_e1_weight_gradient_wave_pipeline(lib_built, tmp_path):
    """e1's weight gradient at a batch where the per-wave pipeline runs (wgrad_e1.hip: from 512 strips per launch) against the fp64 gradient
    of the same bf16 operands (conv2d_same + autograd on the CPU) and against the tile kernel (SV_NO_WGRAD_E1=1): dW, dbias; an odd batch
    (257: ragged last workgroup) and the two-network launch shape are covered by the step tests."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "g = torch.Generator().manual_seed(21)\n"
        "outs = []\n"
        "for B in (256, 257):\n"
        "    x = torch.randn(B, 64, 64, 3, generator=g).bfloat16()\n"
        "    dy = torch.randn(B, 32, 32, 32, generator=g).bfloat16()\n"
        "    c = ops.Conv2D(B, 64, 64, 3, 32, 6, 2, act='relu', dtype=torch.bfloat16); c.prep((torch.randn(6, 6, 3, 32, generator=g) * 0.05).cuda())\n"
        "    xp = torch.zeros(B, 64, 64, c.desc.ldx, dtype=torch.bfloat16); xp[..., :3] = x\n"
        "    dw, db = c.wgrad(xp.cuda(), dy.cuda(), workspace=True)\n"
        "    outs += [dw.float().cpu().numpy(), db.float().cpu().numpy()]\n"
        "    if len(sys.argv) > 2:\n"
        "        import torch.nn.functional as F\n"
        "        w = torch.zeros(32, 3, 6, 6, dtype=torch.float64, requires_grad=True)\n"
        "        xr = F.pad(x.double().permute(0, 3, 1, 2), (2, 2, 2, 2))\n"
        "        y = F.conv2d(xr, w, stride=2)\n"
        "        y.backward(dy.double().permute(0, 3, 1, 2))\n"
        "        outs += [w.grad.permute(2, 3, 1, 0).numpy(), dy.double().sum((0, 1, 2)).numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env, extra in (("wave", {}, ["ref"]), ("tile", {"SV_NO_WGRAD_E1": "1"}, [])):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out] + extra, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    wave, tile = res
    wk, tk = wave.files, tile.files
    for i in range(2):                       # per batch: wave = [dw, db, ref_dw, ref_db], tile = [dw, db]
        dw, db, rdw, rdb = (wave[wk[4 * i + k]].astype(np.float64) for k in range(4))
        tdw, tdb = (tile[tk[2 * i + k]].astype(np.float64) for k in range(2))
        assert np.linalg.norm(dw - rdw) <= 2e-3 * np.linalg.norm(rdw) and np.abs(dw - rdw).max() <= 1e-2 * np.abs(rdw).max()
        assert np.linalg.norm(db - rdb) <= 2e-3 * np.linalg.norm(rdb)
        assert np.linalg.norm(dw - tdw) <= 2e-3 * np.linalg.norm(tdw) and np.linalg.norm(db - tdb) <= 2e-3 * np.linalg.norm(tdb)


def test_e2_weight_gradient_rolling_window(lib_built, tmp_path):
    """e2's weight gradient on wgrad_e2.hip (forced on at small batches with SV_WGRAD_E2_MIN=1; in the step it runs from 512 images per
    launch) against the fp64 gradient of the same bf16 operands and against the tile kernel (SV_NO_WGRAD_E2=1): B = 5 (one image per
    workgroup) and B = 300 (two per workgroup, a ragged last one)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "import torch.nn.functional as F\n"
        "g = torch.Generator().manual_seed(22)\n"
        "outs = []\n"
        "for B in (5, 300):\n"
        "    x = torch.randn(B, 32, 32, 32, generator=g).bfloat16()\n"
        "    dy = torch.randn(B, 16, 16, 64, generator=g).bfloat16()\n"
        "    c = ops.Conv2D(B, 32, 32, 32, 64, 6, 2, act='relu', dtype=torch.bfloat16); c.prep((torch.randn(6, 6, 32, 64, generator=g) * 0.05).cuda())\n"
        "    dw, db = c.wgrad(x.cuda(), dy.cuda(), workspace=True)\n"
        "    outs += [dw.float().cpu().numpy(), db.float().cpu().numpy()]\n"
        "    if len(sys.argv) > 2:\n"
        "        w = torch.zeros(64, 32, 6, 6, dtype=torch.float64, requires_grad=True)\n"
        "        y = F.conv2d(F.pad(x.double().permute(0, 3, 1, 2), (2, 2, 2, 2)), w, stride=2)\n"
        "        y.backward(dy.double().permute(0, 3, 1, 2))\n"
        "        outs += [w.grad.permute(2, 3, 1, 0).numpy(), dy.double().sum((0, 1, 2)).numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env, extra in (("roll", {"SV_WGRAD_E2_MIN": "1"}, ["ref"]), ("tile", {"SV_NO_WGRAD_E2": "1"}, [])):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out] + extra, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    roll, tile = res
    rk, tk = roll.files, tile.files
    for i in range(2):
        dw, db, rdw, rdb = (roll[rk[4 * i + k]].astype(np.float64) for k in range(4))
        tdw, tdb = (tile[tk[2 * i + k]].astype(np.float64) for k in range(2))
        assert np.linalg.norm(dw - rdw) <= 2e-3 * np.linalg.norm(rdw) and np.abs(dw - rdw).max() <= 1e-2 * np.abs(rdw).max()
        assert np.linalg.norm(db - rdb) <= 2e-3 * np.linalg.norm(rdb)
        assert np.linalg.norm(dw - tdw) <= 2e-3 * np.linalg.norm(tdw) and np.linalg.norm(db - tdb) <= 2e-3 * np.linalg.norm(tdb)


ADJ_LAYERS = [  # name, H (hi-res conv input = output size), Cin, Cout, k, y_f32
    ("d3_64", 16, 128, 64, 4, False),
    ("d4_64", 32, 64, 32, 6, False),
    ("d5_64", 64, 32, 6, 6, True),
]


@pytest.mark.parametrize("B", [2, 37])
@pytest.mark.parametrize("layer", ADJ_LAYERS, ids=[l[0] for l in ADJ_LAYERS])
def test_input_gradient_fused_with_resize_adjoint(ops, layer, B):
    """sv_conv2d_nhwc_dgrad_lowres (Conv2DBackpropInput + ResizeBilinearGrad + ReluGrad in one launch, the hi-res gradient
    kept in LDS) against (a) the two-launch form dgrad -> upsample2x_bwd: BITWISE (same bf16-rounded hi-res values, same
    adjoint arithmetic), (b) fp64 autograd through relu -> tf.image.resize -> conv of the oracle restatement."""
    name, H, Cin, Cout, k, yf32 = layer
    rng = np.random.default_rng(sum(map(ord, name)) + 7 * B)
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.bfloat16, y_f32=yf32, ups_in=True)
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    conv.prep(w.cuda())
    gdy = (Cout + 7) // 8 * 8
    dy = torch.zeros(B, H, H, gdy)
    dy[..., :Cout] = torch.from_numpy(rng.standard_normal((B, H, H, Cout)).astype(np.float32))
    dy = dy.bfloat16()
    pre_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).bfloat16()   # pre-activation sign pattern
    act_lo = torch.relu(pre_lo)                                    # the low-res activation the mask is read from
    fused = conv.dgrad_lowres(dy.cuda(), act_lo.cuda())
    assert fused is not None, "no fused kernel for %s" % name
    two = ops.upsample2x_bwd(conv.dgrad(dy.cuda()), act_lo.cuda())

    def same(a, b_):
        if name not in ("d4_64", "d3_64"):
            return torch.equal(a, b_)
        # d3 / d4 (round 4): the adjoint runs on the matrix pipe, chained from the accumulators (row_conv.hip RowCfg::MA) -- the same bf16-rounded
        # hi-res values and exact products, but the fp32 sums run in another order than adj2x_row_bf16's: equal to a bf16 ulp, zeros (the mask) exactly
        a, b_ = a.float(), b_.float()
        return bool(((a == 0) == (b_ == 0)).all()) and float((a - b_).abs().max()) <= 2.0 ** -7 * float(b_.abs().max()) and \
            float((a != b_).float().mean()) < 0.2 and float((a - b_).norm() / b_.norm()) < 1e-3
    assert same(fused, two)
    # fp64: d/d(pre_lo) of <conv(resize(relu(pre_lo))), dy>
    xr = pre_lo.double().requires_grad_(True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(torch.relu(xr)), w.bfloat16().double(), None, 1, None)
    (y * dy[..., :Cout].double()).sum().backward()
    want = xr.grad
    got = fused[..., :Cin].double().cpu()
    assert float((got - want).norm() / want.norm()) < 6e-3                       # bf16 rounding of the hi-res gradient and of the result
    torch.testing.assert_close(got, want, rtol=BF16_RTOL, atol=1.5e-2 * float(want.abs().max()))
    assert bool((got[act_lo.double() <= 0] == 0).all())                            # the mask, exactly
    nomask = conv.dgrad_lowres(dy.cuda(), None)
    assert same(nomask, ops.upsample2x_bwd(conv.dgrad(dy.cuda()), None))


def test_d5_input_gradient_on_the_row_ring_kernel(ops):
    """8-channel pixels (the head's gradient): four taps per MFMA K step.  Against fp64 and, in a second process, against the
    LDS-tile kernel."""
    rng = np.random.default_rng(11)
    B, H, Cin, Cout, k = 5, 64, 32, 6, 6
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    conv.prep(w.cuda())
    dy = torch.zeros(B, H, H, 8)
    dy[..., :Cout] = torch.from_numpy(rng.standard_normal((B, H, H, Cout)).astype(np.float32))
    dy = dy.bfloat16()
    dx = conv.dgrad(dy.cuda())
    xr = torch.zeros(B, H, H, Cin, dtype=torch.float64, requires_grad=True)
    y = torch_ref.conv2d_same(xr, w.bfloat16().double(), None, 1, None)
    (y * dy[..., :Cout].double()).sum().backward()
    assert float((dx.double().cpu() - xr.grad).norm() / xr.grad.norm()) < 4e-3
    torch.testing.assert_close(dx.double().cpu(), xr.grad, rtol=BF16_RTOL, atol=1e-2 * float(xr.grad.abs().max()))


def test_d5_input_gradient_ignores_stale_lds(ops):
    """The tap-packed form reads two DUMMY taps (zero weights) from rows behind its window: they must never meet non-finite
    LDS leftovers (0 x NaN).  Interleave kernels that leave fp32 bit patterns in LDS; the result stays finite and bitwise stable."""
    g = torch.Generator().manual_seed(3)
    B, H = 64, 64
    conv = ops.Conv2D(B, H, H, 32, 6, 6, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
    conv.prep((torch.randn(6, 6, 32, 6, generator=g) * 0.05).cuda())
    dy = torch.zeros(B, H, H, 8)
    dy[..., :6] = torch.randn(B, H, H, 6, generator=g)
    dy = dy.bfloat16().cuda()
    first = conv.dgrad(dy).clone()
    assert torch.isfinite(first.float()).all()
    f32conv = ops.Conv2D(8, 32, 32, 32, 64, 6, 2, act="relu", dtype=torch.float32)
    f32conv.prep(torch.full((6, 6, 32, 64), float("nan"), device="cuda"))
    xin = torch.full((8, 32, 32, 32), float("inf"), device="cuda")
    for _ in range(6):
        f32conv.fwd(xin, torch.zeros(64, device="cuda"))           # NaN / Inf operands through LDS on every CU
        again = conv.dgrad(dy)
        assert torch.equal(again, first)


# 128: the frame kernel's 75.6 KB of line buffers (> the 64 KB default cap); B = 130 / 257 at 64 x 64: the rolling-window main term (wgrad_p5.hip, from 128
# images per launch): 257 images = 514 strips over 129 workgroups of two teams -- the last workgroup's second team has no strip
@pytest.mark.parametrize("H,B", [(32, 2), (32, 9), (64, 2), (64, 9), (128, 2), (64, 130), (64, 257)])
def test_polyphase_weight_gradient_of_the_head(ops, H, B):
    """Conv2DBackpropFilter + BiasAddGrad of the decoder head in polyphase form (poly_wgrad.hip; the algebra is pinned on CPU
    by tests/test_polyphase_math.py) against autograd of the fp64 resize -> conv on the same bf16 operands, and against the
    direct fused-upsample kernel; run-to-run identical weights (fixed-order slabs; the bias uses atomics)."""
    rng = np.random.default_rng(H + B)
    Cin, Cout, k = 32, 6, 6
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).bfloat16()
    dy = torch.from_numpy(rng.standard_normal((B, H, H, 8)).astype(np.float32)).bfloat16()
    dy[..., Cout:] = 0
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.bfloat16, y_f32=True, ups_in=True)
    conv.prep(torch.zeros(k, k, Cin, Cout).cuda())
    dw, db = conv.wgrad_poly(x_lo.cuda(), dy.cuda())
    wt = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(x_lo.double()), wt, bt, 1, None)
    (y * dy[..., :Cout].double()).sum().backward()
    assert float((dw.double().cpu() - wt.grad).norm() / wt.grad.norm()) < 2e-3
    torch.testing.assert_close(db.double().cpu(), bt.grad, rtol=1e-4, atol=1e-4 * float(bt.grad.abs().max()))
    for ky, kx in ((0, 0), (0, 5), (5, 0), (5, 5), (2, 2), (3, 1)):                 # corner taps carry the largest frame terms
        a, r = dw[ky, kx].double().cpu(), wt.grad[ky, kx]
        assert float((a - r).norm() / r.norm()) < 5e-3, (ky, kx)
    dw0, db0 = conv.wgrad(x_lo.cuda(), dy.cuda(), workspace=True)
    assert float((dw - dw0).norm() / dw0.norm()) < 6e-3
    dw2, _ = conv.wgrad_poly(x_lo.cuda(), dy.cuda())
    assert torch.equal(dw, dw2)


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [("d2_64", 8, 128, 128, 4, 1), ("e3_64", 16, 64, 128, 4, 2), ("d3_plain", 16, 128, 64, 4, 1)], ids=lambda l: l[0])
def test_weight_gradient_pipeline_for_whole_image_tiles(ops, layer, monkeypatch):
    """wgrad_tile_pipe_kernel (wgrad_tile.hip): the tile weight gradient of the layers whose tiles are whole images as a DMA-fed pipeline -- taken
    from 64 tiles per problem, so the 3-image cases of test_conv_fwd_dgrad_wgrad never reach it.  Against autograd of the fp64 conv on the same
    bf16 operands, against the two-group form of the same build (SV_WT_NO_PIPE is read once per process: a subprocess would be needed to flip it,
    so the comparison is with the fp64 reference and with a second run), bias gradient included; B = 129 leaves the last two-image tile of d2 / e3 half full
    (the pipelined form needs whole tiles: that launch falls back) and exercises a ragged tile run per workgroup."""
    name, H, Cin, Cout, k, s = layer
    for B in (128, 129):
        rng = np.random.default_rng(B + H)
        x = torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32)).bfloat16()
        OH = H // s
        dy = torch.from_numpy(rng.standard_normal((B, OH, OH, Cout)).astype(np.float32)).bfloat16()
        conv = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=None, dtype=torch.bfloat16, y_f32=False)
        conv.prep(torch.zeros(k, k, Cin, Cout).cuda())
        dw, db = conv.wgrad(x.cuda(), dy.cuda(), workspace=True)
        wt = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
        bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        y = torch_ref.conv2d_same(x.double(), wt, bt, s, None)
        (y * dy.double()).sum().backward()
        torch.testing.assert_close(dw.double().cpu(), wt.grad, rtol=BF16_RTOL, atol=1e-4 * float(wt.grad.abs().max()))
        torch.testing.assert_close(db.double().cpu(), bt.grad, rtol=1e-4, atol=1e-4 * float(bt.grad.abs().max()))
        dw2, db2 = conv.wgrad(x.cuda(), dy.cuda(), workspace=True)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)          # fixed-order slabs: run-to-run identical


@pytest.mark.gpu
@pytest.mark.parametrize("layer", [l for l in LAYERS if l[0] in ("d3", "d4", "d4_64", "d5", "d5_64")], ids=lambda l: l[0])
def test_fp32_fused_upsample_forward_and_weight_gradient(ops, layer):
    """The fp32 step fuses the bilinear resizes into the tile staging since round 4 (forward: tile_conv_kernel<float>, weight gradient:
    wgrad_tile_f32_kernel with stage_tile_upsampled<float>, the head in the x-packed form): against the written-out resize + plain conv of the
    same build (the resize itself is bitwise upsample2x_fwd's) and against the fp64 composition; with and without a workspace (fixed-order
    slabs / fp32 atomics -- the im2col kernel cannot take a fused resize)."""
    name, H, Cin, Cout, k, s, act, yf32 = layer
    rng = np.random.default_rng(sum(map(ord, name)) + 7)
    B = 3
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)).cuda() * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)).cuda() * 0.1
    plain = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=act, dtype=torch.float32, y_f32=yf32)
    fused = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=act, dtype=torch.float32, y_f32=yf32, ups_in=True)
    plain.prep(w); fused.prep(w)
    x_hi = ops.upsample2x_fwd(x_lo)
    y0 = plain.fwd(x_hi, b)
    y1 = fused.fwd(x_lo, b)
    torch.testing.assert_close(y1[..., :Cout], y0[..., :Cout], rtol=F32_RTOL, atol=F32_ATOL * float(y0.abs().max()))
    yr = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(x_lo.cpu().double()), w.cpu().double(), b.cpu().double(), s, act)
    torch.testing.assert_close(y1[..., :Cout].double().cpu(), yr, rtol=F32_RTOL, atol=F32_ATOL * float(yr.abs().max()))
    dy = torch.from_numpy(rng.standard_normal((B, H, H, (Cout + 7) // 8 * 8)).astype(np.float32)).cuda()
    if Cout % 8:
        dy[..., Cout:] = 0
    dw0, db0 = plain.wgrad(x_hi, dy, workspace=True)
    dw1, db1 = fused.wgrad(x_lo, dy, workspace=True)
    dw2, db2 = fused.wgrad(x_lo, dy)                       # no workspace: the tile kernel's atomics flush
    for dw, db in ((dw1, db1), (dw2, db2)):
        torch.testing.assert_close(dw, dw0, rtol=F32_RTOL, atol=F32_ATOL * float(dw0.abs().max()))
        torch.testing.assert_close(db, db0, rtol=F32_RTOL, atol=F32_ATOL * float(db0.abs().max()))
    dw3, _ = fused.wgrad(x_lo, dy, workspace=True)
    assert torch.equal(dw1, dw3)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 5])
@pytest.mark.parametrize("layer", [("d3_64", 16, 128, 64, 4, "relu", False), ("d4", 16, 64, 32, 6, "relu", False), ("d4_64", 32, 64, 32, 6, "relu", False),
                                   ("d4_128", 64, 64, 32, 6, "relu", False), ("d5", 32, 32, 6, 6, None, True), ("d5_64", 64, 32, 6, 6, None, True)], ids=lambda l: l[0])
def test_fp32_polyphase_forward_against_fp64(ops, layer, B, monkeypatch):
    """UpSampling2D(bilinear) -> Conv2D(padding='same') (vae/model.py:154-156,:163-167) at the reference's precision in POLYPHASE form: the head as one 5 x 5 conv over
    the low-res tensor (svg_poly), d4 / d3 as four per-parity-class convs (svg_polyc; 81 of 144 / 49 of 64 tap products), border rows / columns
    corrected by poly_fix.hip.  Against the fp64 composition resize -> zero-padded conv from the same fp32 operands: every pixel at the fp32 bar, the
    border ring (whose taps leave the image) on its own, bias and ReLU included; bitwise run to run."""
    name, H, Cin, Cout, k, act, yf32 = layer
    if k == 4:
        monkeypatch.setenv("SV_POLYC_K", "64")               # k = 4 (d3) keeps the direct form by default (no gain measured): the form itself is still pinned here
    rng = np.random.default_rng(H * 100 + Cin + B)
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=act, dtype=torch.float32, y_f32=yf32, ups_in=True)
    conv.prep(w.cuda())
    y = conv.fwd(x_lo.cuda(), b.cuda())
    ref = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(x_lo.double()), w.double(), b.double(), 1, act)
    got = y[..., :Cout].double().cpu()
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    torch.testing.assert_close(got, ref, rtol=F32_RTOL, atol=F32_ATOL * scale)
    ring = torch.ones(H, H, dtype=torch.bool)
    ring[3:H - 3, 3:H - 3] = False
    torch.testing.assert_close(got[:, ring], ref[:, ring], rtol=F32_RTOL, atol=F32_ATOL * scale)
    y2 = conv.fwd(x_lo.cuda(), b.cuda())
    assert torch.equal(y, y2)
    # the plain entry point (sv_conv2d_nhwc_fwd: no workspace for the border terms) must serve the same descriptor: the per-class layers fall back to the
    # direct fused-resize form on the second weight image sv_conv2d_prep_weights keeps (include/splitvae.h), the head adds its border terms with atomics
    y3 = conv.fwd(x_lo.cuda(), b.cuda(), workspace=False)
    torch.testing.assert_close(y3[..., :Cout].double().cpu(), ref, rtol=F32_RTOL, atol=F32_ATOL * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 5, 40])
@pytest.mark.parametrize("layer", [("d4", 16, 64, 32, 6, False), ("d4_64", 32, 64, 32, 6, False), ("d4_128", 64, 64, 32, 6, False),
                                   ("d5", 32, 32, 6, 6, True), ("d5_64", 64, 32, 6, 6, True)], ids=lambda l: l[0])
def test_fp32_polyphase_weight_gradient_against_fp64(ops, layer, B):
    """Conv2DBackpropFilter + BiasAddGrad of UpSampling2D(bilinear) -> Conv2D (vae/model.py:154-156,:163-167) at the reference's precision in POLYPHASE form
    (polyc_wgrad.hip: d4 per parity class -- 25 / 20 / 20 / 16 taps on the low-res grid --, the head as one 25-tap problem over the space-to-depth view of dY;
    frame term for the taps that leave the image; projection onto the 6 x 6 kernel) against the fp64 autograd of resize -> conv from the same fp32 operands;
    accumulated INTO dw / db; bitwise run to run (fixed-order slabs and frame groups)."""
    name, H, Cin, Cout, k, yf32 = layer
    rng = np.random.default_rng(H * 7 + Cin + B)
    x_lo = torch.from_numpy(rng.standard_normal((B, H // 2, H // 2, Cin)).astype(np.float32))
    c8 = (Cout + 7) // 8 * 8
    dy = torch.from_numpy(rng.standard_normal((B, H, H, c8)).astype(np.float32))
    dy[..., Cout:] = 0
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.float32, y_f32=yf32, ups_in=True)
    dw, db = conv.wgrad(x_lo.cuda(), dy.cuda(), workspace=True)
    wt = torch.zeros(k, k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(x_lo.double()), wt, bt, 1, None)
    (y * dy[..., :Cout].double()).sum().backward()
    torch.testing.assert_close(dw.double().cpu(), wt.grad, rtol=F32_RTOL, atol=F32_ATOL * float(wt.grad.abs().max()))
    torch.testing.assert_close(db.double().cpu(), bt.grad, rtol=F32_RTOL, atol=F32_ATOL * float(bt.grad.abs().max()))
    # every tap of the kernel on its own (the border taps carry the frame term)
    for ky in range(k):
        for kx in range(k):
            torch.testing.assert_close(dw[ky, kx].double().cpu(), wt.grad[ky, kx], rtol=F32_RTOL, atol=F32_ATOL * float(wt.grad.abs().max()))
    dw2, db2 = conv.wgrad(x_lo.cuda(), dy.cuda(), workspace=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    # accumulation: a second call into the same buffers doubles them
    conv.wgrad(x_lo.cuda(), dy.cuda(), workspace=True, dw=dw2, db=db2)
    torch.testing.assert_close(dw2, 2 * dw, rtol=1e-5, atol=0)
    torch.testing.assert_close(db2, 2 * db, rtol=1e-5, atol=0)


@pytest.mark.gpu
def test_fp32_class_pairs_and_four_classes_per_launch_equal_the_class_launches(lib_built, tmp_path):
    """d4's polyphase weight gradient (vae/model.py:165, Conv2DBackpropFilter behind the resize) with the input tile staged once per class PAIR
    (SV_WGRAD_POLYC_FUSED=2: classes {0, 3} and {1, 2}) and once for all FOUR classes (=1) against the default of one launch per class, in separate
    processes (the knob is latched at first use): same slabs, same reduce, same projection -- only which launch leaves a class's slab differs, so the
    weight gradients agree to fp32 addition order of the bias partials; the opt-in forms are also bitwise run to run.  32 / 64-pixel outputs, a ragged batch."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from split_vae_amd import ops\n"
        "g = torch.Generator().manual_seed(11)\n"
        "outs = []\n"
        "for (H, Cin, Cout, k, B) in ((32, 64, 32, 6, 5), (64, 64, 32, 6, 19), (16, 64, 32, 6, 40)):\n"
        "    x = torch.randn(B, H // 2, H // 2, Cin, generator=g).cuda()\n"
        "    dy = torch.randn(B, H, H, Cout, generator=g).cuda()\n"
        "    c = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.float32, ups_in=True)\n"
        "    dw, db = c.wgrad(x, dy, workspace=True)\n"
        "    dw2, db2 = c.wgrad(x, dy, workspace=True)\n"
        "    assert torch.equal(dw, dw2) and torch.equal(db, db2)\n"
        "    outs += [dw.cpu().numpy(), db.cpu().numpy()]\n"
        "np.savez(sys.argv[1], *outs)\n" % root)
    res = []
    for tag, env in (("classes", {"SV_WGRAD_POLYC_FUSED": "0"}), ("pairs", {"SV_WGRAD_POLYC_FUSED": "2"}), ("four", {"SV_WGRAD_POLYC_FUSED": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    for other in res[1:]:
        for key in res[0].files:
            a, b = other[key].astype(np.float64), res[0][key].astype(np.float64)
            assert np.abs(b).max() > 0
            assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), key


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 5, 19])
@pytest.mark.parametrize("layer", [("d4_64", 32, 64, 32, 6, False), ("d4_128", 64, 64, 32, 6, False), ("d5", 32, 32, 6, 6, True), ("d5_64", 64, 32, 6, 6, True)],
                         ids=lambda l: l[0])
def test_fp32_polyphase_input_gradient_against_fp64(ops, layer, B):
    """Conv2DBackpropInput + ResizeBilinearGrad + ReluGrad of UpSampling2D(bilinear) -> Conv2D (vae/model.py:155-156 behind :165 / :167) at the reference's precision
    as ONE stride-2 conv with 9 x 9 taps over the hi-res dy (polyd_dgrad.hip), edge rows / columns and corners corrected through the workspace, delivered at the
    LOW-RES tensor with the ReLU gate: against fp64 autograd of resize -> conv from the same fp32 operands -- all pixels, the border ring and the four corners on
    their own -- and bitwise run to run."""
    name, H, Cin, Cout, k, yf32 = layer
    rng = np.random.default_rng(H * 13 + Cin + B)
    h = H // 2
    c8 = (Cout + 7) // 8 * 8
    x_lo = torch.from_numpy(rng.standard_normal((B, h, h, Cin)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    dy = torch.from_numpy(rng.standard_normal((B, H, H, c8)).astype(np.float32))
    dy[..., Cout:] = 0
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 1, act=None, dtype=torch.float32, y_f32=yf32, ups_in=True)
    conv.prep(w.cuda())
    dx = conv.dgrad_lowres(dy.cuda(), relu_mask_lo=x_lo.cuda())
    assert dx is not None
    xt = x_lo.double().requires_grad_(True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(xt), w.double(), torch.zeros(Cout, dtype=torch.float64), 1, None)
    (y * dy[..., :Cout].double()).sum().backward()
    want = xt.grad * (x_lo.double() > 0)
    got = dx[..., :Cin].double().cpu()
    scale = float(xt.grad.abs().max())
    torch.testing.assert_close(got, want, rtol=F32_RTOL, atol=F32_ATOL * scale)
    ring = torch.ones(h, h, dtype=torch.bool)
    ring[1:h - 1, 1:h - 1] = False
    torch.testing.assert_close(got[:, ring], want[:, ring], rtol=F32_RTOL, atol=F32_ATOL * scale)
    for i in (0, h - 1):
        for j in (0, h - 1):
            torch.testing.assert_close(got[:, i, j], want[:, i, j], rtol=F32_RTOL, atol=F32_ATOL * scale)
    # without the gate: the raw gradient
    dx_raw = conv.dgrad_lowres(dy.cuda())
    torch.testing.assert_close(dx_raw[..., :Cin].double().cpu(), xt.grad, rtol=F32_RTOL, atol=F32_ATOL * scale)
    dx2 = conv.dgrad_lowres(dy.cuda(), relu_mask_lo=x_lo.cuda())
    assert torch.equal(dx, dx2)
    # the two-launch form of the same build (conv-transpose at hi-res, then the stand-alone resize adjoint)
    two = ops.upsample2x_bwd(conv.dgrad(dy.cuda()), x_lo.cuda())
    torch.testing.assert_close(dx[..., :Cin], two[..., :Cin], rtol=F32_RTOL, atol=F32_ATOL * scale)


SPAIR_OBJECT_LAYERS = [  # name, H, Cin, Cout, k, stride: the 3 x 3 layers of LG-SPAIR's object encoder / decoder on 32 x 32 glimpses
    ("obj_conv1", 32, 3, 32, 3, 2),      # RGB padded to 8 channels: fragment rows 8..15 are dropped (no tap pairs at an odd kernel width)
    ("obj_conv2", 16, 32, 64, 3, 2),
    ("obj_d2", 8, 32, 64, 3, 1),
    ("obj_d3", 16, 64, 32, 3, 1),
    ("obj_d5", 32, 32, 4, 3, 1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("B", [5, 64])
@pytest.mark.parametrize("layer", SPAIR_OBJECT_LAYERS, ids=lambda l: l[0])
def test_fp32_three_by_three_weight_gradient_on_lds_tiles(ops, layer, B):
    """wgrad_tile_f32_kernel<3, ...>: nine taps on twelve slots (three per wave; the fixed-order reduce drops the padding).  Against the fp64
    gradient of Conv2D(padding='same') from the same operands, against the im2col kernel of the same build (no workspace), bitwise run to run;
    B = 5: a ragged last image group."""
    name, H, Cin, Cout, k, s = layer
    rng = np.random.default_rng(sum(map(ord, name)) + B)
    x = torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32))
    OH = H // s
    cp = (Cout + 7) // 8 * 8
    dy = torch.zeros((B, OH, OH, cp))
    dy[..., :Cout] = torch.from_numpy(rng.standard_normal((B, OH, OH, Cout)).astype(np.float32))
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=None, dtype=torch.float32)
    wr = torch.zeros((k, k, Cin, Cout), dtype=torch.float64, requires_grad=True)
    br = torch.zeros((Cout,), dtype=torch.float64, requires_grad=True)
    torch_ref.conv2d_same(x.double(), wr, br, s, None).backward(dy[..., :Cout].double())
    xg = _pad_c(x, conv.desc.ldx).cuda()
    dw_t, db_t = conv.wgrad(xg, dy.cuda(), workspace=True)                # LDS tiles + slabs
    dw_i, db_i = conv.wgrad(xg, dy.cuda())                                # im2col + atomics
    for dw, db in ((dw_t, db_t), (dw_i, db_i)):
        torch.testing.assert_close(dw.double().cpu(), wr.grad, rtol=F32_RTOL, atol=F32_ATOL * float(wr.grad.abs().max()))
        torch.testing.assert_close(db.double().cpu(), br.grad, rtol=F32_RTOL, atol=F32_ATOL * float(br.grad.abs().max()))
    dw_u, db_u = conv.wgrad(xg, dy.cuda(), workspace=True)
    assert torch.equal(dw_t, dw_u) and torch.equal(db_t, db_u)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [("gm_c3", 8, 128, 128, 4, 64), ("gm_c2", 16, 128, 128, 6, 16), ("gm_c3_b5", 8, 128, 128, 4, 5)], ids=lambda g: g[0])
def test_fp32_conv_forward_with_k_split_over_workgroups(ops, geom, monkeypatch):
    """SPLIT-GMVAE's 128 -> 128 stride-2 encoder layers (vae/model.py:49-52, no activation in the descriptor: ELU follows) at small grids: the forward's K is split
    over workgroups on the im2col GEMM (conv_api.hip: SV_CONV_SPLITK_TILES), partial sums added into the zeroed fp32 output, the bias on slice 0.  Against the fp64
    conv from the same fp32 operands, and against the tile kernel (SV_CONV_SPLITK_TILES=0 is read once per process: the comparison runs through SV_DETERMINISTIC,
    which keeps one workgroup per output tile)."""
    name, H, Cin, Cout, k, B = geom
    rng = np.random.default_rng(7 + H + B)
    x = torch.from_numpy(rng.standard_normal((B, H, H, Cin)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, 2, act=None, dtype=torch.float32)
    conv.prep(w.cuda())
    y = conv.fwd(x.cuda(), b.cuda())
    ref = torch_ref.conv2d_same(x.double(), w.double(), b.double(), 2, None)
    scale = float(ref.abs().max())
    torch.testing.assert_close(y[..., :Cout].double().cpu(), ref, rtol=F32_RTOL, atol=F32_ATOL * scale)
    from split_vae_amd import _lib
    lib = _lib.load()
    assert lib.sv_set_deterministic(1) == 0
    try:
        y_det = conv.fwd(x.cuda(), b.cuda())                   # fixed-order mode: the tile kernel (no K split)
    finally:
        assert lib.sv_set_deterministic(-1) == 0
    torch.testing.assert_close(y, y_det, rtol=1e-5, atol=1e-5 * scale)
