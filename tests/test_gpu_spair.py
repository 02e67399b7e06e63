"""SPLIT-SPAIR, first blocker: the conv backbone of spair.Encoder (spair/spair.py:382-388) -- 48 -> 24 ->
12 -> 4x4 cells with strides 2, 2, 3 and three 1x1 convs, the only non-power-of-two extents and the only stride 3 of the repo --
on the MFMA im2col kernels (divide-based row decode), at the reference's shape [32, 48, 48, 3] (batch 32 is hard-coded,
spair/trainer.py:346), against the fp64 oracle restatement: forward chain, and per layer the input and weight gradients."""
import math

import numpy as np
import pytest
import torch

from oracle import spair_ref, torch_ref

pytestmark = pytest.mark.gpu
B = 32


@pytest.fixture(scope="module")
def ops(lib_built):
    assert torch.cuda.is_available()
    from split_vae_amd import ops as o
    return o


def _pad8(t):
    c = t.shape[-1]
    return torch.nn.functional.pad(t, (0, (c + 7) // 8 * 8 - c))


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_backbone_forward_chain(ops, dtype, rtol):
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.integers(0, 256, (B, 48, 48, 3)) / 255.0).astype(np.float32))       # spair/data.py: canvases in [0,1]
    params = spair_ref.backbone_init(3)
    for i in range(1, len(params), 2):
        params[i] = (rng.standard_normal(params[i].shape) * 0.05).astype(np.float32)             # exercise the bias path
    # fp64 reference from the operands as the device sees them (bf16 path: weights and every activation rounded to bf16)
    rnd = (lambda t: t.to(dtype).double())
    h_ref, refs = rnd(x), []
    h = _pad8(x).to(dtype).cuda()
    H = 48
    for i, (name, k, s, ci, co) in enumerate(spair_ref.BACKBONE):
        w, b = torch.from_numpy(params[2 * i]), torch.from_numpy(params[2 * i + 1])
        conv = ops.Conv2D(B, H, H, ci, co, k, s, act="relu", dtype=dtype)
        conv.prep(w.cuda())
        h = conv.fwd(h.contiguous(), b.cuda())
        h_ref = rnd(torch_ref.conv2d_same(h_ref, rnd(w), b.double(), s, "relu"))
        H = (H + s - 1) // s
        assert tuple(h.shape) == (B, H, H, (co + 7) // 8 * 8)
        got = h[..., :co].double().cpu()
        torch.testing.assert_close(got, h_ref, rtol=rtol, atol=rtol * float(h_ref.abs().max()), msg=lambda m: name + ": " + m)
        if co % 8:
            assert float(h[..., co:].abs().max()) == 0.0                                          # pad channels stay zero
        h_ref = got if dtype == torch.bfloat16 else h_ref                                         # bf16: continue from the device's rounding
    assert H == 4


@pytest.mark.parametrize("layer", [l for l in spair_ref.BACKBONE if l[0] != "z3"], ids=lambda l: l[0])
def test_backbone_layer_gradients(ops, layer):
    """Input gradient (stride 3: nine parity classes with 1, 2 or 4 taps) and weight gradient, bf16 operands, fp64 reference."""
    name, k, s, ci, co = layer
    H = {"conv1": 48, "conv2": 24, "conv3": 12}.get(name, 4)
    rng = np.random.default_rng(sum(map(ord, name)))
    x = torch.from_numpy(rng.standard_normal((B, H, H, ci)).astype(np.float32)).bfloat16()
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, ci, co)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (ci + co)))
    OH = (H + s - 1) // s
    dy = torch.from_numpy(rng.standard_normal((B, OH, OH, co)).astype(np.float32)).bfloat16()
    conv = ops.Conv2D(B, H, H, ci, co, k, s, act="relu", dtype=torch.bfloat16)
    conv.prep(w.cuda())
    xr = x.double().requires_grad_(True)
    wr = w.bfloat16().double().requires_grad_(True)
    br = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    torch_ref.conv2d_same(xr, wr, br, s, None).backward(dy.double())
    dw, db = conv.wgrad(_pad8(x).cuda(), dy.cuda())
    torch.testing.assert_close(dw.double().cpu(), wr.grad, rtol=3e-2, atol=3e-2 * float(wr.grad.abs().max()))
    torch.testing.assert_close(db.double().cpu(), br.grad, rtol=3e-2, atol=3e-2 * float(br.grad.abs().max()))
    if name == "conv1":
        return                                                 # the canvas needs no gradient
    dx = conv.dgrad(dy.cuda())
    assert float((dx[..., :ci].double().cpu() - xr.grad).norm() / xr.grad.norm()) < 5e-3
    torch.testing.assert_close(dx[..., :ci].double().cpu(), xr.grad, rtol=3e-2, atol=1.5e-2 * float(xr.grad.abs().max()))


def test_unsupported_geometries_are_refused(ops):
    from split_vae_amd import _lib
    with pytest.raises(_lib.SplitVaeError):
        ops.Conv2D(4, 50, 50, 8, 8, 4, 3)                       # stride must divide the extent
    with pytest.raises(_lib.SplitVaeError):
        ops.Conv2D(4, 48, 48, 8, 8, 4, 4)                       # strides 1..3


# ------------------------------------------------------------------ spatial transformer (spair/utils.py:119-330)
STN_CASES = [  # name, inverse, image extent, channels, output extent
    ("glimpses", False, 48, 3, 32),      # STN(H_img=32...) on the [32,48,48,3] canvas: one 32x32 glimpse per cell
    ("render", True, 32, 4, 48),         # the renderer's inverse STN: each object's rgb+alpha pasted on a 48x48 canvas
]


def _stn_inputs(inverse, Hi, C, seed):
    rng = np.random.default_rng(seed)
    Bs, Hc = 5, 4
    shape = (Bs, Hc * Hc, Hi, Hi, C) if inverse else (Bs, Hi, Hi, C)
    img = torch.from_numpy(rng.uniform(0, 1, shape).astype(np.float32))
    z = torch.from_numpy((rng.standard_normal((Bs, Hc, Hc, 4)) * 1.5).astype(np.float32))
    return img, z


@pytest.mark.parametrize("case", STN_CASES, ids=lambda c: c[0])
def test_stn_forward_matches_oracle(ops, case):
    """Affine grid + 4-tap gather + obj_bbox_mask against the restatement, both STN directions.  fp32 on the device vs fp64:
    the output is continuous in the sampling point, so a floor() that lands on the other side of an integer in fp32 moves the
    value by O(eps) only."""
    name, inverse, Hi, C, Ho = case
    img, z = _stn_inputs(inverse, Hi, C, 1)
    out, bbox = ops.stn_sample(img.cuda(), z.cuda(), Ho, Ho, inverse=inverse)
    ref, rbox = spair_ref.stn_forward(img.double(), z.double(), Ho, Ho, inverse=inverse)
    assert tuple(out.shape) == tuple(ref.shape)
    # inverse form: sampling coordinates are scaled by 1/sx (up to ~1e5 when a cell's box collapses): fp32 coordinate error
    torch.testing.assert_close(out.double().cpu(), ref, rtol=0, atol=2e-3 if inverse else 2e-5)
    assert float((out.double().cpu() - ref).norm() / ref.norm()) < (2e-4 if inverse else 1e-5)
    torch.testing.assert_close(bbox.double().cpu(), rbox, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", STN_CASES, ids=lambda c: c[0])
def test_stn_backward_matches_autograd_of_the_oracle(ops, case):
    """g_img (scatter of the four tap weights) and g_z_where (through the tap weights; floor / clip carry none) against
    torch autograd of the fp64 restatement."""
    name, inverse, Hi, C, Ho = case
    img, z = _stn_inputs(inverse, Hi, C, 2)
    if inverse:
        z = z * 0.5                                       # keep 1/sx moderate: the comparison is about the chain rule
    rng = np.random.default_rng(3)
    g = torch.from_numpy(rng.standard_normal((img.shape[0], 16, Ho, Ho, C)).astype(np.float32))
    ri, rz = img.double().requires_grad_(True), z.double().requires_grad_(True)
    ref, _ = spair_ref.stn_forward(ri, rz, Ho, Ho, inverse=inverse)
    (ref * g.double()).sum().backward()
    g_img, g_z = ops.stn_sample_bwd(img.cuda(), z.cuda(), g.cuda(), inverse=inverse)
    ei = float((g_img.double().cpu() - ri.grad).norm() / ri.grad.norm())
    ez = float((g_z.double().cpu() - rz.grad).norm() / rz.grad.norm())
    assert ei < (1e-3 if inverse else 1e-5), ei
    assert ez < (2e-2 if inverse else 1e-3), ez          # a sampling point within fp32 eps of an integer takes the other cell's slope
    # run-to-run: g_z_where is reduced in a fixed order
    _, g_z2 = ops.stn_sample_bwd(img.cuda(), z.cuda(), g.cuda(), inverse=inverse)
    assert torch.equal(g_z, g_z2)


# ------------------------------------------------------------------ Renderer (spair/spair.py:534-579)
def _render_inputs(seed, Bs=6, Bp=16, H=48, C=3):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(a.astype(np.float32))
    obj = t(rng.uniform(-0.2, 1.2, (Bs, Bp, H, H, C + 1)))          # rgb and alpha partly outside their clip ranges
    bg = t(rng.uniform(0, 1, (Bs, H, H, C)))
    zd = t(rng.standard_normal((Bs, 4, 4, 1)) * 2)
    zp = t(rng.uniform(0.01, 0.99, (Bs, 4, 4, 1)))
    zl = t(rng.standard_normal((Bs, 4, 4, 1)) * 3)
    noise = t(rng.standard_normal((Bs, Bp, H, H, C)) * 0.01)
    return obj, bg, zd, zp, zl, noise


@pytest.mark.parametrize("mode", ["train", "train_noise", "test"])
def test_renderer_forward_matches_oracle(ops, mode):
    obj, bg, zd, zp, zl, noise = _render_inputs(4)
    training = mode != "test"
    nz = noise if mode == "train_noise" else None
    out = ops.spair_render(obj.cuda(), bg.cuda(), zd.cuda(), z_pres=zp.cuda() if training else None,
                           z_pres_logits=None if training else zl.cuda(), training=training, noise=None if nz is None else nz.cuda())
    ref = spair_ref.renderer(obj.double(), bg.double(), zd.double(), zp.double(), zl.double(), training=training,
                             noise=None if nz is None else nz.double())
    torch.testing.assert_close(out.double().cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("with_noise", [False, True])
def test_renderer_backward_matches_autograd_of_the_oracle(ops, with_noise):
    obj, bg, zd, zp, zl, noise = _render_inputs(5)
    nz = noise if with_noise else None
    rng = np.random.default_rng(6)
    g = torch.from_numpy(rng.standard_normal(tuple(bg.shape)).astype(np.float32))
    r = [t.double().requires_grad_(True) for t in (obj, bg, zd, zp)]
    ref = spair_ref.renderer(r[0], r[1], r[2], r[3], zl.double(), training=True, noise=None if nz is None else nz.double())
    (ref * g.double()).sum().backward()
    got = ops.spair_render_bwd(obj.cuda(), bg.cuda(), zd.cuda(), zp.cuda(), g.cuda(), noise=None if nz is None else nz.cuda())
    want = (r[0].grad, r[1].grad, r[3].grad.reshape(-1, 16), r[2].grad.reshape(-1, 16))
    for name, a, b in zip(("g_obj", "g_bg", "g_z_pres", "g_z_depth"), got, want):
        err = float((a.double().cpu() - b).norm() / b.norm())
        assert err < 1e-4, (name, err)
    again = ops.spair_render_bwd(obj.cuda(), bg.cuda(), zd.cuda(), zp.cuda(), g.cuda(), noise=None if nz is None else nz.cuda())
    assert all(torch.equal(a, b) for a, b in zip(got, again))      # fixed-order reductions: run-to-run identical


# ------------------------------------------------------------------ z_pres KL and clipnorm Adam
@pytest.mark.parametrize("prior_prob,temp", [(0.1, 1.0), (0.5, 2.5), (0.01, 0.5)])
def test_zpres_kl_and_its_gradient(ops, prior_prob, temp):
    """compute_z_pres_kl_yolo_air (spair/trainer.py:45-94) on the device: per-image sums against the fp64 restatement, the
    gradients against its autograd (the count prior sees thresholded samples only: no gradient to z_pres)."""
    rng = np.random.default_rng(int(prior_prob * 1000))
    Bs = 32
    pre = torch.from_numpy(rng.standard_normal((Bs, 4, 4, 1)).astype(np.float32) * 2)
    logits = torch.from_numpy(rng.standard_normal((Bs, 4, 4, 1)).astype(np.float32) * 2)
    zp = torch.sigmoid(pre)
    zp[0] = 0.9; zp[1] = 0.1                               # all on / all off: the count recursion's extremes
    kl, g_pre, g_log = ops.spair_zpres_kl(zp.cuda(), logits.cuda(), pre.cuda(), prior_prob, temp)
    rp, rl = pre.double().requires_grad_(True), logits.double().requires_grad_(True)
    ref = spair_ref.compute_z_pres_kl_yolo_air(zp.double(), rl, rp, prior_prob, temp)
    ref.backward()
    assert abs(float(kl.double().mean()) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-5
    torch.testing.assert_close(g_pre.double().cpu(), rp.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(g_log.double().cpu(), rl.grad, rtol=1e-4, atol=1e-7)


def test_adam_with_clipnorm_matches_keras_formula(ops):
    """Adam(clipnorm=1.0) (spair/main.py:109): per-TENSOR tf.clip_by_norm, then the Keras update; three steps, tensors whose
    norm is above and below the threshold, a ragged table."""
    rng = np.random.default_rng(9)
    sizes = [7, 4096, 33, 100000, 1]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(off[-1])
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    P, M, V = (torch.from_numpy(a.copy()).cuda() for a in (p, m, v))
    OFF = torch.from_numpy(off).cuda()
    lr, b1, b2, eps, clip = 1e-3, 0.9, 0.999, 1e-7, 1.0
    p64, m64, v64 = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32) * np.repeat([0.01, 1.0, 0.2, 0.001, 5.0], sizes).astype(np.float32)
        ops.adam_step_clipnorm(P, torch.from_numpy(g).cuda(), M, V, OFF, clip, t, lr, b1, b2, eps)
        g64 = g.astype(np.float64)
        for a, b in zip(off[:-1], off[1:]):
            nrm = np.sqrt((g64[a:b] ** 2).sum())
            g64[a:b] *= clip / max(nrm, clip)
        m64 += (g64 - m64) * (1 - b1); v64 += (g64 * g64 - v64) * (1 - b2)
        alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        p64 -= alpha * m64 / (np.sqrt(v64) + eps)
    np.testing.assert_allclose(P.cpu().numpy(), p64, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(M.cpu().numpy(), m64, rtol=2e-5, atol=1e-8)


# ------------------------------------------------------------------ ImageEncoder / ImageDecoder geometry (spair/spair.py:113-160)
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("H,Cin,Cout", [(48, 3, 32), (24, 32, 64), (12, 64, 128), (32, 3, 32), (16, 32, 64)])
def test_odd_kernel_stride_2_layers(ops, H, Cin, Cout, dtype, rtol):
    """Conv2D(kernel_size=3, strides=2, padding='same') of SPAIR's ImageEncoder (48 -> 24 -> 12 -> 6; TF pads 0 before / 1 after)
    and the same layer on power-of-two extents: forward, input gradient (parity classes with 4 / 2 / 2 / 1 taps) and weight
    gradient against the fp64 reference."""
    rng = np.random.default_rng(H + Cin)
    Bs, k, s = 8, 3, 2
    x = torch.from_numpy(rng.standard_normal((Bs, H, H, Cin)).astype(np.float32)).to(dtype)
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    b = torch.from_numpy(rng.standard_normal((Cout,)).astype(np.float32)) * 0.1
    conv = ops.Conv2D(Bs, H, H, Cin, Cout, k, s, act="relu", dtype=dtype)
    conv.prep(w.cuda())
    xg = _pad8(x).cuda()
    y = conv.fwd(xg, b.cuda())
    wr, xr, br = w.to(dtype).double().requires_grad_(True), x.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch_ref.conv2d_same(xr, wr, br, s, "relu")
    assert tuple(y.shape[1:3]) == tuple(yr.shape[1:3]) == (H // 2, H // 2)
    torch.testing.assert_close(y[..., :Cout].double().cpu(), yr.detach(), rtol=rtol, atol=rtol * float(yr.abs().max()))
    dy = torch.from_numpy(rng.standard_normal(tuple(yr.shape)).astype(np.float32)).to(dtype)
    pre = torch_ref.conv2d_same(xr, wr, br, s, None)
    pre.backward(dy.double())
    dyg = _pad8(dy).cuda()
    dw, db = conv.wgrad(xg, dyg)
    torch.testing.assert_close(dw.double().cpu(), wr.grad, rtol=rtol, atol=rtol * float(wr.grad.abs().max()))
    torch.testing.assert_close(db.double().cpu(), br.grad, rtol=rtol, atol=rtol * float(br.grad.abs().max()))
    # Cin = 3: the glimpse encoder's first layer (spair/spair.py:250) -- its input gradient reaches z_where through the STN
    dx = conv.dgrad(dyg)
    torch.testing.assert_close(dx[..., :Cin].double().cpu(), xr.grad, rtol=rtol, atol=rtol * float(xr.grad.abs().max()))
    assert float(dx[..., Cin:].abs().max()) == 0.0 if Cin % 8 else True


@pytest.mark.parametrize("mode", ["xent", "kl", "kl_prior"])
def test_loss_reductions_and_their_gradients(ops, mode):
    """sv_spair_loss: xent_loss / kl_divergence / kl_divergence_two_gauss (spair/trainer.py:13-24, :103-109) as per-image sums with
    gradients, against autograd of the fp64 restatement -- including predictions at exactly 0 and 1 (tf_safe_log's 1e-8)."""
    from oracle import spair_model_ref as R
    g = torch.Generator().manual_seed(4)
    B, shape = 5, (6, 7, 3)
    if mode == "xent":
        a = torch.rand(B, *shape, generator=g)
        b = torch.rand(B, *shape, generator=g)
        b.view(-1)[:4] = torch.tensor([0.0, 1.0, 1e-9, 1 - 1e-7])
        f = lambda x, y: R.xent_loss(x, y)
    elif mode == "kl":
        a = torch.randn(B, *shape, generator=g)
        b = torch.nn.functional.softplus(torch.randn(B, *shape, generator=g))
        f = lambda x, y: -0.5 * (1 + spair_ref.tf_safe_log(y * y) - x * x - torch.exp(spair_ref.tf_safe_log(y * y)))
    else:
        a = torch.randn(B, *shape, generator=g)
        b = torch.nn.functional.softplus(torch.randn(B, *shape, generator=g) - 1.0)
        pm, ps = 3.7, 0.5
        f = lambda x, y: (spair_ref.tf_safe_log(torch.full_like(y, ps)) - spair_ref.tf_safe_log(y) + (y * y + (x - pm) ** 2) / (2 * ps * ps) - 0.5)
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    t = f(ar, br).reshape(B, -1).sum(dim=1)
    w = torch.rand(B, generator=g).double()
    (t * w).sum().backward()
    sums, ga, gb = ops.spair_loss(mode, a.cuda(), b.cuda(), *((pm, ps) if mode == "kl_prior" else ()))
    torch.testing.assert_close(sums.double().cpu(), t.detach(), rtol=2e-5, atol=1e-4)
    wv = w.view(B, 1, 1, 1)
    torch.testing.assert_close(gb.double().cpu() * wv, br.grad, rtol=2e-5, atol=1e-5 * float(br.grad.abs().max()))
    if mode == "xent":
        assert ga is None
    else:
        torch.testing.assert_close(ga.double().cpu() * wv, ar.grad, rtol=2e-5, atol=1e-5 * float(ar.grad.abs().max()))


def test_clipnorm_adam_over_separate_gradient_tensors(ops):
    """sv_adam_step_clipnorm_ptrs (gradient tensors as the autograd engine returns them, addresses by value) == the flat-buffer call,
    including tensors whose flat offset is not a multiple of 4 (a 1-element bias shifts everything behind it)."""
    g = torch.Generator().manual_seed(8)
    shapes = [(37, 64), (64,), (64, 1), (1,), (3, 3, 8, 32), (32,), (4099,), (5,)]
    grads = [torch.randn(s, generator=g).cuda() * (3.0 if i % 2 else 0.01) for i, s in enumerate(shapes)]
    offs = np.concatenate([[0], np.cumsum([int(np.prod(s)) for s in shapes])])
    off = torch.tensor(offs, dtype=torch.int64).cuda()
    n = int(offs[-1])
    res = []
    for mode in ("flat", "ptrs"):
        p = torch.linspace(-1, 1, n).cuda()
        m, v = torch.full((n,), 0.01).cuda(), torch.full((n,), 0.02).cuda()
        if mode == "flat":
            ops.adam_step_clipnorm(p, torch.cat([x.reshape(-1) for x in grads]), m, v, off, 1.0, 3, 1e-3)
        else:
            ops.adam_step_clipnorm_tensors(p, grads, m, v, off, 1.0, 3, 1e-3)
        res.append((p, m, v))
    for a, b in zip(*res):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)       # the norms' partial sums are taken in a different split
