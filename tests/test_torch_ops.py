"""The `split_vae::*` torch.library operators (split_vae_amd/torch_ops.py): HIP dispatch key only, autograd pairing.
GPU: a conv -> upsample-fused conv -> discretised-logistic ELBO graph composed in torch, backpropagated through the custom
ops, against fp64 autograd of the oracle restatement; reparam+KL likewise."""
import numpy as np
import pytest
import torch

from oracle import torch_ref


def test_ops_are_registered_without_a_cpu_kernel():
    """No CPU dispatch key (and so no route into the oracle): a CPU tensor is refused by the dispatcher."""
    import split_vae_amd.torch_ops  # noqa: F401
    for name in ("scramble_gather", "conv2d_nhwc_fwd", "conv2d_nhwc_dgrad", "conv2d_nhwc_wgrad", "dlogistic_nll", "reparam_kl_fwd",
                 "reparam_kl_bwd", "upsample2x_bwd", "adam_step"):
        assert hasattr(torch.ops.split_vae, name), name
    p = torch.zeros(8)
    with pytest.raises(NotImplementedError):
        torch.ops.split_vae.adam_step(p, p.clone(), p.clone(), p.clone(), 1, 1e-4, 0.9, 0.999, 1e-7, 1.0)
    with pytest.raises(NotImplementedError):
        torch.ops.split_vae.scramble_gather(torch.zeros(1, 8, 8, 3), torch.zeros(1, 4, dtype=torch.int32), 4)
    src = open(split_vae_amd.torch_ops.__file__).read()
    assert "import oracle" not in src and "from oracle" not in src


def _pad8(t):
    c = t.shape[-1]
    return torch.nn.functional.pad(t, (0, (c + 7) // 8 * 8 - c))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
def test_conv_elbo_graph_backpropagates_through_the_custom_ops(lib_built, dtype, tol):
    from split_vae_amd import torch_ops as T
    g = torch.Generator().manual_seed(0)
    B, H, C1, C2 = 4, 16, 32, 16
    x = torch.randn(B, H, H, C1, generator=g) * 0.5
    w1 = torch.randn(4, 4, C1, C2, generator=g) * (1.0 / np.sqrt(16 * C1)); b1 = torch.randn(C2, generator=g) * 0.1
    w2 = torch.randn(6, 6, C2, 6, generator=g) * (1.0 / np.sqrt(36 * C2)); b2 = torch.randn(6, generator=g) * 0.1
    img = (torch.randint(0, 256, (B, 2 * H, 2 * H, 6), generator=g) / 255.0 * 2 - 1).float()
    # ---- fp64 reference graph: conv k4 relu -> 2x bilinear -> conv k6 (6-channel head) -> NLL of channels 3..5 -> batch mean
    r = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    h = torch_ref.conv2d_same(r[0], r[1], r[2], 1, "relu")
    o = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(h), r[3], r[4], 1, None)
    ref = torch_ref.discretised_logistic_loss(img[..., 3:].double(), o[..., :3], o[..., 3:]).sum(dim=(1, 2, 3)).mean()
    ref.backward()
    # ---- the same graph on the HIP ops
    d = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    xh = d[0].to(dtype)
    hh = T.conv2d(xh, d[1], d[2], stride=1, act="relu")
    oh = T.conv2d(hh, d[3], d[4], stride=1, act=None, ups_in=True, y_f32=True)
    assert oh.dtype == torch.float32 and tuple(oh.shape) == (B, 2 * H, 2 * H, 6)
    nll = T.dlogistic_nll(img.cuda(), 3, oh)
    loss = nll.mean()
    loss.backward()
    assert abs(float(loss) - float(ref)) <= tol * abs(float(ref))
    for name, got, want in zip(("x", "w1", "b1", "w2", "b2"), d, r):
        gw, gg = want.grad, got.grad.double().cpu()
        err = float((gg - gw).norm() / gw.norm())
        assert err <= (5 * tol if dtype == torch.bfloat16 else 10 * tol), (name, err)


@pytest.mark.gpu
def test_reparam_kl_op_pairs_forward_and_backward(lib_built):
    from split_vae_amd import torch_ops as T
    g = torch.Generator().manual_seed(1)
    B, L, beta = 8, 128, 40.0
    pre = torch.randn(B, 2 * L, generator=g); bias = torch.randn(2 * L, generator=g) * 0.1; eps = torch.randn(B, L, generator=g)
    gz = torch.randn(B, L, generator=g)
    r = [t.double().requires_grad_(True) for t in (pre, bias)]
    a = r[0] + r[1]
    zm, zs = a[:, :L], torch.nn.functional.softplus(a[:, L:])
    z = zm + zs * eps.double()
    ref = (z * gz.double()).sum() + beta * torch_ref.kl_divergence(zm, zs)
    ref.backward()
    d = [t.cuda().requires_grad_(True) for t in (pre, bias)]
    zh, kl, zmh, zsh = T.reparam_kl(d[0], d[1], eps.cuda())
    torch.testing.assert_close(zh.double().cpu(), z.detach(), rtol=1e-5, atol=1e-5)
    ((zh * gz.cuda()).sum() + beta * kl.mean()).backward()
    for got, want in zip(d, r):
        torch.testing.assert_close(got.grad.double().cpu(), want.grad, rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
def test_scramble_and_adam_ops(lib_built):
    import split_vae_amd.torch_ops  # noqa: F401
    from oracle import np_ref
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 16, 16, 3)).astype(np.float32)
    perm = np.stack([rng.permutation(16) for _ in range(3)]).astype(np.int32)
    out = torch.ops.split_vae.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), 4)
    assert np.array_equal(out.cpu().numpy(), np_ref.scramble_batch(x, perm, 4).astype(np.float32))
    p = torch.randn(1000, device="cuda"); g = torch.randn(1000, device="cuda")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p0 = p.clone()
    torch.ops.split_vae.adam_step(p, g, m, v, 1, 1e-3, 0.9, 0.999, 1e-7, 1.0)
    want = p0.double() - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g.double()) / ((0.001 * g.double() ** 2).sqrt() + 1e-7)
    torch.testing.assert_close(p.double(), want, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_spair_ops_compose_in_one_autograd_graph(lib_built):
    """glimpses = STN(x, z_where); objects pasted back with the inverse STN; Renderer over a background; xent + z_pres KL
    loss; backward through all of it -- against torch autograd of the fp64 oracle composition (spair/utils.py:119-330,
    spair/spair.py:534-579, spair/trainer.py:45-101)."""
    from oracle import spair_ref as S
    from split_vae_amd import torch_ops as T
    g = torch.Generator().manual_seed(3)
    B, Hc = 4, 4
    x = torch.rand(B, 48, 48, 3, generator=g)
    z_where = torch.randn(B, Hc, Hc, 4, generator=g) * 0.7
    z_depth = torch.randn(B, Hc, Hc, 1, generator=g)
    pre = torch.randn(B, Hc, Hc, 1, generator=g)
    logits = torch.randn(B, Hc, Hc, 1, generator=g)
    bg = torch.rand(B, 48, 48, 3, generator=g)
    alpha = torch.rand(B, 16, 32, 32, 1, generator=g)

    def graph(stn, render, zkl, t):
        x_, zw, zd, pr, lg, bg_ = t
        glimpses, _ = stn(x_, zw, 32, 32, False)                             # [B,16,32,32,3]
        obj = torch.cat([glimpses, alpha.to(glimpses)], dim=-1)              # rgb from the glimpses + a fixed alpha
        pasted, _ = stn(obj, zw, 48, 48, True)                               # [B,16,48,48,4]
        zp = torch.sigmoid(pr)
        canvas = render(pasted, bg_, zd, zp)
        recon = -(x_ * torch.log(canvas + 1e-8) + (1 - x_) * torch.log(1 - canvas + 1e-8)).sum(dim=(1, 2, 3)).mean()
        return recon + zkl(zp, lg, pr)

    r = [t.double().requires_grad_(True) for t in (x, z_where, z_depth, pre, logits, bg)]
    ref = graph(lambda a, b, h, w, inv: S.stn_forward(a, b, h, w, inverse=inv),
                lambda o, b_, zd, zp: S.renderer(o, b_, zd, zp, None, training=True),
                lambda zp, lg, pr: S.compute_z_pres_kl_yolo_air(zp, lg, pr, 0.3, 1.5), r)
    ref.backward()
    d = [t.cuda().requires_grad_(True) for t in (x, z_where, z_depth, pre, logits, bg)]
    got = graph(lambda a, b, h, w, inv: T.stn_sample(a, b, h, w, inv),
                lambda o, b_, zd, zp: T.spair_render(o, b_, zd, zp),
                lambda zp, lg, pr: T.spair_zpres_kl(zp, lg, pr, 0.3, 1.5).mean(), d)
    got.backward()
    assert abs(float(got) - float(ref)) <= 1e-4 * abs(float(ref))
    for name, a, b in zip(("x", "z_where", "z_depth", "pre_sigmoid", "logits", "bg"), d, r):
        err = float((a.grad.double().cpu() - b.grad).norm() / b.grad.norm())
        assert err < 5e-3, (name, err)
