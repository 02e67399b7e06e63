"""GPU parity of the whole SPLIT-VAE step (LGVae.call + train_step_lg_vae + Keras-Adam) through
the C ABI plan, against the oracle restatement on identical inputs / eps / perm / weights."""
import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref

pytestmark = pytest.mark.gpu

NAMES10 = ["x_mean", "x_log_scale", "z_x", "z_mean_x", "z_sig_x", "z_x_hat", "x_hat_mean", "x_hat_log_scale",
           "z_mean_x_hat", "z_sig_x_hat"]
LOSS_KEYS = ["x_recon_loss", "x_kl_loss", "x_hat_recon_loss", "x_hat_kl_loss", "total_kl_loss", "total_loss"]


def make_inputs(B, H, patch, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = (rng.integers(0, 256, size=(B, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
    G2 = (H // patch) ** 2
    perm = np.stack([np.random.Generator(np.random.PCG64(seed + 1 + b)).permutation(G2) for b in range(B)]).astype(np.int32)
    eps = np.random.Generator(np.random.PCG64(seed + 2)).standard_normal((2, B, 128)).astype(np.float32)
    return x, perm, eps


def flat_params(plan, params_np):
    flat = torch.zeros(plan.n_params, dtype=torch.float32)
    for (name, off, shape), p in zip(plan.param_table, params_np):
        assert tuple(p.shape) == tuple(shape), (name, p.shape, shape)
        flat[off:off + p.size] = torch.from_numpy(np.ascontiguousarray(p)).flatten()
    return flat.cuda()


def unflat(plan, flat):
    f = flat.cpu()
    return [f[off:off + int(np.prod(shape))].reshape(shape) for (_, off, shape) in plan.param_table]


def outputs10(plan, B, H, L=128):
    o = {}
    out6x = plan.buffer("out6_x", torch.float32, (B, H, H, 6))
    out6h = plan.buffer("out6_xh", torch.float32, (B, H, H, 6))
    o["x_mean"], o["x_log_scale"] = out6x[..., :3], out6x[..., 3:]
    o["x_hat_mean"], o["x_hat_log_scale"] = out6h[..., :3], out6h[..., 3:]
    for k, b in [("z_x", "z_x"), ("z_mean_x", "z_mean_x"), ("z_sig_x", "z_sig_x"), ("z_x_hat", "z_xh"),
                 ("z_mean_x_hat", "z_mean_xh"), ("z_sig_x_hat", "z_sig_xh")]:
        o[k] = plan.buffer(b, torch.float32, (B, L))
    return o


@pytest.fixture(scope="module")
def ops(lib_built):
    assert torch.cuda.is_available()
    from split_vae_amd import ops as o
    return o


@pytest.mark.parametrize("H,patch,beta,B", [(32, 1, 40.0, 4), (64, 8, 120.0, 3)])
def test_step_fp32_matches_oracle(ops, H, patch, beta, B):
    """fp32 MFMA path: forward 10-tuple, 5 scalars, all 40 gradients, weights after 1..3 Adam steps.
    Tolerance (fp32, summation order differs): rtol 1e-4 on reconstructions/ELBO terms; gradients
    2e-3 of each tensor's max |g| (atomics + long reductions); weights atol 2e-6 per step (lr 1e-4)."""
    x, perm, eps = make_inputs(B, H, patch)
    images = ops.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), patch)
    assert np.array_equal(images.cpu().numpy(), np_ref.scramble_batch(x, perm, patch).astype(np.float32))
    params_np = np_ref.glorot_init(H, H, seed=3)
    rng = np.random.default_rng(9)
    for i in range(1, len(params_np), 2):          # non-zero biases so bias paths are exercised
        params_np[i] = (rng.standard_normal(params_np[i].shape) * 0.05).astype(np.float32)
    ref = torch_ref.RefTrainer(params_np, beta, dtype=torch.float64)
    plan = ops.LGVaePlan(B, H, H, beta=beta, dtype=torch.float32)
    P = flat_params(plan, params_np)
    G = torch.zeros_like(P); M = torch.zeros_like(P); V = torch.zeros_like(P)
    ex, eh = torch.from_numpy(eps[0]).cuda(), torch.from_numpy(eps[1]).cuda()
    imgs_cpu = images.cpu().double()
    L = ops._lib  # noqa
    from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
    for t in range(1, 4):
        fwd_ref, loss_ref, g_ref = ref.grads(imgs_cpu, eps[0], eps[1])
        plan.step(PHASE_ALL & ~PHASE_ADAM, params=P, grads=G, images6=images, eps_x=ex, eps_x_hat=eh, t=t)
        torch.cuda.synchronize()
        got = outputs10(plan, B, H)
        for name, r in zip(NAMES10, fwd_ref):
            torch.testing.assert_close(got[name].cpu().double(), r.detach(), rtol=1e-4, atol=1e-4, msg=lambda m: name + ": " + m)
        losses = plan.buffer("losses", torch.float32, (8,)).cpu().double()
        for i, k in enumerate(LOSS_KEYS):
            assert abs(float(losses[i]) - float(loss_ref[k])) <= 1e-4 * abs(float(loss_ref[k])) + 1e-3, k
        for (name, off, shape), gr, gg in zip(plan.param_table, g_ref, unflat(plan, G)):
            tol = 2e-3 * float(gr.abs().max()) + 1e-7
            err = float((gg.double() - gr).abs().max())
            assert err <= tol, "grad %s: err %g tol %g" % (name, err, tol)
        plan.step(PHASE_ADAM, params=P, grads=G, adam_m=M, adam_v=V, t=t)
        ref.t += 1
        torch_ref.keras_adam_(ref.params, g_ref, ref.m, ref.v, ref.t, ref.lr)
        for (name, off, shape), pr, pg in zip(plan.param_table, ref.params, unflat(plan, P)):
            err = float((pg.double() - pr.detach()).abs().max())
            assert err <= 2.5e-6 * t + 2e-7, "param %s after step %d: err %g" % (name, t, err)


def test_step_bf16_close_to_oracle(ops):
    """bf16 MFMA path (config 2's compute type): operands rounded to 8 significant bits.  Stated
    tolerance: ELBO scalars 1e-2 relative; reconstructions atol 3e-2; gradients within 5e-2 of
    each tensor's max |g| (cosine > 0.995)."""
    B, H, patch, beta = 4, 64, 8, 120.0
    x, perm, eps = make_inputs(B, H, patch, seed=5)
    images = ops.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), patch)
    params_np = np_ref.glorot_init(H, H, seed=3)
    ref = torch_ref.RefTrainer(params_np, beta, dtype=torch.float64)
    fwd_ref, loss_ref, g_ref = ref.grads(images.cpu().double(), eps[0], eps[1])
    plan = ops.LGVaePlan(B, H, H, beta=beta, dtype=torch.bfloat16)
    P = flat_params(plan, params_np)
    G = torch.zeros_like(P)
    from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
    plan.step(PHASE_ALL & ~PHASE_ADAM, params=P, grads=G, images6=images, eps_x=torch.from_numpy(eps[0]).cuda(),
              eps_x_hat=torch.from_numpy(eps[1]).cuda())
    torch.cuda.synchronize()
    got = outputs10(plan, B, H)
    for name, r in zip(NAMES10, fwd_ref):
        torch.testing.assert_close(got[name].cpu().double(), r.detach(), rtol=3e-2, atol=3e-2, msg=lambda m: name + ": " + m)
    losses = plan.buffer("losses", torch.float32, (8,)).cpu().double()
    for i, k in enumerate(LOSS_KEYS):
        assert abs(float(losses[i]) - float(loss_ref[k])) <= 1e-2 * abs(float(loss_ref[k])) + 1e-2, k
    for (name, off, shape), gr, gg in zip(plan.param_table, g_ref, unflat(plan, G)):
        gg = gg.double()
        err = float((gg - gr).abs().max())
        assert err <= 5e-2 * float(gr.abs().max()) + 1e-6, "grad %s: err %g max %g" % (name, err, float(gr.abs().max()))
        cos = float((gg * gr).sum() / (gg.norm() * gr.norm() + 1e-30))
        assert cos > 0.995, "grad %s cosine %g" % (name, cos)


def test_step_determinism_of_forward(ops):
    """non-atomic paths (forward, ELBO) are bitwise reproducible run to run."""
    B, H = 4, 32
    x, perm, eps = make_inputs(B, H, 1, seed=7)
    images = ops.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), 1)
    plan = ops.LGVaePlan(B, H, H, beta=40.0, dtype=torch.bfloat16)
    P = flat_params(plan, np_ref.glorot_init(H, H, seed=3))
    from split_vae_amd._lib import PHASE_PREP, PHASE_FORWARD, PHASE_LOSS
    outs = []
    for _ in range(2):
        plan.step(PHASE_PREP | PHASE_FORWARD | PHASE_LOSS, params=P, images6=images, seed=1, step=2)
        torch.cuda.synchronize()
        outs.append((plan.buffer("out6_x", torch.float32, (B, H, H, 6)).clone(),
                     plan.buffer("losses", torch.float32, (8,)).clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1][:6], outs[1][1][:6])
