"""GPU parity of the SPLIT-GMVAE step (LGGMVae.call + train_step_lg_gm_vae + Keras-Adam; SURVEY 8a row A9,
config 3: SVHN-32, y_size 30, tau 0.4, beta 40, alpha 40, patch 4) against the oracle restatement
(oracle/gm_ref.py) on identical inputs, weights and random draws (eps, Gumbel uniforms, dropout masks)."""
import os

import numpy as np
import pytest
import torch

from oracle import gm_ref, np_ref

pytestmark = pytest.mark.gpu

H, PATCH, BETA, ALPHA, K, TAU = 32, 4, 40.0, 40.0, 30, 0.4
NAMES14 = ["x_mean", "x_log_scale", "z_x", "z_mean_x", "z_sig_x", "z_x_hat", "x_hat_mean", "x_hat_log_scale", "z_mean_x_hat",
           "z_sig_x_hat", "y", "y_logits", "z_prior_mean", "z_prior_sig"]


@pytest.fixture(scope="module")
def ops(lib_built):
    assert torch.cuda.is_available()
    from split_vae_amd import ops as o
    return o


def _inputs(B, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    x = (rng.integers(0, 256, size=(B, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
    G2 = (H // PATCH) ** 2
    perm = np.stack([rng.permutation(G2) for _ in range(B)]).astype(np.int32)
    images = np_ref.scramble_batch(x, perm, PATCH).astype(np.float32)
    F_ = (H // 8) ** 2 * 128
    noise = dict(eps_x=rng.standard_normal((B, 128)).astype(np.float32), eps_h=rng.standard_normal((B, 128)).astype(np.float32),
                 u=rng.uniform(0.02, 0.98, (B, K)).astype(np.float32),
                 keep1=(rng.uniform(size=(B, 1024)) > 0.2).astype(np.float32),
                 keep5=(rng.uniform(size=(B, F_)) > 0.2).astype(np.float32))
    return images, noise


def _params(seed=3):
    ps = gm_ref.gm_glorot_init(H, H, seed=seed, y_size=K)
    rng = np.random.default_rng(9)
    names = [n for n, _ in gm_ref.gm_param_shapes(H, H, y_size=K)]
    for i, n in enumerate(names):
        if n.endswith("bias"):
            ps[i] = (ps[i] + rng.standard_normal(ps[i].shape) * 0.05).astype(np.float32)    # exercise the bias paths
    return ps


def test_gm_variable_table_matches_reference_order(ops):
    from split_vae_amd.gm import LGGMVae
    m = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="f32", device="cuda", seed=1)
    want = gm_ref.gm_param_shapes(H, H, y_size=K)
    assert m.keras_names() == [n + ":0" for n, _ in want]
    assert [tuple(v.shape) for v in m.trainable_variables] == [tuple(s) for _, s in want]
    assert sum(v.numel() for v in m.trainable_variables) == 6775370          # 6.78 M parameters
    b = dict(zip(m.keras_names(), m.get_weights()))
    assert np.all(b["encoder_x/z_sig/bias:0"] == 1) and np.all(b["encoder_x/z_prior_sig/bias:0"] == 1)   # vae/model.py:68,:78
    assert np.all(b["encoder_x/z_mean/bias:0"] == 0)


STRICT_INPUT_SEED = int(os.environ.get("SV_GM_TEST_SEED", "0"))     # inputs of the strict test (no ReLU unit within fp32 rounding of its kink)


@pytest.mark.parametrize("dropout", [False, True], ids=["tf2.0-no-dropout", "tf2.1-dropout"])
def test_gm_step_fp32_matches_oracle(ops, deterministic, dropout):
    """THE fp32 parity test of the SPLIT-GMVAE step (primary): fixed-order reductions against the fp64 oracle -- the 14-tuple at
    rtol 1e-4 / atol 1e-5 of each tensor's scale, the 5 metrics + total, all 54 gradients within 2e-3 with NO element-fraction allowance,
    weights after Adam.  Both readings of `training` (oracle/gm_ref.py::encoder_gmvae): dropout off in training (tensorflow 2.0.0, the
    default) and on (tensorflow >= 2.1)."""
    _gm_step_vs_oracle(dropout, strict=True, seed=STRICT_INPUT_SEED)


@pytest.mark.parametrize("dropout", [False, True], ids=["tf2.0-no-dropout", "tf2.1-dropout"])
def test_gm_step_fp32_default_summation_order_matches_oracle(ops, dropout):
    """The same comparison on the DEFAULT path (split-K fp32 atomics in the Dense layers: the summation order changes from run to run),
    with the looser output bound and the one-ReLU-gate allowance described in the body."""
    _gm_step_vs_oracle(dropout, strict=bool(os.environ.get("SV_TEST_STRICT")), seed=0)


def _gm_step_vs_oracle(dropout, strict, seed):
    from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae, LOSS_KEYS
    from split_vae_amd.optimizer import Adam
    B = 4
    images, nz = _inputs(B, seed=seed)
    params = _params()
    ref = gm_ref.GMRefTrainer(params, BETA, ALPHA, y_size=K, tau=TAU, dtype=torch.float64, dropout=dropout)
    model = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="f32", device="cuda", seed=1, dropout_in_training=dropout)
    assert LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="f32", device="cuda").dropout_in_training is False   # pinned-TF default
    model.beta, model.alpha = BETA, ALPHA
    model.set_weights(params)
    opt = Adam(learning_rate=1e-4)
    img = torch.from_numpy(images).cuda()
    cu = lambda a: torch.from_numpy(a).cuda()
    eps = (cu(nz["eps_x"]), cu(nz["eps_h"]))
    noise = (cu(nz["u"]), cu(nz["keep1"]), cu(nz["keep5"]))
    args = (images, nz["eps_x"], nz["eps_h"], nz["u"], nz["keep1"], nz["keep5"])

    # forward surface, training=True semantics pinned through the masks (vae/trainer.py:149)
    fwd_ref, loss_ref, g_ref = ref.grads(*args)
    out = model(img, training=True, eps=eps, noise=noise)
    for name, got, want in zip(NAMES14, out, fwd_ref):
        want = want.detach()
        torch.testing.assert_close(got.double().cpu(), want, rtol=1e-4 if strict else 2e-4,
                                   atol=(1e-5 if strict else 2e-4) * max(1.0, float(want.abs().max())), msg=lambda m: name + ": " + m)
    assert torch.allclose(out[10].sum(dim=1).cpu(), torch.ones(B), atol=1e-5)       # Gumbel-softmax rows sum to one

    for t in range(1, 3):
        fwd_ref, loss_ref, g_ref = ref.grads(*args)
        metrics = train_step_lg_gm_vae(model, img, opt, eps=eps, noise=noise).cpu().double()
        for i, k in enumerate(LOSS_KEYS):
            want = float(loss_ref[k])
            assert abs(float(metrics[i]) - want) <= 2e-4 * abs(want) + 1e-5, (t, k, float(metrics[i]), want)
        names = model.keras_names()
        for name, got, want in zip(names, model.gradients, g_ref):
            scale = float(want.abs().max())
            got = got.double().cpu()
            try:
                torch.testing.assert_close(got, want, rtol=2e-3, atol=2e-3 * scale + 1e-9,
                                           msg=lambda m: "step %d grad %s: %s" % (t, name, m))
            except AssertionError:
                if strict:     # fixed-order reductions: the plain bound, no second outcome
                    raise
                # About one run in twenty a ReLU unit of decoder_x (one channel of h4 at one pixel, for these inputs) has a
                # pre-activation within fp32 summation-order noise (split-K atomics upstream) of ZERO after the first update,
                # and its gate falls on the other side than in the fp64 oracle.  The forward value hardly moves (the unit is
                # ~0 either way: out6 and the loss gradient are identical), but that unit's share of the gradients is
                # switched on or off: a deterministic second outcome that moves a few thousandths of a gradient's scale in
                # well under 0.1 % of its elements -- d4's bias in ONE element, its kernel in that channel's slice, d1's
                # kernel in 0.08 % (scripts/stress_gm_race2.py: the same deviation every time, also without the side stream
                # and with the weights pinned, the loss gradient g5 bit-for-bit equal: not a race).  Accept exactly that.
                bad = ((got - want).abs() > 2e-3 * scale + 2e-3 * want.abs()).double().mean()
                rel = float((got - want).norm() / want.norm().clamp_min(1e-30))
                assert float(bad) <= 2e-3 and rel <= 5e-3, "step %d grad %s: %.2e of the elements off, relative L2 error %.2e" % (t, name, float(bad), rel)
        # oracle takes the same Adam step from ITS gradients; then re-synchronise so that a sign flip of a
        # near-zero gradient in one place does not compound (tests/test_gpu_step.py does the same)
        before = [p.detach().clone() for p in ref.params]
        ref.t += 1
        from oracle import torch_ref
        torch_ref.keras_adam_(ref.params, g_ref, ref.m, ref.v, ref.t, ref.lr)
        upd_ref = torch.cat([(a.detach() - b).flatten() for a, b in zip(ref.params, before)])
        upd_got = torch.cat([(torch.as_tensor(w).double() - b).flatten() for w, b in zip(model.get_weights(), before)])
        agree = float((torch.sign(upd_ref) == torch.sign(upd_got)).double().mean())
        assert agree > 0.995, agree
        assert float((upd_ref - upd_got).abs().max()) <= 2.1e-4                        # never more than 2 lr apart
        with torch.no_grad():
            for p, w in zip(ref.params, model.get_weights()):
                p.copy_(torch.as_tensor(w).double())


def test_gm_step_bf16_close_and_eval_mode(ops):
    """bf16 contractions: loss terms near the fp64 oracle; evaluation (training=False) applies no dropout."""
    from split_vae_amd.gm import LGGMVae, test_step_lg_gm_vae as eval_step, LOSS_KEYS
    B = 8
    images, nz = _inputs(B, seed=5)
    params = _params()
    ref = gm_ref.GMRefTrainer(params, BETA, ALPHA, y_size=K, tau=TAU, dtype=torch.float64)
    ones1, ones5 = np.ones_like(nz["keep1"]), np.ones_like(nz["keep5"])
    # no dropout: the oracle's "training" scaling 1/(1-rate) must not be applied either -> compare against a forward with
    # masks of (1-rate) so that mask/(1-rate) == 1
    _, loss_ref = ref.forward_losses(images, nz["eps_x"], nz["eps_h"], nz["u"], ones1 * 0.8, ones5 * 0.8)
    model = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="bf16", device="cuda", seed=1)
    model.beta, model.alpha = BETA, ALPHA
    model.set_weights(params)
    cu = lambda a: torch.from_numpy(a).cuda()
    m = eval_step(model, cu(images), eps=(cu(nz["eps_x"]), cu(nz["eps_h"])), noise=(cu(nz["u"]), None, None)).cpu()
    for i, k in enumerate(LOSS_KEYS):
        want = float(loss_ref[k])
        assert abs(float(m[i]) - want) <= 2e-2 * abs(want) + 2e-2, (k, float(m[i]), want)


def test_gm_training_descends_with_device_rng(ops):
    """No pinned noise: eps, Gumbel uniforms and dropout masks come from the Philox streams; Adam descends."""
    from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
    from split_vae_amd.optimizer import Adam
    B = 16
    images, _ = _inputs(B, seed=7)
    for dropout in (True, False):
        model = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="bf16", device="cuda", seed=4, dropout_in_training=dropout)
        model.beta, model.alpha = BETA, ALPHA
        opt = Adam(learning_rate=1e-3)
        img = torch.from_numpy(images).cuda()
        tot = [float(train_step_lg_gm_vae(model, img, opt)[5]) for _ in range(15)]
        assert all(np.isfinite(tot)) and min(tot[-3:]) < tot[0], tot
        if dropout:
            keep1 = model.encoder(B).buf["keep1"]
            frac = float(keep1.mean())
            assert 0.75 < frac < 0.85, frac                             # Dropout(rate=0.2) keeps ~80 %


def test_two_stream_step_on_staged_inputs_equals_the_plain_step(ops):
    """Inputs staged by the augmentation (Augmentator.augment(..., plan=)): the global (GM) encoder runs on a second HIP stream beside the local encoder, forward
    and backward (gm.py: _forward / _train_step_lg_gm_vae; vae/model.py:236-240 has no order between the two encoders).  Three train steps on the same batch,
    weights and Philox streams against the un-staged step (single-stream forward): the six metrics of every step and the updated variables agree to the noise of
    the step's split-K atomics -- a missing stream dependency would feed the decoders a zcat that does not exist yet."""
    from split_vae_amd import data
    from split_vae_amd.augmentation import Augmentator
    from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
    from split_vae_amd.optimizer import Adam
    B = 32
    x = data.synthetic_images(B, H, H, seed=0, device="cuda")
    res = []
    for staged in (False, False, True):
        model = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype="f32", device="cuda", seed=4)
        model.beta, model.alpha = BETA, ALPHA
        opt = Adam(learning_rate=1e-3)
        aug = Augmentator("scramble", size=PATCH, seed=1)
        ms = []
        for _ in range(3):
            img = aug.augment(x, plan=model.plan(B) if staged else None)
            assert (getattr(img, "_sv_staged_plan", None) is not None) == staged
            ms.append(train_step_lg_gm_vae(model, img, opt).cpu().numpy().astype(np.float64))
        torch.cuda.synchronize()
        res.append((np.stack(ms), model.flat.cpu().numpy().astype(np.float64), model.gm_flat.cpu().numpy().astype(np.float64)))
    (m0, p0, g0), (m1, p1, g1), (m2, p2, g2) = res
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30))
    assert rel(m0, m2) <= 10 * rel(m0, m1) + 1e-5, (rel(m0, m2), rel(m0, m1))
    assert rel(p0, p2) <= 10 * rel(p0, p1) + 1e-5 and rel(g0, g2) <= 10 * rel(g0, g1) + 1e-5, (rel(p0, p2), rel(p0, p1), rel(g0, g2), rel(g0, g1))
