"""The SPAIR / SPLIT-SPAIR model restatement (oracle/spair_model_ref.py) against hand-derived facts of spair/spair.py and
spair/trainer.py: variable tables, tensor shapes of the returned tuple, closed-form loss terms, the clipnorm update."""
import math

import numpy as np
import pytest
import torch

from oracle import spair_model_ref as R


def _count(cfg):
    return sum(int(np.prod(s)) + s[-1] for _, s in R.param_spec(cfg))


def test_variable_tables():
    # Encoder (spair/spair.py:382-401) + ObjEncoder (:250-255) + ObjDecoder (:348-352), latent 128, object 32, 3 channels
    L = 128
    enc = (4 * 4 * 3 * 128 + 128) + 2 * (4 * 4 * 128 * 128 + 128) + 2 * (128 * 128 + 128) + (128 * 100 + 100)
    where = (100 * 128 + 128) + (128 * 64 + 64) + (64 * 16 + 16)
    depth = ((100 + 8 + 4 + L) * 64 + 64) + (64 * 10 + 10)
    pres = ((100 + 8 + 4 + L + 1) * 64 + 64) + (64 * 1 + 1)
    obj_enc = (3 * 3 * 3 * 32 + 32) + (3 * 3 * 32 * 64 + 64) + (8 * 8 * 64 * 2 * L + 2 * L) + 2 * (2 * L * L + L)
    obj_dec = (L * 2 * L + 2 * L) + (2 * L * 2048 + 2048) + (3 * 3 * 32 * 64 + 64) + (3 * 3 * 64 * 32 + 32) + (3 * 3 * 32 * 4 + 4)
    assert _count(R.default_config(model="spair")) == enc + where + depth + pres + obj_enc + obj_dec
    spec = R.param_spec(R.default_config(model="spair"))
    assert [n for n, _ in spec[:6]] == ["encoder/" + n for n in ("conv1", "conv2", "conv3", "z1", "z2", "z3")]
    assert len(spec) == 23
    # README.md:93 SPLIT-SPAIR: dense bg / local nets; concat_z_what widens ObjDecoder.d0 by the local latent
    cfg = R.default_config(model="lg_spair", latent_size=64, bg_latent_size=4, local_latent_size=4, split_z_l=True, concat_z_what=True,
                           dense_local=True, dense_bg=True)
    d = dict(R.param_spec(cfg))
    assert d["decoder/obj_decoder/d0"] == (68, 128)
    assert d["bg_encoder/e1"] == (48 * 48 * 3, 1024) and d["bg_decoder/d1"] == (4, 500) and d["x_hat_decoder/d3"] == (1024, 6912)
    assert "encoder/dense_z_l/0" not in d
    # concat_backbone: 16 more features per cell (:401, :406-409)
    d2 = dict(R.param_spec(R.default_config(model="lg_spair", concat_backbone=True)))
    assert d2["encoder/dense_z_where/0"] == (116, 128) and d2["encoder/dense_z_l/0"] == (64, 16)
    assert d2["bg_encoder/z_mu"] == (6 * 6 * 128, 4) and d2["bg_decoder/d1"] == (4, 6 * 6 * 128)
    # bg_spair: BackgroundModel (:205-244)
    d3 = [n for n, _ in R.param_spec(R.default_config(model="bg_spair"))]
    assert d3[-10:] == ["bg_model/" + n for n in ("e1", "e2", "e3", "z_bg_mu", "z_bg_sigma", "d1", "d2", "d3", "d4", "d5")]


@pytest.mark.parametrize("kw", [dict(model="spair", latent_size=16), dict(model="bg_spair", latent_size=16),
                                dict(model="lg_spair", latent_size=16, local_latent_size=4, split_z_l=True, concat_z_what=True,
                                     dense_local=True, dense_bg=True)])
def test_forward_shapes_and_gradients(kw):
    cfg = R.default_config(**kw)
    B = 2
    p = R.init_params(cfg, 0, torch.float32)
    for v in p.values():
        v.requires_grad_(True)
    images = torch.rand(B, 48, 48, 6 if cfg.model == "lg_spair" else 3)
    o = R.forward(p, cfg, images, R.draw_noise(cfg, B, 1, torch.float32), training=True)
    Lw = 16 + (4 if cfg.concat_z_what else 0)
    assert o["x_recon"].shape == (B, 48, 48, 3) and o["z_what"].shape == (B, 4, 4, Lw) and o["z_where"].shape == (B, 4, 4, 4)
    assert o["all_glimpses"].shape == (B, 16, 32, 32, 3) and o["obj_full_recon_unnorm"].shape == (B, 16, 48, 48, 4)
    assert o["obj_recon_alpha"].shape == (B, 16, 32, 32, 1) and o["obj_bbox_mask"].shape == (B, 16, 4)
    assert float(o["x_recon"].min()) >= 0 and float(o["x_recon"].max()) <= 1
    assert float(o["z_pres_logits"].abs().max()) <= 10
    total, lst = R.losses(cfg, images, o, step=0)
    assert len(lst) == {"spair": 6, "bg_spair": 7, "lg_spair": 9}[cfg.model]
    g = torch.autograd.grad(total, list(p.values()), allow_unused=True)
    assert all(x is not None and bool(torch.isfinite(x).all()) for x in g)      # every variable is on the loss's path


def test_loss_terms_closed_form():
    m = torch.zeros(3, 5, dtype=torch.float64)
    s = torch.ones(3, 5, dtype=torch.float64)
    # -0.5 * sum(1 + log(1 + 1e-8) - 0 - (1 + 1e-8)) ~ 0
    assert abs(float(R.kl_divergence(m, s))) < 1e-12
    s2 = torch.full((3, 5), 2.0, dtype=torch.float64)
    want = 5 * (-0.5) * (1 + math.log(4 + 1e-8) - 0.25 - (4 + 1e-8))
    assert abs(float(R.kl_divergence(m + 0.5, s2)) - want) < 1e-12
    a = torch.full((2, 1, 1, 2), 0.3, dtype=torch.float64)
    assert abs(float(R.kl_divergence_two_gauss(a, a + 0.2, a, a + 0.2))) < 1e-12
    want = 2 * (math.log(0.5 + 1e-8) - math.log(1.0 + 1e-8) + (1.0 + 0.09) / (2 * 0.25) - 0.5)
    assert abs(float(R.kl_divergence_two_gauss(a, torch.ones_like(a), torch.zeros_like(a), torch.full_like(a, 0.5))) - want) < 1e-12
    lab = torch.tensor([[0.0, 1.0, 0.25]], dtype=torch.float64)
    x = R.xent_loss(lab, torch.tensor([[0.0, 1.0, 0.5]], dtype=torch.float64))
    assert abs(float(x[0, 0]) + math.log(1 + 1e-8)) < 1e-12 and abs(float(x[0, 2]) + math.log(0.5 + 1e-8)) < 1e-12


def test_annealing_and_model_branches():
    cfg = R.default_config(model="spair", latent_size=8)
    p = R.init_params(cfg, 0)
    images = torch.rand(2, 48, 48, 3, dtype=torch.float64)
    o = R.forward(p, cfg, images, R.draw_noise(cfg, 2, 1), training=True)
    t0, l0 = R.losses(cfg, images, o, step=0)
    t1, l1 = R.losses(cfg, images, o, step=20000)
    # the zoom prior anneals from prior_z_zoom + 10 to prior_z_zoom (trainer.py:156): only that term and the z_pres prior move
    assert float(l0[1]) != float(l1[1]) and float(l0[5]) != float(l1[5])
    for i in (0, 2, 3, 4):
        assert float(l0[i]) == float(l1[i])
    obj = cfg.z_what_beta * l1[2] + l1[4] + l1[3] + l1[1] + l1[5]
    assert abs(float(t1) - float(cfg.reconstruction_weight * l1[0] + cfg.beta * obj)) < 1e-9 * abs(float(t1))


def test_clipnorm_adam_first_step():
    p = [torch.zeros(4, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)]
    g = [torch.tensor([3.0, 0.0, 4.0, 0.0], dtype=torch.float64), torch.tensor([0.3, 0.4], dtype=torch.float64)]    # norms 5 and 0.5
    m = [torch.zeros_like(x) for x in p]
    v = [torch.zeros_like(x) for x in p]
    R.clipnorm_adam_(p, g, m, v, 1, lr=0.1, clipnorm=1.0, clip_in_apply=True)                        # TF >= 2.4: apply_gradients clips
    assert torch.allclose(m[0], 0.1 * torch.tensor([0.6, 0.0, 0.8, 0.0], dtype=torch.float64))      # clipped to unit norm
    assert torch.allclose(m[1], 0.1 * g[1])                                                           # below the threshold: untouched
    assert float(p[0][0]) < 0 and float(p[0][1]) == 0.0
    # [TF-2.0 semantics] the pinned tensorflow_gpu==2.0.0: apply_gradients does not clip -> the plain Keras Adam (the default)
    p2 = [torch.zeros(4, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)]
    m2 = [torch.zeros_like(x) for x in p2]
    v2 = [torch.zeros_like(x) for x in p2]
    R.clipnorm_adam_(p2, g, m2, v2, 1, lr=0.1, clipnorm=1.0)
    assert torch.allclose(m2[0], 0.1 * g[0]) and torch.allclose(m2[1], 0.1 * g[1])


@pytest.mark.parametrize("fixture", ["lgspair_b2.npz", "lgspair_hard_b2.npz"])
def test_oracle_reproduces_the_committed_spair_fixture(fixture):
    """tests/golden/lgspair_b2.npz (README.md:93, Multi-Bird-Easy) and lgspair_hard_b2.npz (README.md:107, Multi-Bird-Hard = BASELINE config 5) (made by tests/golden/make_golden_spair.py from this restatement): any change to the oracle's
    arithmetic shows up here before it silently moves the GPU tests' target."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_spair", os.path.join(here, "golden", "make_golden_spair.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.compute(mod.FIXTURES[fixture])
    with np.load(os.path.join(here, "golden", fixture)) as G:
        assert set(G.files) == set(got)
        for k in G.files:
            if k == "grad_err_f32":          # (a noise scale of this host's fp32 evaluation: order of magnitude only)
                assert got[k].shape == G[k].shape and np.all(got[k] <= 10 * G[k] + 1e-6), k
                continue
            if k == "grad_norms_f32":        # the fp32 evaluation of the graph on THIS host's CPU (thread count / vector width reorder its sums): a noise scale, not a pinned value
                np.testing.assert_allclose(got[k], G[k], rtol=2e-2, err_msg=k)
                continue
            np.testing.assert_allclose(got[k], G[k], rtol=1e-9, atol=1e-12, err_msg=k)
