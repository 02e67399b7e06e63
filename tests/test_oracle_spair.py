"""CPU known-answer tests that pin oracle/spair_ref.py (first SPLIT-SPAIR pieces): TF-2.0 is not
installable and the reference holds no vectors, so each function is checked against facts derivable from its definition."""
import math

import numpy as np
import torch

from oracle import np_ref, spair_ref as S


def test_backbone_geometry_48_24_12_4():
    """spair/spair.py:382-384: SAME padding with strides 2, 2, 3 takes 48 -> 24 -> 12 -> 4 (the 4x4 cell grid hard-coded at
    spair/trainer.py:346, spair/utils.py:102); stride 3, k 4 on 12 pads 0 before / 1 after."""
    assert np_ref.same_pads(48, 4, 2)[0] == 24 and np_ref.same_pads(24, 4, 2)[0] == 12
    assert np_ref.same_pads(12, 4, 3) == (4, 0, 1)
    x = torch.randn(2, 48, 48, 3, dtype=torch.float64)
    p = [torch.from_numpy(a).double() for a in S.backbone_init(1)]
    outs = S.backbone_forward(x, p)
    assert [tuple(o.shape[1:]) for o in outs] == [(24, 24, 128), (12, 12, 128), (4, 4, 128), (4, 4, 128), (4, 4, 128), (4, 4, 100)]
    # stride-3 layer against a direct loop at one output pixel
    h2, w, b = outs[1], p[4], p[5]
    oy, ox, co = 3, 1, 17
    acc = b[co].clone()
    for ky in range(4):
        for kx in range(4):
            iy, ix = oy * 3 + ky, ox * 3 + kx            # pad before = 0
            if iy < 12 and ix < 12:
                acc = acc + (h2[1, iy, ix] * w[ky, kx, :, co]).sum()
    assert abs(float(torch.relu(acc)) - float(outs[2][1, oy, ox, co])) < 1e-10


def test_bilinear_sampler_identity_and_constants():
    """Grid == pixel centres reproduces the image; a constant image stays constant for any in-range grid; an affine image
    a*x + b*y is reproduced exactly at fractional positions (bilinear interpolation is exact on affine functions)."""
    g = torch.Generator().manual_seed(0)
    img = torch.rand(2, 7, 9, 3, dtype=torch.float64, generator=g)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, 7, dtype=torch.float64), torch.linspace(-1, 1, 9, dtype=torch.float64), indexing="ij")
    grid = torch.stack([xs, ys])[None, None].repeat(2, 1, 1, 1, 1)                     # [B,1,2,7,9]
    out = S.bilinear_sampler(img, grid)
    assert torch.allclose(out[:, 0, :-1, :-1], img[:, :-1, :-1], atol=1e-12)
    assert float(out[:, 0, -1].abs().max()) == 0.0 and float(out[:, 0, :, -1].abs().max()) == 0.0   # the border quirk (next test)
    const = torch.full((1, 5, 6, 2), 3.25, dtype=torch.float64)
    rnd = torch.rand(1, 4, 2, 8, 8, dtype=torch.float64, generator=g) * 1.9 - 0.95
    assert torch.allclose(S.bilinear_sampler(const, rnd), torch.full((1, 4, 8, 8, 2), 3.25, dtype=torch.float64), atol=1e-12)
    yy, xx = torch.meshgrid(torch.arange(5, dtype=torch.float64), torch.arange(6, dtype=torch.float64), indexing="ij")
    aff = (2.0 * xx - 0.5 * yy)[None, :, :, None]
    o = S.bilinear_sampler(aff, rnd)
    px = 0.5 * (rnd[:, :, 0] + 1) * 5
    py = 0.5 * (rnd[:, :, 1] + 1) * 4
    assert torch.allclose(o[..., 0], 2.0 * px - 0.5 * py, atol=1e-10)


def test_bilinear_sampler_border_clamp_semantics():
    """x exactly on the last column: x0 = W-1, x1 = W clamps to W-1, so (x1 - x) = 0 and (x - x0) = 0: the reference's
    formula returns ZERO there (weights from clamped corners), not the border pixel -- restated as written (utils.py:233-246)."""
    img = torch.ones(1, 4, 4, 1, dtype=torch.float64)
    grid = torch.tensor([1.0, 0.0], dtype=torch.float64).view(1, 1, 2, 1, 1)         # x = +1 (last column), y = centre
    assert float(S.bilinear_sampler(img, grid)) == 0.0
    inside = torch.tensor([0.999, 0.0], dtype=torch.float64).view(1, 1, 2, 1, 1)
    assert abs(float(S.bilinear_sampler(img, inside)) - 1.0) < 1e-12


def test_stn_grid_and_bbox():
    """z_where = 0: sx = sy = 0.25, tx = bias_tx: the glimpse of cell (i, j) is centred on the cell's bias point and spans a
    quarter of the normalised canvas; the bbox mask is (centre +- s/2) in [0,1] coordinates (:146-154)."""
    x = torch.rand(1, 48, 48, 3, dtype=torch.float64)
    z = torch.zeros(1, 4, 4, 4, dtype=torch.float64)
    glimpses, bbox = S.stn_forward(x, z, 32, 32)
    assert tuple(glimpses.shape) == (1, 16, 32, 32, 3) and tuple(bbox.shape) == (1, 16, 4)
    _, btx, bty = S.stn_constants(4, 4, 32, 32)
    assert torch.allclose(btx[0], torch.tensor([-0.75, -0.25, 0.25, 0.75], dtype=torch.float64), atol=1e-7)   # cell centres, ratio 0.5
    c = 5                                                     # cell (1, 1)
    ty, tx = float(bty[1, 1]), float(btx[1, 1])
    assert torch.allclose(bbox[0, c], torch.tensor([(ty + 1) / 2 - 0.125, (tx + 1) / 2 - 0.125, (ty + 1) / 2 + 0.125, (tx + 1) / 2 + 0.125],
                                                   dtype=torch.float64), atol=1e-7)
    # the glimpse's centre sample (between grid points 15/16) reads the canvas around the cell centre
    cx = 0.5 * (tx + 1) * 47
    cy = 0.5 * (ty + 1) * 47
    mid = glimpses[0, c, 15:17, 15:17].mean(dim=(0, 1))
    ref = x[0, int(cy) - 1:int(cy) + 3, int(cx) - 1:int(cx) + 3].mean(dim=(0, 1))
    assert float((mid - ref).abs().max()) < 0.35             # same neighbourhood (loose: a smoke check of the geometry)


def test_concrete_kl_and_count_prior():
    """concrete_binary_sample_kl is zero when posterior == prior; the first cell's prior is the mean object fraction of the
    truncated-geometric count prior; switching every cell on moves the running count as the loop says."""
    y = torch.randn(3, 1, dtype=torch.float64)
    lo = torch.randn(3, 1, dtype=torch.float64)
    assert torch.allclose(S.concrete_binary_sample_kl(y, lo, 0.7, lo, 0.7), torch.zeros(3, 1, dtype=torch.float64), atol=1e-12)
    B, H, W = 2, 4, 4
    prior_prob, temp = 0.01, 1.0
    # posterior == the sequential prior at every cell  =>  total KL == 0: build the prior odds with the same recursion
    support = torch.arange(17, dtype=torch.float64)
    cpp = 1 - prior_prob
    dist = (1 - cpp) * cpp ** support
    dist = dist / dist.sum()
    p0 = float((dist * support / 16).sum())                  # P(cell 0 on) = E[count] / 16
    z_pres = torch.zeros(B, H, W, 1, dtype=torch.float64)    # nothing switched on: count_so_far stays 0
    logits = torch.zeros(B, H, W, 1, dtype=torch.float64)
    d = dist.clone()
    for i in range(16):
        pz_given = torch.clamp(support, min=0) / (16 - i)
        pz = float((d * pz_given).sum())
        logits[:, i // 4, i % 4, 0] = float(S.tf_safe_log(torch.tensor(pz, dtype=torch.float64)) - S.tf_safe_log(torch.tensor(1 - pz, dtype=torch.float64)))
        if i == 0:
            assert abs(pz - p0) < 1e-12
        d = (1 - pz_given) * d
        d = d / d.sum()
    pre = torch.randn(B, H, W, 1, dtype=torch.float64)
    kl = S.compute_z_pres_kl_yolo_air(z_pres, logits, pre, prior_prob, temp)
    assert abs(float(kl)) < 1e-9
    # a confident "on" posterior against a 1 % prior costs information
    kl_on = S.compute_z_pres_kl_yolo_air(torch.ones(B, H, W, 1, dtype=torch.float64), torch.full((B, H, W, 1), 4.0, dtype=torch.float64),
                                         torch.full((B, H, W, 1), 4.0, dtype=torch.float64), prior_prob, temp)
    assert float(kl_on) > 1.0 and math.isfinite(float(kl_on))
    assert float(S.tf_safe_log(torch.tensor(0.0))) == float(torch.log(torch.tensor(1e-8)))       # log(0 + 1e-8), never the replacement


def test_renderer_known_answers():
    """Renderer.call (spair/spair.py:534-579): one opaque present object hides the background wherever its alpha is 1; an absent
    object (test form, sigmoid(logit) < 0.5 -> z_pres 1e-8) leaves the background; two equally deep objects average."""
    B, H, C = 1, 4, 3
    bg = torch.full((B, H, H, C), 0.25, dtype=torch.float64)
    obj = torch.zeros((B, 2, H, H, C + 1), dtype=torch.float64)
    obj[:, 0, ..., :C] = 0.75; obj[:, 0, ..., C] = 1.0            # object 0: opaque, colour 0.75
    obj[:, 1, ..., :C] = 0.5; obj[:, 1, ..., C] = 1e-8            # object 1: transparent
    zd = torch.zeros((B, 1, 2, 1), dtype=torch.float64)
    zp = torch.ones((B, 1, 2, 1), dtype=torch.float64)
    out = S.renderer(obj, bg, zd, zp, None, training=True)
    assert torch.allclose(out, torch.full_like(out, 0.75), atol=1e-6)
    logits = torch.tensor([-5.0, -5.0], dtype=torch.float64).reshape(B, 1, 2, 1)
    out = S.renderer(obj, bg, zd, None, logits, training=False)
    assert torch.allclose(out, bg, atol=1e-6)
    obj[:, 1, ..., C] = 1.0                                        # both opaque, same depth: importance-weighted mean
    out = S.renderer(obj, bg, zd, zp, None, training=True)
    assert torch.allclose(out, torch.full_like(out, 0.625), atol=1e-6)
    zd2 = torch.tensor([-20.0, 20.0], dtype=torch.float64).reshape(B, 1, 2, 1)     # object 0 in front: sigmoid(-z) + .5 = 1.5 vs .5
    out = S.renderer(obj, bg, zd2, zp, None, training=True)
    assert torch.allclose(out, torch.full_like(out, (1.5 * 0.75 + 0.5 * 0.5) / 2.0), atol=1e-6)


def test_inverse_stn_pastes_the_object_inside_its_box():
    """The renderer's STN (inverse form, spair/utils.py:158-162): a constant object lands on the canvas inside obj_bbox_mask
    and the canvas is ~0 well outside it only where the sampler's clipped corners coincide (its border quirk keeps the edge
    value elsewhere) -- here: the centre of the box carries the object's value."""
    B, Hc = 1, 4
    z = torch.zeros((B, Hc, Hc, 4), dtype=torch.float64)
    obj = torch.ones((B, Hc * Hc, 32, 32, 1), dtype=torch.float64)
    out, bbox = S.stn_forward(obj, z, 48, 48, inverse=True)
    assert tuple(out.shape) == (B, 16, 48, 48, 1)
    for cell in (0, 5, 15):
        y0, x0, y1, x1 = (bbox[0, cell] * 47).tolist()
        cy, cx = int(round((y0 + y1) / 2)), int(round((x0 + x1) / 2))
        assert abs(float(out[0, cell, cy, cx, 0]) - 1.0) < 1e-9
