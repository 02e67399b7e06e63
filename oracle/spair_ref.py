"""CPU restatement of the first SPLIT-SPAIR pieces (config 5) of 51616/split-vae.

TEST INFRASTRUCTURE ONLY (see oracle/np_ref.py header).  PARITY UNPINNED against TensorFlow 2.0 (not installable here);
pinned by the known-answer tests in tests/test_oracle_spair.py.

What is restated (the reference's first blockers for a MI355X SPLIT-SPAIR step):
  * the conv backbone of spair.Encoder (spair/spair.py:382-388, :411-416): 48 -> 24 -> 12 -> 4x4 cells, strides 2, 2, 3,
    then three 1x1 convs -- the only layers of the repo with non-power-of-two extents and a stride of 3;
  * STN.build / STN.call / STN.bilinear_sampler / STN.get_pixel_value (spair/utils.py:65-116, :119-200, :202-272,
    :274-330): affine grid from z_where, 4-tap bilinear gather with the corner indices clamped AFTER the weights' x1 = x0+1;
  * concrete_binary_sample_kl, compute_z_pres_kl_yolo_air, tf_safe_log, tf_mean_sum (spair/trainer.py:28-42, :45-94,
    :97-101, :107-109): the sequential 16-cell count-prior KL of z_pres.
torch float64/float32 functional code; layouts as the reference (NHWC, HWIO).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import torch_ref

BACKBONE = [  # name, k, stride, Cin, Cout   (spair/spair.py:382-388; all ReLU; SAME for the 4x4 convs, 1x1 VALID == SAME)
    ("conv1", 4, 2, 3, 128), ("conv2", 4, 2, 128, 128), ("conv3", 4, 3, 128, 128),
    ("z1", 1, 1, 128, 128), ("z2", 1, 1, 128, 128), ("z3", 1, 1, 128, 100),
]


def backbone_shapes():
    return [(n, (k, k, ci, co)) for n, k, s, ci, co in BACKBONE]


def backbone_init(seed=0, dtype=np.float32):
    """Glorot-uniform kernels, zero biases (Keras defaults)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for n, k, s, ci, co in BACKBONE:
        lim = math.sqrt(6.0 / (k * k * (ci + co)))
        out.append(rng.uniform(-lim, lim, size=(k, k, ci, co)).astype(dtype))
        out.append(np.zeros((co,), dtype))
    return out


def backbone_forward(x, params):
    """Encoder.call up to z (spair/spair.py:411-416): x [B,48,48,3] -> [h1 [B,24,24,128], h2 [B,12,12,128], h3 [B,4,4,128],
    z1, z2, z [B,4,4,100]] (all post-ReLU)."""
    outs, h = [], x
    for i, (n, k, s, ci, co) in enumerate(BACKBONE):
        h = torch_ref.conv2d_same(h, params[2 * i], params[2 * i + 1], s, "relu")
        outs.append(h)
    return outs


# --------------------------------------------------------------------------------------------- STN (spair/utils.py:47-330)
def stn_constants(H_obj, W_obj, H_out, W_out, dtype=torch.float64):
    """STN.build (:65-116): the normalised sampling grid [3, H_out*W_out] and the per-cell translation biases; the cell
    ratio (2*12)/48 is hard-coded in the reference (:102-103)."""
    xs = np.linspace(-1.0, 1.0, W_out)
    ys = np.linspace(-1.0, 1.0, H_out)
    X, Y = np.meshgrid(xs, ys)
    grid = torch.tensor(np.stack([X.reshape(-1), Y.reshape(-1), np.ones(H_out * W_out)]), dtype=torch.float32).to(dtype)
    ratio = (2.0 * 12) / 48
    bias_tx = np.zeros([H_obj, W_obj])
    bias_ty = np.zeros([H_obj, W_obj])
    for i in range(H_obj):
        i_p = (2. - ratio) * i / (H_obj - 1) - (1 - 0.5 * ratio)
        for j in range(W_obj):
            j_p = (2. - ratio) * j / (W_obj - 1) - (1 - 0.5 * ratio)
            bias_ty[i, j] = i_p
            bias_tx[i, j] = j_p
    f32 = lambda a: torch.tensor(a, dtype=torch.float32).to(dtype)
    return grid, f32(bias_tx), f32(bias_ty)


def bilinear_sampler(img, batch_grids, inverse=False):
    """STN.bilinear_sampler, forward (non-inverse) form (:202-272): img [B,H,W,C], batch_grids [B,B',2,Ho,Wo] (x then y in
    [-1,1]) -> [B,B',Ho,Wo,C].  Order of operations as the reference: x1 = floor(x)+1 BEFORE the clamps, so at the right /
    bottom border both corners clamp to the last pixel and the weights (x1-x), (x-x0) are taken from the CLAMPED corners."""
    if inverse:                                               # img [B,B',H,W,C]: every cell samples its own image (:290-308)
        B, Bp, H, W, C = img.shape
    else:
        B, H, W, C = img.shape
    x = batch_grids[:, :, 0]
    y = batch_grids[:, :, 1]
    x = 0.5 * (x + 1.0) * (W - 1)
    y = 0.5 * (y + 1.0) * (H - 1)
    x0 = torch.floor(x); x1 = x0 + 1
    y0 = torch.floor(y); y1 = y0 + 1
    x0 = torch.clamp(x0, 0., W - 1.); x1 = torch.clamp(x1, 0., W - 1.)
    y0 = torch.clamp(y0, 0., H - 1.); y1 = torch.clamp(y1, 0., H - 1.)
    wa = (x1 - x) * (y1 - y)
    wb = (x1 - x) * (y - y0)
    wc = (x - x0) * (y1 - y)
    wd = (x - x0) * (y - y0)
    xi0, xi1, yi0, yi1 = x0.long(), x1.long(), y0.long(), y1.long()
    bidx = torch.arange(B).view(B, 1, 1, 1).expand_as(xi0)
    if inverse:
        pidx = torch.arange(Bp).view(1, Bp, 1, 1).expand_as(xi0)
        g = lambda yy, xx: img[bidx, pidx, yy, xx]             # gather_nd on (b, b', y, x)
    else:
        g = lambda yy, xx: img[bidx, yy, xx]                   # get_pixel_value (:274-330): gather_nd on (b, y, x)
    Ia, Ib, Ic, Id = g(yi0, xi0), g(yi1, xi0), g(yi0, xi1), g(yi1, xi1)
    return wa[..., None] * Ia + wb[..., None] * Ib + wc[..., None] * Ic + wd[..., None] * Id


def stn_forward(x, z_where, H_out=32, W_out=32, inverse=False):
    """STN.call, forward form (:119-200): x [B,H,W,C], z_where [B,Hc,Wc,4] -> (glimpses [B,Hc*Wc,H_out,W_out,C],
    obj_bbox_mask [B,Hc*Wc,4])."""
    B, Hc, Wc, _ = z_where.shape
    grid, bias_tx, bias_ty = stn_constants(Hc, Wc, H_out, W_out, x.dtype)
    sx = 0.5 * torch.sigmoid(z_where[..., 0])
    sy = 0.5 * torch.sigmoid(z_where[..., 1])
    tx = 0.5 * torch.tanh(z_where[..., 2]) + bias_tx[None]
    ty = 0.5 * torch.tanh(z_where[..., 3]) + bias_ty[None]
    bh, bw = (sy / 2.0)[..., None], (sx / 2.0)[..., None]
    bty, btx = (ty[..., None] + 1.0) / 2.0, (tx[..., None] + 1.0) / 2.0
    bbox = torch.cat([bty - bh, btx - bw, bty + bh, btx + bw], dim=-1).reshape(B, Hc * Wc, 4)
    if inverse:                                               # the renderer's STN (:158-162)
        tx = -tx / (sx + 1e-5); ty = -ty / (sy + 1e-5)
        sx = 1 / (sx + 1e-5); sy = 1 / (sy + 1e-5)
    sx, sy, tx, ty = (t.reshape(B, Hc * Wc) for t in (sx, sy, tx, ty))
    zeros = torch.zeros_like(sx)
    A = torch.stack([torch.stack([sx, zeros, tx], dim=2), torch.stack([zeros, sy, ty], dim=2)], dim=2)     # [B,B',2,3]
    batch_grids = (A @ grid[None, None]).reshape(B, Hc * Wc, 2, H_out, W_out)
    return bilinear_sampler(x, batch_grids, inverse), bbox


# --------------------------------------------------------------------------------------------- z_pres KL (spair/trainer.py)
def tf_safe_log(value, replacement_value=-100.0):
    """:97-101."""
    lv = torch.log(value + 1e-8)
    bad = torch.isnan(lv) | torch.isinf(lv)
    return torch.where(bad, torch.full_like(lv, replacement_value), lv)


def tf_mean_sum(t):
    """:107-109: mean over the batch of the sum over everything else."""
    return t.reshape(t.shape[0], -1).sum(dim=1).mean()


def concrete_binary_sample_kl(pre_sigmoid_sample, prior_log_odds, prior_temperature, posterior_log_odds, posterior_temperature, eps=1e-8):
    """:28-42."""
    y = pre_sigmoid_sample
    ypt = y * prior_temperature
    log_prior = math.log(prior_temperature + eps) - ypt + prior_log_odds - 2.0 * torch.log(1.0 + torch.exp(-ypt + prior_log_odds) + eps)
    yqt = y * posterior_temperature
    log_post = math.log(posterior_temperature + eps) - yqt + posterior_log_odds - 2.0 * torch.log(1.0 + torch.exp(-yqt + posterior_log_odds) + eps)
    return log_post - log_prior


def compute_z_pres_kl_yolo_air(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob, temperature):
    """:45-94: cells visited in raster order; the prior odds of cell i follow from the count distribution conditioned on the
    objects switched on so far (z_pres > 0.5).  Inputs [B,H,W,1]."""
    B, H, W, _ = z_pres.shape
    dt = z_pres.dtype
    support = torch.arange(H * W + 1, dtype=dt)
    cpp = 1 - prior_prob
    dist = (1 - cpp) * (cpp ** support)
    dist = dist / torch.clamp(dist.sum(), min=1e-6)
    dist = dist[None, :].repeat(B, 1)
    so_far = torch.zeros((B, 1), dtype=dt)
    kls, i, nmax = [], 0, H * W
    for h in range(H):
        for w in range(W):
            p_z_given = torch.clamp(support[None, :] - so_far, min=0) / (nmax - i)
            p_z = (dist[:, None, :] @ p_z_given[:, :, None])[:, :, 0]
            prior_log_odds = tf_safe_log(p_z) - tf_safe_log(1 - p_z)
            kls.append(concrete_binary_sample_kl(z_pres_pre_sigmoid[:, h, w, :], prior_log_odds, temperature,
                                                 z_pres_logits[:, h, w, :], temperature))
            sample = (z_pres[:, h, w, :] > 0.5).to(dt)
            dist = (sample * p_z_given + (1 - sample) * (1 - p_z_given)) * dist
            dist = dist / torch.clamp(dist.sum(dim=1, keepdim=True), min=1e-6)
            so_far = so_far + sample
            i += 1
    return tf_mean_sum(torch.stack(kls, dim=1))       # [B, cells, 1]


# --------------------------------------------------------------------------------------------- Renderer (spair/spair.py:534-579)
def renderer(obj_full_recon_unnorm, background_img, z_depth, z_pres, z_pres_logits, training=False, noise=None, num_channel=3):
    """Renderer.call: obj_full_recon_unnorm [B,B',H,W,C+1] (every object's rgb + alpha on its own canvas, from the inverse STN),
    background_img [B,H,W,C], z_depth / z_pres / z_pres_logits [B,Hc,Wc,1] -> canvas_with_bg [B,H,W,C].
    training: z_pres as given and GaussianNoise(0.01) on the object images (`noise` = that N(0, 0.01) draw, or None for no
    noise); else z_pres = max(round(sigmoid(logits)), 1e-8) and no noise."""
    B, Bp = obj_full_recon_unnorm.shape[:2]
    if not training:
        z_pres = torch.sigmoid(z_pres_logits)
    z_depth = z_depth.reshape(B, Bp, 1, 1, 1)
    z_pres = z_pres.reshape(B, Bp, 1, 1, 1)
    if not training:
        z_pres = torch.maximum(torch.round(z_pres), torch.full_like(z_pres, 1e-8))
    obj_img = obj_full_recon_unnorm[..., :num_channel]
    obj_alpha = torch.clamp(obj_full_recon_unnorm[..., num_channel:], 1e-8, 1.0)
    transparency_map = z_pres * obj_alpha
    importance_map = z_pres * obj_alpha * (torch.sigmoid(-z_depth) + 0.5)
    if training and noise is not None:
        obj_img = obj_img + noise
    obj_img = torch.clamp(obj_img, 0.0, 1.0)
    unnorm_canvas = (importance_map * obj_img).sum(dim=1)
    normalise_const = importance_map.sum(dim=1)
    normalised_canvas = unnorm_canvas / (normalise_const + 1e-8)
    normalised_alpha_canvas = (transparency_map * importance_map).sum(dim=1) / (normalise_const + 1e-8)
    return normalised_alpha_canvas * normalised_canvas + (1.0 - normalised_alpha_canvas) * background_img
