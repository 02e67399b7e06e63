"""PyTorch-CPU restatement of the SPLIT-VAE (LGVae) train step of 51616/split-vae.

TEST INFRASTRUCTURE ONLY (see oracle/np_ref.py header).  Used by tests/ as the checker, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg (kind "port").  The product path in
split_vae_amd/ never imports it.

PARITY UNPINNED against TensorFlow 2.0 (not installable here); pinned by KATs, by agreement with
the independent NumPy-float64 restatement (oracle/np_ref.py) and by finite differences.

Second, independent machinery: torch functional ops (F.conv2d on explicitly padded NCHW views,
F.interpolate(bilinear, align_corners=False), F.softplus) + torch.autograd for the 40 gradients.
All public functions take/return the reference's layouts (NHWC activations, HWIO conv kernels,
[in,out] dense kernels); the NCHW permutes are internal.
"""
import math
import torch
import torch.nn.functional as F

from . import np_ref


def _same_pad(x_nchw, k, s):
    """TF SAME padding, asymmetric (extra on bottom/right) [TF-2.0 semantics]."""
    H, W = x_nchw.shape[2:]
    _, pt, pb = np_ref.same_pads(H, k, s)
    _, pl, pr = np_ref.same_pads(W, k, s)
    return F.pad(x_nchw, (pl, pr, pt, pb))


def conv2d_same(x_nhwc, w_hwio, b, stride, act=None):
    """Conv2D(padding='same') -- vae/model.py:36-38, :153-156."""
    x = x_nhwc.permute(0, 3, 1, 2)
    w = w_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(_same_pad(x, w_hwio.shape[0], stride), w, b, stride=stride)
    y = y.permute(0, 2, 3, 1)
    if act == 'relu':
        y = F.relu(y)
    return y


def resize_bilinear_2x(x_nhwc):
    """tf.image.resize default (bilinear, half-pixel centres) -- vae/model.py:163-167."""
    x = x_nhwc.permute(0, 3, 1, 2)
    y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
    return y.permute(0, 2, 3, 1)


def encoder_conv(x, p, eps):
    """Encoder.call_conv -- vae/model.py:100-114; Sampling -- :9-13."""
    h = conv2d_same(x, p[0], p[1], 2, 'relu')
    h = conv2d_same(h, p[2], p[3], 2, 'relu')
    h = conv2d_same(h, p[4], p[5], 2, 'relu')
    f = h.reshape(h.shape[0], -1)
    z_mean = f @ p[6] + p[7]
    z_sig = F.softplus(f @ p[8] + p[9])
    return z_mean + z_sig * eps, z_mean, z_sig


def decoder(z, p, H, W):
    """Decoder.call -- vae/model.py:158-169."""
    h = F.relu(z @ p[0] + p[1]).reshape(-1, H // 8, W // 8, 128)
    h = conv2d_same(h, p[2], p[3], 1, 'relu')
    h = conv2d_same(resize_bilinear_2x(h), p[4], p[5], 1, 'relu')
    h = conv2d_same(resize_bilinear_2x(h), p[6], p[7], 1, 'relu')
    h = conv2d_same(resize_bilinear_2x(h), p[8], p[9], 1, None)
    return h[..., :3], h[..., 3:]


def lgvae_forward(images, params, eps_x, eps_x_hat):
    """LGVae.call -- vae/model.py:189-200; same 10-tuple order."""
    H, W = images.shape[1:3]
    x, x_hat = images[..., :3], images[..., 3:]
    z_x, z_mean_x, z_sig_x = encoder_conv(x, params[0:10], eps_x)
    z_x_hat, z_mean_x_hat, z_sig_x_hat = encoder_conv(x_hat, params[10:20], eps_x_hat)
    x_mean, x_log_scale = decoder(torch.cat([z_x, z_x_hat], 1), params[20:30], H, W)
    x_hat_mean, x_hat_log_scale = decoder(z_x_hat, params[30:40], H, W)
    return (x_mean, x_log_scale, z_x, z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale,
            z_mean_x_hat, z_sig_x_hat)


def kl_divergence(z_mean, z_sig):
    """vae/trainer.py:11-15."""
    z_log_var = torch.log(torch.square(z_sig))
    return torch.mean(-0.5 * torch.sum(1 + z_log_var - torch.square(z_mean) - torch.exp(z_log_var), dim=1))


def discretised_logistic_loss(x, m, log_scales):
    """vae/trainer.py:21-38."""
    centered_x = x - m
    inv_stdv = torch.exp(-log_scales)
    plus_in = inv_stdv * (centered_x + 1. / 255.)
    min_in = inv_stdv * (centered_x - 1. / 255.)
    cdf_plus = torch.sigmoid(plus_in)
    cdf_min = torch.sigmoid(min_in)
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv_stdv * centered_x
    log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    log_prob = torch.where(x < -0.999, log_cdf_plus,
                           torch.where(x > 0.999, log_one_minus_cdf_min,
                                       torch.where(cdf_delta > 1e-5,
                                                   torch.log(torch.clamp(cdf_delta, min=1e-12)),
                                                   log_pdf_mid - math.log(127.5))))
    return -log_prob


def lgvae_losses(images, fwd, beta):
    """vae/trainer.py:125-135."""
    (x_mean, x_log_scale, z_x, z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale,
     z_mean_x_hat, z_sig_x_hat) = fwd
    x, x_hat = images[..., :3], images[..., 3:]
    x_recon = discretised_logistic_loss(x, x_mean, x_log_scale).sum(dim=(1, 2, 3)).mean()
    x_hat_recon = discretised_logistic_loss(x_hat, x_hat_mean, x_hat_log_scale).sum(dim=(1, 2, 3)).mean()
    total_kl = beta * kl_divergence(torch.cat([z_mean_x, z_mean_x_hat], 1), torch.cat([z_sig_x, z_sig_x_hat], 1))
    x_kl = kl_divergence(z_mean_x, z_sig_x)
    x_hat_kl = kl_divergence(z_mean_x_hat, z_sig_x_hat)
    total = x_recon + x_hat_recon + total_kl
    return dict(x_recon_loss=x_recon, x_kl_loss=x_kl, x_hat_recon_loss=x_hat_recon,
                x_hat_kl_loss=x_hat_kl, total_kl_loss=total_kl, total_loss=total)


def keras_adam_(params, grads, m, v, t, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7):
    """In-place Keras/TF-2.0 Adam (ResourceApplyAdam): eps outside the bias correction."""
    alpha = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    with torch.no_grad():
        for p, g, mi, vi in zip(params, grads, m, v):
            mi.add_((g - mi) * (1 - beta1))
            vi.add_((g * g - vi) * (1 - beta2))
            p.sub_(alpha * mi / (vi.sqrt() + eps))


class RefTrainer:
    """Stateful restatement of train_step_lg_vae (vae/trainer.py:120-144) + Adam (vae/main.py:65)."""

    def __init__(self, params, beta, lr=1e-4, dtype=torch.float32):
        self.params = [torch.as_tensor(p).to(dtype).clone().requires_grad_(True) for p in params]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0
        self.beta = float(beta)
        self.lr = lr
        self.dtype = dtype

    def forward_losses(self, images, eps_x, eps_x_hat):
        images = torch.as_tensor(images).to(self.dtype)
        fwd = lgvae_forward(images, self.params, torch.as_tensor(eps_x).to(self.dtype),
                            torch.as_tensor(eps_x_hat).to(self.dtype))
        return fwd, lgvae_losses(images, fwd, self.beta)

    def grads(self, images, eps_x, eps_x_hat):
        fwd, losses = self.forward_losses(images, eps_x, eps_x_hat)
        g = torch.autograd.grad(losses['total_loss'], self.params)
        return fwd, losses, list(g)

    def train_step(self, images, eps_x, eps_x_hat):
        fwd, losses, g = self.grads(images, eps_x, eps_x_hat)
        self.t += 1
        keras_adam_(self.params, g, self.m, self.v, self.t, self.lr)
        return {k: float(val.detach()) for k, val in losses.items()}, g


def scramble_batch(x, perm, size):
    """augmentation.py:43-57 on a batch via pure index arithmetic (torch gather):
    x_aug[r*s+i, c*s+j] = x[pr*s+i, pc*s+j], (pr,pc)=divmod(perm[r*G+c], G)."""
    x = torch.as_tensor(x)
    perm = torch.as_tensor(perm).long()
    B, H, W, C = x.shape
    s = int(size)
    G = W // s
    rr = torch.arange(H)
    cc = torch.arange(W)
    n = (rr[:, None] // s) * G + (cc[None, :] // s)             # destination patch id [H,W]
    src = perm[:, n.reshape(-1)].reshape(B, H, W)                  # source patch id
    sy = (src // G) * s + (rr[:, None] % s)
    sx = (src % G) * s + (cc[None, :] % s)
    bidx = torch.arange(B)[:, None, None].expand(B, H, W)
    x_aug = x[bidx, sy, sx]
    return torch.cat([x, x_aug], dim=3)
