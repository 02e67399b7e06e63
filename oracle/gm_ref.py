"""PyTorch-CPU restatement of the SPLIT-GMVAE (LGGMVae) train step of 51616/split-vae (config 3,
SURVEY 8a row A9 / 8f row F1).

TEST INFRASTRUCTURE ONLY (see oracle/np_ref.py header): imported by tests/ as the checker, never by
the product path in split_vae_amd/.

PARITY UNPINNED against TensorFlow 2.0 (not installable here).  Pinned by analytic identities
(tests/test_oracle_kat.py): kl_divergence_two_gauss(mu, sig, 0, 1) == kl_divergence(mu, sig); the
categorical term vanishes at uniform logits and is log(K) at a one-hot limit; the Gumbel-softmax rows
sum to one and tend to argmax(logits + g) as tau -> 0; dropout keeps E[x]; finite differences of the
whole loss against torch.autograd in float64.

What it follows (all randomness is an INPUT so that the HIP path can be compared on identical draws):
  Encoder(type='gmvae').__init__   vae/model.py:48-79     (layer/variable order, bias_initializer=1 of the sig heads)
  Encoder.call_gmvae               vae/model.py:116-135   (do1-4, do6, do7 exist but are never called)
  LGGMVae.__init__/call            vae/model.py:221-246   (14-tuple order)
  kl_divergence_two_gauss          vae/trainer.py:17-18
  train_step_lg_gm_vae             vae/trainer.py:146-173 (loss, 5 metrics)
  ExponentialDecay(1e6, 0.4, staircase) for lggmvae: vae/main.py:66-69 (host logic: optimizer.py)
TF-2.0 semantics encoded: Dropout(rate) in training = x * keep_mask / (1 - rate); elu(x) = x>0 ? x : exp(x)-1;
tf.random.uniform in [0,1); softmax over axis 1.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import np_ref, torch_ref

GM_RATE = 0.2           # Dropout(rate=0.2) of y_block and do5 (vae/model.py:56, :72)


def gm_param_shapes(H, W, global_latent=128, local_latent=128, y_size=30):
    """Trainable variables of LGGMVae in layer-tracking order: encoder_x (gmvae, 24 arrays), then the
    30 arrays LGVae shares (encoder_x_hat, decoder_x, decoder_x_hat: np_ref.param_shapes[10:])."""
    F_ = (H // 8) * (W // 8) * 128
    K, L = y_size, global_latent
    gm = [("encoder_x/h_block/conv2d/kernel", (6, 6, 3, 128)), ("encoder_x/h_block/conv2d/bias", (128,)),
          ("encoder_x/h_block/conv2d_1/kernel", (6, 6, 128, 128)), ("encoder_x/h_block/conv2d_1/bias", (128,)),
          ("encoder_x/h_block/conv2d_2/kernel", (4, 4, 128, 128)), ("encoder_x/h_block/conv2d_2/bias", (128,)),
          ("encoder_x/y_block/dense/kernel", (F_, 1024)), ("encoder_x/y_block/dense/bias", (1024,)),
          ("encoder_x/y_block/dense_1/kernel", (1024, 128)), ("encoder_x/y_block/dense_1/bias", (128,)),
          ("encoder_x/y_dense/kernel", (128, K)), ("encoder_x/y_dense/bias", (K,)),
          ("encoder_x/h_top_dense/kernel", (K, 512)), ("encoder_x/h_top_dense/bias", (512,)),
          ("encoder_x/z_prior_mean/kernel", (K, L)), ("encoder_x/z_prior_mean/bias", (L,)),
          ("encoder_x/z_prior_sig/kernel", (K, L)), ("encoder_x/z_prior_sig/bias", (L,)),
          ("encoder_x/e1/kernel", (F_, 512)), ("encoder_x/e1/bias", (512,)),
          ("encoder_x/z_mean/kernel", (512, L)), ("encoder_x/z_mean/bias", (L,)),
          ("encoder_x/z_sig/kernel", (512, L)), ("encoder_x/z_sig/bias", (L,))]
    rest = np_ref.param_shapes(H, W, global_latent, local_latent)[10:]
    return gm + list(rest)


def gm_glorot_init(H, W, seed=3, global_latent=128, local_latent=128, y_size=30, dtype=np.float32):
    """Keras defaults: Glorot-uniform kernels, zero biases, except z_prior_sig / z_sig biases = 1 (vae/model.py:68,:78)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for name, shp in gm_param_shapes(H, W, global_latent, local_latent, y_size):
        if name.endswith("kernel"):
            fan_in = int(np.prod(shp[:-1]))
            fan_out = int(np.prod(shp[:-2])) * shp[-1] if len(shp) == 4 else shp[-1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            out.append(rng.uniform(-lim, lim, size=shp).astype(dtype))
        else:
            one = name in ("encoder_x/z_prior_sig/bias", "encoder_x/z_sig/bias")
            out.append((np.ones if one else np.zeros)(shp, dtype=dtype))
    return out


def conv_elu(x, w, b, stride):
    return F.elu(torch_ref.conv2d_same(x, w, b, stride, None))


def gumbel_softmax(logits, u, tau):
    """vae/model.py:122: softmax((y_logits - log(-log(noise))) / tau, axis=1)."""
    return torch.softmax((logits - torch.log(-torch.log(u))) / tau, dim=1)


def encoder_gmvae(x, p, eps, u, keep1, keep5, tau, dropout=True):
    """Encoder.call_gmvae -- vae/model.py:116-135.  keep1 [B,1024], keep5 [B,F]: 0/1 dropout keep masks.

    `dropout`: whether y_block's Dropout(0.2) (:56) and do5 (:72, called at :129) act in this call.  The reference
    trains with model(images, training=True) (vae/trainer.py:149) but LGGMVae.call invokes self.encoder_x(x) WITHOUT
    forwarding `training` (:241) and call_gmvae's own default is False (:116).  Whether the flag reaches the two
    Dropout layers is a property of the Keras version [TF-2.0 semantics, not executable here]:
      * tensorflow 2.0.0 (the pinned version, requirements.txt:7): no call-context propagation of `training`; a layer
        called without it falls back to K.learning_phase() = 0 outside the Keras graph -> dropout=False even in training;
      * tensorflow >= 2.1: `training=True` of the outer Model call propagates to every nested layer that was called
        without the argument -> dropout=True in training.
    Both are restated; dropout=False ignores the masks."""
    if not dropout:
        keep1 = torch.ones_like(keep1) * (1.0 - GM_RATE)
        keep5 = torch.ones_like(keep5) * (1.0 - GM_RATE)
    h = conv_elu(x, p[0], p[1], 2)
    h = conv_elu(h, p[2], p[3], 2)
    h = conv_elu(h, p[4], p[5], 2)
    h = h.reshape(h.shape[0], -1)
    yh = F.elu(h @ p[6] + p[7])
    yh = yh * keep1 / (1.0 - GM_RATE)
    yh = F.elu(yh @ p[8] + p[9])
    y_logits = yh @ p[10] + p[11]
    y = gumbel_softmax(y_logits, u, tau)
    z_prior_mean = y @ p[14] + p[15]
    z_prior_sig = F.softplus(y @ p[16] + p[17])
    h_top = F.elu(y @ p[12] + p[13])
    he = F.elu((h * keep5 / (1.0 - GM_RATE)) @ p[18] + p[19])
    hh = he + h_top
    z_mean = hh @ p[20] + p[21]
    z_sig = F.softplus(hh @ p[22] + p[23])
    z = z_mean + z_sig * eps
    return z, z_mean, z_sig, y, y_logits, z_prior_mean, z_prior_sig


def lggmvae_forward(images, params, eps_x, eps_x_hat, u, keep1, keep5, tau=0.4, dropout=True):
    """LGGMVae.call -- vae/model.py:236-246; same 14-tuple order."""
    H, W = images.shape[1:3]
    x, x_hat = images[..., :3], images[..., 3:]
    z_x, z_mean_x, z_sig_x, y, y_logits, zpm, zps = encoder_gmvae(x, params[0:24], eps_x, u, keep1, keep5, tau, dropout)
    z_x_hat, z_mean_x_hat, z_sig_x_hat = torch_ref.encoder_conv(x_hat, params[24:34], eps_x_hat)
    x_mean, x_log_scale = torch_ref.decoder(torch.cat([z_x, z_x_hat], 1), params[34:44], H, W)
    x_hat_mean, x_hat_log_scale = torch_ref.decoder(z_x_hat, params[44:54], H, W)
    return (x_mean, x_log_scale, z_x, z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale, z_mean_x_hat,
            z_sig_x_hat, y, y_logits, zpm, zps)


def kl_divergence_two_gauss(mean1, sig1, mean2, sig2):
    """vae/trainer.py:17-18."""
    mean2 = torch.as_tensor(mean2, dtype=mean1.dtype)
    sig2 = torch.as_tensor(sig2, dtype=mean1.dtype)
    return torch.mean(torch.sum(torch.log(sig2) - torch.log(sig1)
                                + (torch.square(sig1) + torch.square(mean1 - mean2)) / (2 * torch.square(sig2)) - 0.5, dim=1))


def categorical_kl(y_logits, y_size):
    """vae/trainer.py:161-162."""
    py = torch.softmax(y_logits, dim=1)
    return torch.mean(torch.sum(py * (torch.log(py + 1e-8) - math.log(1.0 / y_size)), dim=1))


def lggmvae_losses(images, fwd, beta, alpha, y_size):
    """vae/trainer.py:152-165 (+ the five metrics of :169-173)."""
    (x_mean, x_log_scale, z_x, z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale, z_mean_x_hat, z_sig_x_hat,
     y, y_logits, zpm, zps) = fwd
    x, x_hat = images[..., :3], images[..., 3:]
    x_recon = torch_ref.discretised_logistic_loss(x, x_mean, x_log_scale).sum(dim=(1, 2, 3)).mean()
    x_hat_recon = torch_ref.discretised_logistic_loss(x_hat, x_hat_mean, x_hat_log_scale).sum(dim=(1, 2, 3)).mean()
    x_kl = kl_divergence_two_gauss(z_mean_x, z_sig_x, zpm, zps)
    x_hat_kl = kl_divergence_two_gauss(z_mean_x_hat, z_sig_x_hat, 0.0, 1.0)
    y_kl = categorical_kl(y_logits, y_size)
    total = x_recon + x_hat_recon + beta * (x_kl + x_hat_kl) + alpha * y_kl
    return dict(x_recon_loss=x_recon, x_kl_loss=x_kl, x_hat_recon_loss=x_hat_recon, x_hat_kl_loss=x_hat_kl,
                y_kl_loss=y_kl, total_loss=total)


class GMRefTrainer:
    """Stateful restatement of train_step_lg_gm_vae (vae/trainer.py:146-173) + Keras Adam."""

    def __init__(self, params, beta, alpha, y_size=30, tau=0.4, lr=1e-4, dtype=torch.float32, dropout=True):
        self.dropout = dropout
        self.params = [torch.as_tensor(p).to(dtype).clone().requires_grad_(True) for p in params]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0
        self.beta, self.alpha, self.y_size, self.tau, self.lr, self.dtype = float(beta), float(alpha), y_size, tau, lr, dtype

    def forward_losses(self, images, eps_x, eps_x_hat, u, keep1, keep5):
        c = lambda a: torch.as_tensor(a).to(self.dtype)
        images = c(images)
        fwd = lggmvae_forward(images, self.params, c(eps_x), c(eps_x_hat), c(u), c(keep1), c(keep5), self.tau, self.dropout)
        return fwd, lggmvae_losses(images, fwd, self.beta, self.alpha, self.y_size)

    def grads(self, *a):
        fwd, losses = self.forward_losses(*a)
        return fwd, losses, list(torch.autograd.grad(losses["total_loss"], self.params))

    def train_step(self, *a):
        fwd, losses, g = self.grads(*a)
        self.t += 1
        torch_ref.keras_adam_(self.params, g, self.m, self.v, self.t, self.lr)
        return {k: float(v.detach()) for k, v in losses.items()}, g
