"""NumPy float64 restatement of the SPLIT-VAE (LGVae) training path of 51616/split-vae.

TEST INFRASTRUCTURE ONLY.  Nothing under split_vae_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use oracle/ -- as the checker, never
as the product path.

PARITY UNPINNED: the reference's arithmetic lives in tensorflow_gpu==2.0.0 (requirements.txt:7),
which is not installable here (no wheel, no network) and the reference ships no tests, golden
vectors or fixtures for this path.  This restatement is therefore pinned by
(i) analytic known-answer tests (tests/test_oracle_kat.py), (ii) agreement with an independent
second restatement on different machinery (oracle/torch_ref.py: torch-CPU functional ops +
autograd), (iii) finite-difference gradient checks.  It is NOT pinned against TF outputs.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Layouts are the reference's: activations NHWC, conv kernels HWIO, dense kernels [in, out].
"""
import numpy as np

F = np.float64


# --------------------------------------------------------------------------- augmentation
def scramble(x, perm, size):
    """augmentation.py:43-57 (Augmentator.scramble) for ONE image x[H,W,C].

    extract_patches (VALID, stride=size) enumerates patches row-major; reshape gives
    patches[n] = x[pr*s:(pr+1)*s, pc*s:(pc+1)*s, :] with (pr,pc)=divmod(n,G); tf.random.shuffle
    permutes axis 0 -- here the permutation is the explicit input `perm` (patches' = patches[perm]);
    split/unstack/concat (:51-53) re-tiles row-major.  Returns concat([x, x_aug], axis=2) (:57).
    """
    H, W, C = x.shape
    s = int(size)
    G = W // s
    assert H == W and H % s == 0, "reference assumes square images, size | H (augmentation.py:44-46)"
    n_patch = H * W // (s * s)
    perm = np.asarray(perm).reshape(n_patch)
    x_aug = np.empty_like(x)
    for n in range(n_patch):
        r, c = divmod(n, G)
        pr, pc = divmod(int(perm[n]), G)
        x_aug[r * s:(r + 1) * s, c * s:(c + 1) * s, :] = x[pr * s:(pr + 1) * s, pc * s:(pc + 1) * s, :]
    return np.concatenate([x, x_aug], axis=2)


def scramble_batch(x, perm, size):
    """vae/main.py:57-61: the map is applied per image before batching."""
    return np.stack([scramble(x[b], perm[b], size) for b in range(x.shape[0])], 0)


# --------------------------------------------------------------------------- TF op semantics
def same_pads(n_in, k, s):
    """TF 'SAME' padding [TF-2.0 semantics]: out=ceil(in/s); pad=max((out-1)s+k-in,0); before=pad//2."""
    out = -(-n_in // s)
    pad = max((out - 1) * s + k - n_in, 0)
    return out, pad // 2, pad - pad // 2


def conv2d_same(x, w, b, stride, act=None):
    """tf.keras.layers.Conv2D(padding='same') as used at vae/model.py:36-38,:153-156.
    x[B,H,W,Ci] NHWC, w[kh,kw,Ci,Co] HWIO, b[Co]."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    oh, pt, pb = same_pads(H, kh, stride)
    ow, pl, pr = same_pads(W, kw, stride)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, Ci), F)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    out = np.zeros((B, oh, ow, Co), F)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride, :]
            out += patch @ w[i, j]
    out += b
    return activation(out, act)


def activation(x, act):
    if act is None:
        return x
    if act == 'relu':
        return np.maximum(x, 0.0)
    if act == 'softplus':
        return softplus(x)
    raise ValueError(act)


def softplus(x):
    """tf.nn.softplus = log(1+exp(x)), evaluated stably."""
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


def resize_bilinear_2x(x):
    """tf.image.resize(x, [2H,2W]) (vae/model.py:163,:165,:167): TF-2.0 default = bilinear,
    half-pixel centres, no antialias, edge clamp.  src = (dst+0.5)/2-0.5."""
    B, H, W, C = x.shape

    def idx(n_out, n_in):
        src = (np.arange(n_out, dtype=F) + 0.5) * (n_in / n_out) - 0.5
        lo = np.floor(src)
        frac = src - lo
        i0 = np.clip(lo, 0, n_in - 1).astype(int)
        i1 = np.clip(lo + 1, 0, n_in - 1).astype(int)
        return i0, i1, frac

    y0, y1, fy = idx(2 * H, H)
    x0, x1, fx = idx(2 * W, W)
    top = x[:, y0][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y0][:, :, x1] * fx[None, None, :, None]
    bot = x[:, y1][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y1][:, :, x1] * fx[None, None, :, None]
    return top * (1 - fy)[None, :, None, None] + bot * fy[None, :, None, None]


def dense(x, w, b, act=None):
    """tf.keras.layers.Dense: x@w+b, w[in,out]."""
    return activation(x @ w + b, act)


# --------------------------------------------------------------------------- model
ENC_NAMES = ['e1', 'e2', 'e3', 'e4_mean', 'e4_sd']
DEC_NAMES = ['d1', 'd2', 'd3', 'd4', 'd5']


def param_shapes(H, W, global_latent=128, local_latent=128):
    """The 40 trainable variables in Keras creation order (kernel then bias per layer):
    encoder_x, encoder_x_hat, decoder_x, decoder_x_hat (vae/model.py:182-186; layers :36-42,:152-156)."""
    shapes = []

    def enc(prefix, latent):
        flat = (H // 8) * (W // 8) * 128
        for name, shp in [('e1', (6, 6, 3, 32)), ('e2', (6, 6, 32, 64)), ('e3', (4, 4, 64, 128)),
                          ('e4_mean', (flat, latent)), ('e4_sd', (flat, latent))]:
            shapes.append((prefix + '/' + name + '/kernel', shp))
            shapes.append((prefix + '/' + name + '/bias', (shp[-1],)))

    def dec(prefix, latent):
        # vae/model.py:152: image_shape[1]//8*image_shape[2]//8*128 == ((H//8)*W)//8*128
        d1_out = ((H // 8) * W) // 8 * 128
        for name, shp in [('d1', (latent, d1_out)), ('d2', (4, 4, 128, 128)), ('d3', (4, 4, 128, 64)),
                          ('d4', (6, 6, 64, 32)), ('d5', (6, 6, 32, 6))]:
            shapes.append((prefix + '/' + name + '/kernel', shp))
            shapes.append((prefix + '/' + name + '/bias', (shp[-1],)))

    enc('encoder_x', global_latent)
    enc('encoder_x_hat', local_latent)
    dec('decoder_x', global_latent + local_latent)
    dec('decoder_x_hat', local_latent)
    return shapes


def glorot_init(H, W, seed=3, global_latent=128, local_latent=128, dtype=np.float32):
    """Keras defaults: glorot_uniform kernels (limit=sqrt(6/(fan_in+fan_out)), conv fans include
    kh*kw), zero biases [TF-2.0 semantics]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    params = []
    for name, shp in param_shapes(H, W, global_latent, local_latent):
        if name.endswith('bias'):
            params.append(np.zeros(shp, dtype))
        else:
            if len(shp) == 4:
                rf = shp[0] * shp[1]
                fan_in, fan_out = rf * shp[2], rf * shp[3]
            else:
                fan_in, fan_out = shp
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            params.append(rng.uniform(-lim, lim, size=shp).astype(dtype))
    return params


def encoder_conv(x, p, eps):
    """Encoder.call_conv, vae/model.py:100-114 (+ Sampling :9-13).  p = 10 arrays of one encoder."""
    h = conv2d_same(x, p[0], p[1], 2, 'relu')
    h = conv2d_same(h, p[2], p[3], 2, 'relu')
    h = conv2d_same(h, p[4], p[5], 2, 'relu')
    f = h.reshape(h.shape[0], -1)  # Flatten: (h,w,c) order
    z_mean = dense(f, p[6], p[7])
    z_sig = dense(f, p[8], p[9], 'softplus')
    z = z_mean + z_sig * eps
    return z, z_mean, z_sig


def decoder(z, p, H, W):
    """Decoder.call, vae/model.py:158-169."""
    h = dense(z, p[0], p[1], 'relu')
    h = h.reshape(-1, H // 8, W // 8, 128)
    h = conv2d_same(h, p[2], p[3], 1, 'relu')
    h = resize_bilinear_2x(h)
    h = conv2d_same(h, p[4], p[5], 1, 'relu')
    h = resize_bilinear_2x(h)
    h = conv2d_same(h, p[6], p[7], 1, 'relu')
    h = resize_bilinear_2x(h)
    h = conv2d_same(h, p[8], p[9], 1, None)
    return h[..., :3], h[..., 3:]


def lgvae_forward(images, params, eps_x, eps_x_hat):
    """LGVae.call, vae/model.py:189-200.  Returns the reference's 10-tuple, same order."""
    params = [np.asarray(p, F) for p in params]
    images = np.asarray(images, F)
    H, W = images.shape[1:3]
    x, x_hat = images[..., :3], images[..., 3:]
    z_x, z_mean_x, z_sig_x = encoder_conv(x, params[0:10], np.asarray(eps_x, F))
    z_x_hat, z_mean_x_hat, z_sig_x_hat = encoder_conv(x_hat, params[10:20], np.asarray(eps_x_hat, F))
    x_mean, x_log_scale = decoder(np.concatenate([z_x, z_x_hat], 1), params[20:30], H, W)
    x_hat_mean, x_hat_log_scale = decoder(z_x_hat, params[30:40], H, W)
    return (x_mean, x_log_scale, z_x, z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale,
            z_mean_x_hat, z_sig_x_hat)


# --------------------------------------------------------------------------- losses
def kl_divergence(z_mean, z_sig):
    """vae/trainer.py:11-15."""
    z_log_var = np.log(np.square(z_sig))
    return np.mean(-0.5 * np.sum(1 + z_log_var - np.square(z_mean) - np.exp(z_log_var), axis=1))


def kl_divergence_two_gauss(mean1, sig1, mean2, sig2):
    """vae/trainer.py:17-18."""
    return np.mean(np.sum(np.log(sig2) - np.log(sig1)
                          + (np.square(sig1) + np.square(mean1 - mean2)) / (2 * np.square(sig2)) - 0.5, axis=1))


def discretised_logistic_loss(x, m, log_scales):
    """vae/trainer.py:21-38, element-wise negative log-probability."""
    centered_x = x - m
    inv_stdv = np.exp(-log_scales)
    plus_in = inv_stdv * (centered_x + 1. / 255.)
    min_in = inv_stdv * (centered_x - 1. / 255.)
    cdf_plus = sigmoid(plus_in)
    cdf_min = sigmoid(min_in)
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv_stdv * centered_x
    log_pdf_mid = mid_in - log_scales - 2. * softplus(mid_in)
    log_cdf_plus = plus_in - softplus(plus_in)
    log_one_minus_cdf_min = -softplus(min_in)
    log_prob = np.where(x < -0.999, log_cdf_plus,
                        np.where(x > 0.999, log_one_minus_cdf_min,
                                 np.where(cdf_delta > 1e-5, np.log(np.maximum(cdf_delta, 1e-12)),
                                          log_pdf_mid - np.log(127.5))))
    return -log_prob


def lgvae_losses(images, fwd, beta):
    """vae/trainer.py:125-135: the five scalars of train_step_lg_vae + total."""
    (x_mean, x_log_scale, z_x, z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale,
     z_mean_x_hat, z_sig_x_hat) = fwd
    images = np.asarray(images, F)
    x, x_hat = images[..., :3], images[..., 3:]
    x_recon = np.mean(np.sum(discretised_logistic_loss(x, x_mean, x_log_scale), axis=(1, 2, 3)))
    x_hat_recon = np.mean(np.sum(discretised_logistic_loss(x_hat, x_hat_mean, x_hat_log_scale), axis=(1, 2, 3)))
    total_kl = beta * kl_divergence(np.concatenate([z_mean_x, z_mean_x_hat], 1),
                                    np.concatenate([z_sig_x, z_sig_x_hat], 1))
    x_kl = kl_divergence(z_mean_x, z_sig_x)
    x_hat_kl = kl_divergence(z_mean_x_hat, z_sig_x_hat)
    total = x_recon + x_hat_recon + total_kl
    return dict(x_recon_loss=x_recon, x_kl_loss=x_kl, x_hat_recon_loss=x_hat_recon,
                x_hat_kl_loss=x_hat_kl, total_kl_loss=total_kl, total_loss=total)


def total_loss(images, params, eps_x, eps_x_hat, beta):
    return lgvae_losses(images, lgvae_forward(images, params, eps_x, eps_x_hat), beta)['total_loss']


# --------------------------------------------------------------------------- optimiser
def keras_adam_step(params, grads, m, v, t, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam (vae/main.py:65) = ResourceApplyAdam [TF-2.0 semantics]:
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g*g-v)(1-b2);
    var -= alpha*m/(sqrt(v)+eps)   (eps OUTSIDE the bias correction; t = iterations+1)."""
    alpha = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    out_p, out_m, out_v = [], [], []
    for p, g, mi, vi in zip(params, grads, m, v):
        mi = mi + (g - mi) * (1 - beta1)
        vi = vi + (g * g - vi) * (1 - beta2)
        out_p.append(p - alpha * mi / (np.sqrt(vi) + eps))
        out_m.append(mi)
        out_v.append(vi)
    return out_p, out_m, out_v


def fd_grad(images, params, eps_x, eps_x_hat, beta, which, index, h=1e-5):
    """Central finite difference of total_loss w.r.t. params[which].flat[index] (float64)."""
    ps = [np.array(p, F) for p in params]
    orig = ps[which].flat[index]
    ps[which].flat[index] = orig + h
    lp = total_loss(images, ps, eps_x, eps_x_hat, beta)
    ps[which].flat[index] = orig - h
    lm = total_loss(images, ps, eps_x, eps_x_hat, beta)
    return (lp - lm) / (2 * h)
