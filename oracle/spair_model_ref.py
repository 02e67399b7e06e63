"""CPU restatement of the SPAIR / SPLIT-SPAIR models and their training losses (config 5).

TEST INFRASTRUCTURE ONLY (see oracle/np_ref.py header): imported by tests/ only.  PARITY UNPINNED against TensorFlow 2.0
(not installable here); the operator pieces it composes are pinned by the known-answer tests of tests/test_oracle_spair.py,
the composition by tests/test_oracle_spair_model.py (shapes, variable counts, hand-checked loss terms).

Restates, functionally over a {name: tensor} dictionary (torch float64 / float32, NHWC, HWIO, Dense [in, out]):
  spair/spair.py   SPAIR :19-49, LGSPAIR :52-106, ImageEncoder[Dense] :110-154, ImageDecoder[Dense] :157-202,
                   BackgroundModel :205-244, ObjEncoder :246-273, ObjDecoder :341-366, Encoder :368-496, Decoder :500-532
                   (Renderer :534-579 and the STN live in oracle/spair_ref.py)
  spair/utils.py   concrete_binary_pre_sigmoid_sample :14-17, Sampling :19-24
  spair/trainer.py kl_divergence :13-21, kl_divergence_two_gauss :23-24, xent_loss :103-104, train_step's loss assembly :136-234
Every random draw of the reference (tf.random.normal / uniform, GaussianNoise) is an explicit `noise` entry so that the device
path can be fed the same numbers.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import spair_ref, torch_ref

N_WHERE, N_DEPTH, N_PRES, N_PASS = 4, 1, 1, 8           # spair/spair.py:374-376, :390
CELLS = 4                                               # 48x48 canvas -> 4x4 cells (strides 2, 2, 3; spair/spair.py:382-384)


class Cfg(dict):
    """spair/utils.py:7-11 dotdict: unknown keys read as None (config.bg_model / config.concat_z_bg are never set by a flag)."""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


def default_config(**kw):
    """spair/main.py:19-50 defaults + the image shapes get_cub_dataset reports (spair/data.py:258-278: 48x48x3)."""
    c = Cfg(learning_rate=1e-4, beta=0.5, channel=3, batch_size=32, tau=0.8, object_size=32, latent_size=128, anneal_until=1.0,
            z_pres_anneal_step=10000.0, prior_z_zoom=0.0, prior_z_zoom_start=10.0, reconstruction_weight=1.0, bg_latent_size=4,
            local_latent_size=64, z_bg_beta=10.0, z_l_beta=0.1, z_what_beta=0.1, model="spair", patch_size=4,
            augmentation="scramble", split_z_l=False, dense_bg=False, dense_local=False, concat_bg=False, concat_z_what=False,
            concat_backbone=False, image_size=[48, 48, 3], test_size=[48, 48, 3])
    c.update(kw)
    return c


# ------------------------------------------------------------------------------------------------ variables
def _image_encoder_spec(pfx, image_size, latent, dense):
    H, W, C = image_size
    if dense:                                                                        # ImageEncoderDense :135-154
        return [(pfx + "/e1", (H * W * C, 1024)), (pfx + "/e2", (1024, 500)), (pfx + "/z_mu", (500, latent)), (pfx + "/z_sigma", (500, latent))]
    F_ = ((H + 7) // 8) * ((W + 7) // 8) * 128                                       # ImageEncoder :110-132
    return [(pfx + "/e1", (3, 3, C, 32)), (pfx + "/e2", (3, 3, 32, 64)), (pfx + "/e3", (3, 3, 64, 128)), (pfx + "/z_mu", (F_, latent)),
            (pfx + "/z_sigma", (F_, latent))]


def _image_decoder_spec(pfx, image_size, n_in, dense):
    H, W, C = image_size
    if dense:                                                                        # ImageDecoderDense :185-202
        return [(pfx + "/d1", (n_in, 500)), (pfx + "/d2", (500, 1024)), (pfx + "/d3", (1024, H * W * C))]
    return [(pfx + "/d1", (n_in, H // 8 * W // 8 * 128)), (pfx + "/d2", (3, 3, 128, 128)), (pfx + "/d3", (3, 3, 128, 64)),   # :157-182
            (pfx + "/d4", (3, 3, 64, 32)), (pfx + "/d5", (3, 3, 32, C))]


def param_spec(cfg):
    """[(name, kernel shape)] in the layer-tracking order of model.trainable_variables; every layer also has a bias
    `name + "/bias"` of the kernel's last extent, directly after its kernel."""
    L, Ll, Lbg, C = cfg.latent_size, cfg.local_latent_size, cfg.bg_latent_size, cfg.image_size[2]
    lg = cfg.model == "lg_spair"
    concat = bool(lg and cfg.concat_backbone)
    Fv = 100 + (16 if concat else 0)
    S = cfg.object_size
    enc = [("encoder/conv1", (4, 4, C, 128)), ("encoder/conv2", (4, 4, 128, 128)), ("encoder/conv3", (4, 4, 128, 128)),
           ("encoder/z1", (1, 1, 128, 128)), ("encoder/z2", (1, 1, 128, 128)), ("encoder/z3", (1, 1, 128, 100)),
           ("encoder/dense_z_where/0", (Fv, 128)), ("encoder/dense_z_where/1", (128, 64)), ("encoder/dense_z_where/2", (64, 2 * N_WHERE + N_PASS)),
           ("encoder/dense_z_depth/0", (Fv + N_PASS + N_WHERE + L, 64)), ("encoder/dense_z_depth/1", (64, 2 * N_DEPTH + N_PASS)),
           ("encoder/dense_z_pres/0", (Fv + N_PASS + N_WHERE + L + N_DEPTH, 64)), ("encoder/dense_z_pres/1", (64, N_PRES)),
           ("encoder/obj_encoder/conv1", (3, 3, C, 32)), ("encoder/obj_encoder/conv2", (3, 3, 32, 64)),
           ("encoder/obj_encoder/dense1", ((S // 4) * (S // 4) * 64, 2 * L)), ("encoder/obj_encoder/z_what_mu", (2 * L, L)),
           ("encoder/obj_encoder/z_what_sigma", (2 * L, L))]
    if concat:
        enc += [("encoder/dense_z_l/0", (Ll, 16)), ("encoder/dense_z_l/1", (16, 16))]
    Lw = L + (Ll if (lg and cfg.concat_z_what) else 0)
    dec = [("decoder/obj_decoder/d0", (Lw, 2 * L)), ("decoder/obj_decoder/d1", (2 * L, S // 4 * S // 4 * 32)),
           ("decoder/obj_decoder/d2", (3, 3, 32, 64)), ("decoder/obj_decoder/d3", (3, 3, 64, 32)), ("decoder/obj_decoder/d5", (3, 3, 32, C + 1))]
    spec = enc + dec
    if cfg.model == "bg_spair":                                                      # BackgroundModel :205-244 (conv encoder + conv decoder)
        spec += [(n.replace("/z_mu", "/z_bg_mu").replace("/z_sigma", "/z_bg_sigma"), s)
                 for n, s in _image_encoder_spec("bg_model", cfg.image_size, Lbg, False)]
        spec += _image_decoder_spec("bg_model", cfg.image_size, Lbg, False)
    if lg:                                                                           # LGSPAIR.__init__ :66-82
        spec += _image_encoder_spec("bg_encoder", cfg.image_size, Lbg, cfg.dense_bg)
        spec += _image_decoder_spec("bg_decoder", cfg.image_size, Lbg + (Ll if cfg.concat_z_bg else 0), cfg.dense_bg)
        spec += _image_encoder_spec("x_hat_encoder", cfg.image_size, Ll, cfg.dense_local)
        spec += _image_decoder_spec("x_hat_decoder", cfg.image_size, Ll, cfg.dense_local)
    return spec


def init_params(cfg, seed=0, dtype=torch.float64):
    """Keras defaults: Glorot-uniform kernels, zero biases.  {name: tensor}, insertion order = variable order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in param_spec(cfg):
        rf = shp[0] * shp[1] if len(shp) == 4 else 1
        fan_in, fan_out = rf * shp[-2], rf * shp[-1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        out[name] = torch.tensor(rng.uniform(-lim, lim, size=shp).astype(np.float32)).to(dtype)
        out[name + "/bias"] = torch.zeros((shp[-1],), dtype=dtype)
    return out


def noise_spec(cfg, B):
    """{name: (shape, kind)} of every random draw of one training forward; kind 'normal' | 'uniform' | 'normal0.01'."""
    L, Ll, Lbg = cfg.latent_size, cfg.local_latent_size, cfg.bg_latent_size
    H, W, C = cfg.image_size
    n = B * CELLS * CELLS
    d = {"eps_where": ((n, N_WHERE), "normal"), "eps_what": ((n, L), "normal"), "eps_depth": ((n, N_DEPTH), "normal"),
         "u_pres": ((n, N_PRES), "uniform"), "render": ((B, CELLS * CELLS, H, W, C), "normal0.01")}
    if cfg.model == "bg_spair":
        d["eps_bg"] = ((B, Lbg), "normal")
    if cfg.model == "lg_spair":
        d["eps_bg"] = ((B, Lbg), "normal")
        d["eps_l"] = ((B, Ll), "normal")
    return d


def draw_noise(cfg, B, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, (shp, kind) in noise_spec(cfg, B).items():
        if kind == "uniform":
            out[k] = (torch.rand(shp, generator=g, dtype=torch.float32) * 0.98 + 0.01).to(dtype)
        else:
            out[k] = (torch.randn(shp, generator=g, dtype=torch.float32) * (0.01 if kind == "normal0.01" else 1.0)).to(dtype)
    return out


# ------------------------------------------------------------------------------------------------ layers
def dense(p, name, x, act=None):
    y = x @ p[name] + p[name + "/bias"]
    if act == "relu":
        return F.relu(y)
    if act == "softplus":
        return F.softplus(y)
    if act == "sigmoid":
        return torch.sigmoid(y)
    return y


def conv(p, name, x, stride, act=None):
    y = torch_ref.conv2d_same(x, p[name], p[name + "/bias"], stride, "relu" if act == "relu" else None)
    return torch.sigmoid(y) if act == "sigmoid" else y


def image_encoder(p, pfx, x, eps, is_dense, mu="z_mu", sigma="z_sigma"):
    """ImageEncoder.call :124-132 / ImageEncoderDense.call :145-154 -> z, z_mean, z_sig."""
    B = x.shape[0]
    if is_dense:
        h = dense(p, pfx + "/e2", dense(p, pfx + "/e1", x.reshape(B, -1), "relu"), "relu")
    else:
        h = conv(p, pfx + "/e3", conv(p, pfx + "/e2", conv(p, pfx + "/e1", x, 2, "relu"), 2, "relu"), 2, "relu").reshape(B, -1)
    z_mean = dense(p, f"{pfx}/{mu}", h)
    z_sig = dense(p, f"{pfx}/{sigma}", h, "softplus")
    return z_mean + z_sig * eps, z_mean, z_sig


def image_decoder(p, pfx, z, image_size, is_dense):
    """ImageDecoder.call :171-182 / ImageDecoderDense.call :196-202."""
    H, W, C = image_size
    if is_dense:
        return dense(p, pfx + "/d3", dense(p, pfx + "/d2", dense(p, pfx + "/d1", z, "relu"), "relu"), "sigmoid").reshape(-1, H, W, C)
    x = dense(p, pfx + "/d1", z, "relu").reshape(-1, H // 8, W // 8, 128)
    x = torch_ref.resize_bilinear_2x(conv(p, pfx + "/d2", x, 1, "relu"))
    x = torch_ref.resize_bilinear_2x(conv(p, pfx + "/d3", x, 1, "relu"))
    x = torch_ref.resize_bilinear_2x(conv(p, pfx + "/d4", x, 1, "sigmoid"))
    return conv(p, pfx + "/d5", x, 1, "sigmoid")


def encoder(p, cfg, x, z_l, noise):
    """Encoder.call :403-496 (glimpse_local=False).  x [B,48,48,C]; z_l [B,Ll] or None (concat_backbone)."""
    B = x.shape[0]
    L, S = cfg.latent_size, cfg.object_size
    h = conv(p, "encoder/conv1", x, 2, "relu")
    h = conv(p, "encoder/conv2", h, 2, "relu")
    h = conv(p, "encoder/conv3", h, 3, "relu")
    h = conv(p, "encoder/z1", h, 1, "relu")
    h = conv(p, "encoder/z2", h, 1, "relu")
    z = conv(p, "encoder/z3", h, 1, "relu")
    Hc, Wc = z.shape[1], z.shape[2]
    fv = z.reshape(-1, z.shape[-1])
    if z_l is not None:
        zl = dense(p, "encoder/dense_z_l/1", dense(p, "encoder/dense_z_l/0", z_l, "relu"), "relu")
        zl = zl[:, None, :].repeat(1, 16, 1).reshape(-1, zl.shape[-1])                # tf.tile(...,[1,16,1]) :408 (hard-coded 16 cells)
        fv = torch.cat([fv, zl], dim=-1)
    # box network :424-437
    o = dense(p, "encoder/dense_z_where/2", dense(p, "encoder/dense_z_where/1", dense(p, "encoder/dense_z_where/0", fv, "relu"), "relu"))
    zw_mean, zw_sig, f1 = o[:, :N_WHERE], o[:, N_WHERE:2 * N_WHERE], o[:, 2 * N_WHERE:]
    zw_sig = F.softplus(zw_sig - 1.0)
    zw = zw_mean + zw_sig * noise["eps_where"]
    f1 = F.relu(f1)
    z_where = zw.reshape(B, Hc, Wc, N_WHERE)
    # attr network :440-441
    glimpses, _ = spair_ref.stn_forward(x, z_where, S, S, inverse=False)             # [B,16,S,S,C]
    g = glimpses.reshape(B * Hc * Wc, S, S, -1)
    g = conv(p, "encoder/obj_encoder/conv2", conv(p, "encoder/obj_encoder/conv1", g, 2, "relu"), 2, "relu")
    hh = dense(p, "encoder/obj_encoder/dense1", g.reshape(g.shape[0], -1), "relu")
    zt_mean = dense(p, "encoder/obj_encoder/z_what_mu", hh)
    zt_sig = dense(p, "encoder/obj_encoder/z_what_sigma", hh, "softplus")
    zt = zt_mean + noise["eps_what"] * zt_sig
    prog = torch.cat([zw, zt], dim=1)
    # depth network :455-461
    o = dense(p, "encoder/dense_z_depth/1", dense(p, "encoder/dense_z_depth/0", torch.cat([fv, f1, prog], dim=1), "relu"))
    zd_mean, zd_sig, f2 = o[:, :N_DEPTH], o[:, N_DEPTH:2 * N_DEPTH], o[:, 2 * N_DEPTH:]
    zd_sig = F.softplus(zd_sig)
    zd = zd_mean + zd_sig * noise["eps_depth"]
    prog = torch.cat([prog, zd], dim=1)
    f2 = F.relu(f2)
    # presence network :464-467, concrete sample utils.py:14-17
    o = dense(p, "encoder/dense_z_pres/1", dense(p, "encoder/dense_z_pres/0", torch.cat([fv, f2, prog], dim=1), "relu"))
    logits = torch.clamp(o, -10.0, 10.0)
    u = noise["u_pres"]
    pre = (logits + torch.log(u + 1e-8) - torch.log(1.0 - u + 1e-8)) / cfg.tau
    pres = torch.sigmoid(pre)
    r = lambda t: t.reshape(B, Hc, Wc, -1)
    return dict(z_what=r(zt), z_what_mean=r(zt_mean), z_what_sigma=r(zt_sig), z_where=r(zw), z_where_mean=r(zw_mean),
                z_where_sigma=r(zw_sig), z_depth=r(zd), z_depth_mean=r(zd_mean), z_depth_sigma=r(zd_sig), z_pres=r(pres),
                z_pres_logits=r(logits), z_pres_pre_sigmoid=r(pre), all_glimpses=glimpses)


def obj_decoder(p, cfg, z_what):
    """ObjDecoder.call :355-366 on [B,Hc,Wc,Lw] -> rgb [N,S,S,C], alpha [N,S,S,1], N = B*Hc*Wc."""
    S, C = cfg.object_size, cfg.image_size[2]
    x = dense(p, "decoder/obj_decoder/d1", dense(p, "decoder/obj_decoder/d0", z_what, "relu"), "relu").reshape(-1, S // 4, S // 4, 32)
    x = torch_ref.resize_bilinear_2x(conv(p, "decoder/obj_decoder/d2", x, 1, "relu"))
    x = torch_ref.resize_bilinear_2x(conv(p, "decoder/obj_decoder/d3", x, 1, "relu"))
    x = conv(p, "decoder/obj_decoder/d5", x, 1)
    return torch.sigmoid(x[..., :C]), torch.sigmoid(x[..., C:])


def decoder(p, cfg, z_what, z_where):
    """Decoder.call :514-532."""
    B, Hc, Wc, _ = z_where.shape
    S, (H, W, C) = cfg.object_size, cfg.image_size
    rgb, alpha = obj_decoder(p, cfg, z_what)
    rgb = rgb.reshape(B, Hc * Wc, S, S, C)
    alpha = alpha.reshape(B, Hc * Wc, S, S, 1)
    full, bbox = spair_ref.stn_forward(torch.cat([rgb, alpha], dim=4), z_where, H, W, inverse=True)
    return rgb, alpha, full, bbox


def forward(p, cfg, images, noise, training=True):
    """SPAIR.call :35-49 / LGSPAIR.call :84-106 -> dict of the returned tuple's entries."""
    C = cfg.image_size[2]
    out = {}
    z_l = None
    if cfg.model == "lg_spair":
        x, x_hat = images[..., :3], images[..., 3:]
        z_l, out["z_l_mean"], out["z_l_sig"] = image_encoder(p, "x_hat_encoder", x_hat, noise["eps_l"], cfg.dense_local)
        z_bg, out["z_bg_mean"], out["z_bg_sig"] = image_encoder(p, "bg_encoder", x, noise["eps_bg"], cfg.dense_bg)
        out["z_l"] = z_l
    else:
        x = images
    e = encoder(p, cfg, x, z_l if (cfg.model == "lg_spair" and cfg.concat_backbone) else None, noise)
    out.update(e)
    z_what = e["z_what"]
    bg = 0.0
    if cfg.model == "lg_spair":
        out["x_hat_recon"] = image_decoder(p, "x_hat_decoder", z_l, cfg.image_size, cfg.dense_local)
        if cfg.concat_z_bg:
            z_bg = torch.cat([z_bg, z_l], dim=-1)
        out["z_bg"] = z_bg
        bg = image_decoder(p, "bg_decoder", z_bg, cfg.image_size, cfg.dense_bg)
        if cfg.concat_z_what:
            z_what = torch.cat([z_what, z_l[:, None, None, :].repeat(1, 4, 4, 1)], dim=-1)
            out["z_what"] = z_what                                                   # LGSPAIR returns the concatenated z_what :100-104
    elif cfg.model == "bg_spair":
        z_bg, out["z_bg_mean"], out["z_bg_sig"] = image_encoder(p, "bg_model", x, noise["eps_bg"], False, "z_bg_mu", "z_bg_sigma")
        out["z_bg"] = z_bg
        bg = image_decoder(p, "bg_model", z_bg, cfg.image_size, False)
    rgb, alpha, full, bbox = decoder(p, cfg, z_what, e["z_where"])
    out.update(obj_recon_unnorm=rgb, obj_recon_alpha=alpha, obj_full_recon_unnorm=full, obj_bbox_mask=bbox)
    if not torch.is_tensor(bg):
        bg = torch.zeros_like(x)
    out["x_recon"] = spair_ref.renderer(full, bg, e["z_depth"], e["z_pres"], e["z_pres_logits"], training=training,
                                        noise=noise.get("render") if training else None, num_channel=C)
    return out


# ------------------------------------------------------------------------------------------------ losses (spair/trainer.py)
def kl_divergence(z_mean, z_sig):
    """:13-21: mean over the batch of -0.5 sum(1 + log(sig^2 + 1e-8) - mean^2 - exp(that log))."""
    lv = spair_ref.tf_safe_log(z_sig * z_sig)
    t = 1 + lv - z_mean * z_mean - torch.exp(lv)
    return (-0.5 * t.reshape(t.shape[0], -1).sum(dim=1)).mean()


def kl_divergence_two_gauss(m1, s1, m2, s2):
    """:23-24."""
    t = spair_ref.tf_safe_log(s2) - spair_ref.tf_safe_log(s1) + (s1 * s1 + (m1 - m2) ** 2) / (2 * s2 * s2) - 0.5
    return t.reshape(t.shape[0], -1).sum(dim=1).mean()


def xent_loss(label, pred):
    """:103-104."""
    return -(label * spair_ref.tf_safe_log(pred) + (1.0 - label) * spair_ref.tf_safe_log(1.0 - pred))


def losses(cfg, images, o, step):
    """train_step :136-226 -> (total_loss, [the `losses` list the metrics see])."""
    lg = cfg.model == "lg_spair"
    x = images[..., :3] if lg else images
    x_recon_loss = spair_ref.tf_mean_sum(xent_loss(x, o["x_recon"]))
    anneal = min(1.0, (step + 1) / cfg.z_pres_anneal_step)
    z_pres_kl = spair_ref.compute_z_pres_kl_yolo_air(o["z_pres"], o["z_pres_logits"], o["z_pres_pre_sigmoid"], 0.99 * anneal, cfg.tau)
    zoom_mean = torch.full_like(o["z_where_mean"][..., :2], cfg.prior_z_zoom) + cfg.prior_z_zoom_start * (1 - anneal)
    zoom_sig = torch.full_like(o["z_where_sigma"][..., :2], 0.5)
    zoom_kl = kl_divergence_two_gauss(o["z_where_mean"][..., :2], o["z_where_sigma"][..., :2], zoom_mean, zoom_sig)
    what_kl = kl_divergence(o["z_what_mean"], o["z_what_sigma"])
    where_kl = kl_divergence(o["z_where_mean"][..., 2:], o["z_where_sigma"][..., 2:])
    depth_kl = kl_divergence(o["z_depth_mean"], o["z_depth_sigma"])
    lst = [x_recon_loss, zoom_kl, what_kl, where_kl, depth_kl, z_pres_kl]
    rw = cfg.reconstruction_weight
    obj = lambda wk: cfg.z_what_beta * wk + depth_kl + where_kl + zoom_kl + z_pres_kl
    annealed_beta = min(cfg.beta, cfg.beta * (step + 1.0) / cfg.anneal_until)
    if lg:
        x_hat = images[..., 3:]
        x_hat_loss = spair_ref.tf_mean_sum(xent_loss(x_hat, o["x_hat_recon"]))
        l_kl = kl_divergence(o["z_l_mean"], o["z_l_sig"])
        if not cfg.split_z_l:                                                        # :176-195
            if cfg.concat_z_bg:
                bg_kl = kl_divergence(torch.cat([o["z_bg_mean"], o["z_l_mean"]], dim=1), torch.cat([o["z_bg_sig"], o["z_l_sig"]], dim=1))
            else:
                bg_kl = kl_divergence(o["z_bg_mean"], o["z_bg_sig"])
            if cfg.concat_z_what:
                tile = lambda t: t[:, None, None, :].repeat(1, 4, 4, 1)
                what_kl = kl_divergence(torch.cat([o["z_what_mean"], tile(o["z_l_mean"])], dim=-1),
                                        torch.cat([o["z_what_sigma"], tile(o["z_l_sig"])], dim=-1))
            total = cfg.z_bg_beta * bg_kl + rw * x_recon_loss + cfg.beta * obj(what_kl) + x_hat_loss
        else:                                                                        # :197-207
            bg_kl = kl_divergence(o["z_bg_mean"], o["z_bg_sig"])
            total = cfg.z_bg_beta * bg_kl + cfg.z_l_beta * l_kl + x_hat_loss + rw * x_recon_loss + cfg.beta * obj(what_kl)
        lst += [bg_kl, l_kl, x_hat_loss]
    elif cfg.model == "bg_spair":                                                    # :222-228
        bg_kl = kl_divergence(o["z_bg_mean"], o["z_bg_sig"])
        lst.append(bg_kl)
        total = cfg.z_bg_beta * bg_kl + rw * x_recon_loss + annealed_beta * obj(what_kl)
    else:                                                                            # :165-167
        total = rw * x_recon_loss + annealed_beta * obj(what_kl)
    return total, lst


def clipnorm_adam_(params, grads, m, v, t, lr=1e-4, clipnorm=1.0, beta1=0.9, beta2=0.999, eps=1e-7, clip_in_apply=False):
    """tf.keras.optimizers.Adam(lr, clipnorm=1.0) (spair/main.py:109) as the reference's loop uses it: tape.gradient ->
    apply_gradients (spair/trainer.py:226-227).  [TF-2.0 semantics] the pinned tensorflow_gpu==2.0.0 clips only in
    get_gradients / _compute_gradients, not in apply_gradients (TF >= 2.4 does): clip_in_apply=False (default, pinned version) is
    the plain Keras Adam of torch_ref.keras_adam_; True = tf.clip_by_norm per gradient tensor first."""
    if not clip_in_apply:
        torch_ref.keras_adam_(params, list(grads), m, v, t, lr, beta1, beta2, eps)
        return
    clipped = []
    for g in grads:
        n = torch.sqrt((g * g).sum())
        clipped.append(g * (clipnorm / torch.maximum(n, torch.tensor(clipnorm, dtype=g.dtype))))
    torch_ref.keras_adam_(params, clipped, m, v, t, lr, beta1, beta2, eps)
