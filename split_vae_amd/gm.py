"""SPLIT-GMVAE: LGGMVae (vae/model.py:221-275) and its training step (vae/trainer.py:146-173) -- SURVEY 8a row A9.

LGGMVae = LGVae whose global encoder is Encoder(type='gmvae') (vae/model.py:48-79, call_gmvae :116-135).
encoder_x_hat, both decoders, the two reconstruction terms and the N(0,1) KL of the local latent are the
LGVae step plan, built with `external_global_encoder` (the plan then skips its own encoder_x).  The GMVAE
global encoder is orchestrated here, layer by layer, on the same HIP kernels: its convs and dense layers
are sv_conv2d_* calls (dense = 1x1 conv on a 1x1 grid), the glue between them is gm_pointwise.hip.
torch only owns the buffers.

Trainable variables, in the reference's layer-tracking order: the 24 arrays of the gmvae encoder
(h_block x3, y_block x2, y_dense, h_top_dense, z_prior_mean, z_prior_sig, e1, z_mean, z_sig), then the 30
arrays LGVae has after encoder_x.  The Dropout layers do1-4, do6, do7 exist in the reference but are never
called (vae/model.py:59-75 vs :116-135); only y_block's Dropout and do5 can act.

Whether they DO act in train_step_lg_gm_vae depends on the Keras version (LGGMVae.call drops `training` on the way to
encoder_x, vae/model.py:241; oracle/gm_ref.py::encoder_gmvae has the analysis): under the pinned tensorflow 2.0.0 they
do not, under tensorflow >= 2.1 they do.  `LGGMVae(dropout_in_training=...)` selects; the default False is the pinned
version's behaviour.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import (PHASE_ADAM, PHASE_BWD_DECODERS, PHASE_BWD_ENC_CONVS, PHASE_BWD_ENC_HEADS, PHASE_FWD_DECODERS,
                   PHASE_FWD_ENCODERS, PHASE_LOSS, PHASE_PREP)
from .model import LGVae

GM_RATE = 0.2      # Dropout(rate=0.2): y_block (vae/model.py:56) and do5 (:72)


def _r8(v):
    return (v + 7) // 8 * 8


class _Dense:
    """Dense(in -> out) as a 1x1 conv on a 1x1 grid; fp32 pre-activation out (bias included)."""

    def __init__(self, B, cin, cout, dtype):
        self.cin, self.cout = cin, cout
        self.conv = ops.Conv2D(B, 1, 1, cin, cout, 1, 1, act=None, dtype=dtype, y_f32=True)


class GMEncoder:
    """Encoder(latent_dims, type='gmvae', y_size, tau) for one batch size."""

    LAYERS = ["c1", "c2", "c3", "d1", "d2", "yd", "ht", "pm", "ps", "e1", "zm", "zs"]   # variable order (kernel, bias each)

    def __init__(self, B, H, W, latent, y_size, tau, dtype, device):
        self.B, self.H, self.W, self.L, self.K, self.tau, self.dtype, self.device = B, H, W, latent, y_size, tau, dtype, device
        self.F = (H // 8) * (W // 8) * 128
        F_, K, L = self.F, y_size, latent
        self.shapes = [(6, 6, 3, 128), (128,), (6, 6, 128, 128), (128,), (4, 4, 128, 128), (128,),
                       (F_, 1024), (1024,), (1024, 128), (128,), (128, K), (K,), (K, 512), (512,),
                       (K, L), (L,), (K, L), (L,), (F_, 512), (512,), (512, L), (L,), (512, L), (L,)]
        self.names = ["encoder_x/h_block/conv2d", "encoder_x/h_block/conv2d_1", "encoder_x/h_block/conv2d_2",
                      "encoder_x/y_block/dense", "encoder_x/y_block/dense_1", "encoder_x/y_dense", "encoder_x/h_top_dense",
                      "encoder_x/z_prior_mean", "encoder_x/z_prior_sig", "encoder_x/e1", "encoder_x/z_mean", "encoder_x/z_sig"]
        self.table, off = [], 0
        for i, shp in enumerate(self.shapes):
            self.table.append((self.names[i // 2] + ("/kernel" if i % 2 == 0 else "/bias"), off, shp))
            off += (int(np.prod(shp)) + 3) // 4 * 4
        self.n_params = off
        cv = lambda h, cin, cout, k: ops.Conv2D(B, h, h, cin, cout, k, 2, act=None, dtype=dtype)
        self.conv = {"c1": cv(H, 3, 128, 6), "c2": cv(H // 2, 128, 128, 6), "c3": cv(H // 4, 128, 128, 4)}
        dn = lambda cin, cout: _Dense(B, cin, cout, dtype).conv
        self.conv.update({"d1": dn(F_, 1024), "d2": dn(1024, 128), "yd": dn(128, K), "ht": dn(K, 512), "pm": dn(K, L),
                          "ps": dn(K, L), "e1": dn(F_, 512), "zm": dn(512, L), "zs": dn(512, L)})
        T, f32 = dtype, torch.float32
        z = lambda *s, dt=T: torch.zeros(s, dtype=dt, device=device)
        Kp = _r8(K)
        self.buf = dict(
            h1=z(B, H // 2, H // 2, 128), h2=z(B, H // 4, H // 4, 128), h3=z(B, self.F),
            a1=z(B, 1024, dt=f32), yh1a=z(B, 1024), yh1=z(B, 1024), keep1=z(B, 1024, dt=f32),
            a2=z(B, 128, dt=f32), yh2=z(B, 128), logits=z(B, K, dt=f32), y=z(B, K, dt=f32), y_lp=z(B, Kp), u=z(B, K, dt=f32),
            a_pm=z(B, L, dt=f32), a_ps=z(B, L, dt=f32), a_t=z(B, 512, dt=f32), h_top=z(B, 512),
            h5=z(B, self.F), keep5=z(B, self.F, dt=f32), a_e=z(B, 512, dt=f32), he=z(B, 512), hh=z(B, 512),
            a_m=z(B, L, dt=f32), a_s=z(B, L, dt=f32), zm=z(B, L, dt=f32), zs=z(B, L, dt=f32), z=z(B, L, dt=f32),
            pm=z(B, L, dt=f32), ps=z(B, L, dt=f32), eps=z(B, L, dt=f32), kl2=z(B, dt=f32), ykl=z(B, dt=f32),
            # backward
            g_am=z(B, L), g_as=z(B, L), g_apm=z(B, L), g_aps=z(B, L), g_hh=z(B, 512, dt=f32), g_ae=z(B, 512), g_at=z(B, 512),
            g_h5=z(B, self.F, dt=f32), g_y=z(B, Kp, dt=f32), g_logits=z(B, Kp), g_yh2=z(B, 128, dt=f32), g_a2=z(B, 128),
            g_yh1=z(B, 1024, dt=f32), g_a1=z(B, 1024), g_h1=z(B, self.F, dt=f32), g_c3=z(B, self.F),
            g_c2=z(B, H // 4, H // 4, 128), g_c1=z(B, H // 2, H // 2, 128))

    def views(self, flat):
        # cached per buffer: this runs ~25 times per step (it was over half of the step's host time)
        key = (flat.data_ptr(), flat.numel())
        cache = self.__dict__.setdefault("_view_cache", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = [flat[off:off + int(np.prod(shp))].view(*shp) for (_, off, shp) in self.table]
        return cache[key]

    def _kb(self, flat, name):
        i = self.LAYERS.index(name)
        v = self.views(flat)
        k, b = v[2 * i], v[2 * i + 1]
        if k.dim() == 2:
            k = k.view(1, 1, *k.shape)                  # dense kernel [in,out] == HWIO with H=W=1
        return k, b

    def prep(self, flat):
        for n in self.LAYERS:
            self.conv[n].prep(self._kb(flat, n)[0])

    # ------------------------------------------------------------------ call_gmvae (vae/model.py:116-135)
    def forward(self, flat, in8_x, zcat, training, eps=None, u=None, keep1=None, keep5=None, seed=0, step=0, sample_offset=0):
        b, c, B, K = self.buf, self.conv, self.B, self.K
        rate = GM_RATE if training else 0.0
        bias = lambda n: self._kb(flat, n)[1]
        rows = lambda t: t.view(-1, 128)
        # h_block: three stride-2 convs with ELU (:50-52)
        c["c1"].fwd(in8_x, bias("c1"), out=b["h1"]); ops.act_fwd(rows(b["h1"]), 128, rows(b["h1"]), "elu")
        c["c2"].fwd(b["h1"], bias("c2"), out=b["h2"]); ops.act_fwd(rows(b["h2"]), 128, rows(b["h2"]), "elu")
        h3 = b["h3"].view(B, self.H // 8, self.W // 8, 128)
        c["c3"].fwd(b["h2"], bias("c3"), out=h3); ops.act_fwd(rows(h3), 128, rows(h3), "elu")
        # y_block (:54-58) -> y_dense (:60) -> Gumbel-softmax (:121-122)
        c["d1"].fwd(b["h3"], bias("d1"), out=b["a1"])
        ops.act_fwd(b["a1"], 1024, b["yh1"], "elu", y_act=b["yh1a"], rate=rate, keep_in=keep1, keep_out=b["keep1"], seed=seed,
                    step=step, stream_id=11, sample_offset=sample_offset)
        c["d2"].fwd(b["yh1"], bias("d2"), out=b["a2"]); ops.act_fwd(b["a2"], 128, b["yh2"], "elu")
        c["yd"].fwd(b["yh2"], bias("yd"), out=b["logits"])
        ops.gumbel_softmax_fwd(b["logits"], K, self.tau, b["y"], b["y_lp"], u=u, u_out=b["u"], seed=seed, step=step,
                               sample_offset=sample_offset)
        # prior (:124-125), h_top (:127), encoder block (:128-133)
        c["pm"].fwd(b["y_lp"], bias("pm"), out=b["a_pm"])
        c["ps"].fwd(b["y_lp"], bias("ps"), out=b["a_ps"])
        c["ht"].fwd(b["y_lp"], bias("ht"), out=b["a_t"]); ops.act_fwd(b["a_t"], 512, b["h_top"], "elu")
        ops.act_fwd(b["h3"], self.F, b["h5"], None, rate=rate, keep_in=keep5, keep_out=b["keep5"], seed=seed, step=step,
                    stream_id=15, sample_offset=sample_offset)
        c["e1"].fwd(b["h5"], bias("e1"), out=b["a_e"]); ops.act_fwd(b["a_e"], 512, b["he"], "elu")
        ops.add(b["he"], b["h_top"], b["hh"])
        c["zm"].fwd(b["hh"], bias("zm"), out=b["a_m"])
        c["zs"].fwd(b["hh"], bias("zs"), out=b["a_s"])
        ops.gm_head_fwd(b["a_m"], b["a_s"], b["a_pm"], b["a_ps"], b["zm"], b["zs"], b["z"], b["pm"], b["ps"], zcat, 0, b["kl2"],
                        eps=eps, eps_out=b["eps"], seed=seed, step=step, sample_offset=sample_offset)
        self._rate = rate

    # ------------------------------------------------------------------ adjoint (tape.gradient, vae/trainer.py:167)
    def backward(self, flat, grad_flat, in8_x, gz, beta, alpha):
        """gz [B, >=L] fp32: dL/dz_x from the decoder (columns [0, L)).  Accumulates the 24 gradients into grad_flat
        (zeroed by the caller) and fills ykl."""
        b, c, B, K, L, F_ = self.buf, self.conv, self.B, self.K, self.L, self.F
        rate = self._rate
        gv = self.views(grad_flat)
        for t in ("g_hh", "g_h5", "g_y", "g_yh2", "g_yh1", "g_h1"):
            b[t].zero_()                                # fp32 accumulation targets of the split-K dgrads

        def wg(name, x, dy):
            i = self.LAYERS.index(name)
            dw = gv[2 * i] if gv[2 * i].dim() == 4 else gv[2 * i].view(1, 1, *gv[2 * i].shape)
            c[name].wgrad(x, dy, dw=dw, db=gv[2 * i + 1])

        ops.gm_head_bwd(gz, b["zm"], b["zs"], b["pm"], b["ps"], b["eps"], beta / B, b["g_am"], b["g_as"], b["g_apm"], b["g_aps"])
        wg("zm", b["hh"], b["g_am"]); wg("zs", b["hh"], b["g_as"])
        c["zm"].dgrad(b["g_am"], f32_atomic=True, out=b["g_hh"].view(B, 1, 1, 512))
        c["zs"].dgrad(b["g_as"], f32_atomic=True, out=b["g_hh"].view(B, 1, 1, 512))
        ops.act_bwd(b["g_hh"], 512, b["g_ae"], y_act=b["he"], act="elu")
        ops.act_bwd(b["g_hh"], 512, b["g_at"], y_act=b["h_top"], act="elu")
        wg("e1", b["h5"], b["g_ae"])
        c["e1"].dgrad(b["g_ae"], f32_atomic=True, out=b["g_h5"].view(B, 1, 1, F_))
        Kp = b["y_lp"].shape[1]
        for n, g in (("ht", "g_at"), ("pm", "g_apm"), ("ps", "g_aps")):
            wg(n, b["y_lp"], b[g])
            c[n].dgrad(b[g], f32_atomic=True, out=b["g_y"].view(B, 1, 1, Kp))
        ops.gumbel_softmax_bwd(b["g_y"], b["y"], b["logits"], K, self.tau, alpha / B, b["g_logits"], b["ykl"])
        wg("yd", b["yh2"], b["g_logits"])
        c["yd"].dgrad(b["g_logits"], f32_atomic=True, out=b["g_yh2"].view(B, 1, 1, 128))
        ops.act_bwd(b["g_yh2"], 128, b["g_a2"], y_act=b["yh2"], act="elu")
        wg("d2", b["yh1"], b["g_a2"])
        c["d2"].dgrad(b["g_a2"], f32_atomic=True, out=b["g_yh1"].view(B, 1, 1, 1024))
        ops.act_bwd(b["g_yh1"], 1024, b["g_a1"], y_act=b["yh1a"], act="elu", rate=rate, keep=b["keep1"])
        wg("d1", b["h3"], b["g_a1"])
        c["d1"].dgrad(b["g_a1"], f32_atomic=True, out=b["g_h1"].view(B, 1, 1, F_))
        # h feeds y_block (g_h1) and, through do5, e1 (g_h5): combine, then ELU' of conv3
        ops.act_bwd(b["g_h5"], F_, b["g_c3"], y_act=b["h3"], act="elu", rate=rate, keep=b["keep5"], gx2=b["g_h1"])
        rows = lambda t: t.view(-1, 128)
        g_c3 = b["g_c3"].view(B, self.H // 8, self.W // 8, 128)
        wg("c3", b["h2"], g_c3)
        g_h2 = c["c3"].dgrad(g_c3)
        ops.act_bwd(rows(g_h2), 128, rows(b["g_c2"]), y_act=rows(b["h2"]), act="elu")
        wg("c2", b["h1"], b["g_c2"])
        g_h1 = c["c2"].dgrad(b["g_c2"])
        ops.act_bwd(rows(g_h1), 128, rows(b["g_c1"]), y_act=rows(b["h1"]), act="elu")
        wg("c1", in8_x, b["g_c1"])

    def y_kl_only(self):
        b = self.buf
        ops.gumbel_softmax_bwd(None, b["y"], b["logits"], self.K, self.tau, 0.0, None, b["ykl"])


class NativeGMEncoder:
    """The same encoder sequenced natively (csrc/gm_encoder.hip, include/splitvae.h `sv_gm_encoder_*`): one C call each
    for weight preparation, forward and backward instead of ~110 ctypes calls per step.  `buf[name]` are views of
    its workspace (same names as GMEncoder.buf)."""

    def __init__(self, B, H, W, latent, y_size, tau, dtype, device):
        self.B, self.H, self.W, self.L, self.K, self.tau, self.dtype, self.device = B, H, W, latent, y_size, tau, dtype, device
        self.F = (H // 8) * (W // 8) * 128
        lib = _lib.load()
        self.lib = lib
        self.desc = _lib.GmDesc(B, H, W, latent, y_size, float(tau), ops.sv_dtype(dtype))
        h = C.c_void_p()
        ops.check(lib.sv_gm_encoder_create(C.byref(self.desc), C.byref(h)), "sv_gm_encoder_create")
        self.handle = h
        nbytes = lib.sv_gm_encoder_workspace_bytes(h)
        self.workspace = torch.zeros((nbytes,), dtype=torch.uint8, device=device)
        ops.check(lib.sv_gm_encoder_bind(h, ops._p(self.workspace), nbytes, ops._stream()), "sv_gm_encoder_bind")
        T, f32, F_, K, L, Kp = dtype, torch.float32, self.F, y_size, latent, _r8(y_size)
        shapes = dict(
            h1=(T, (B, H // 2, H // 2, 128)), h2=(T, (B, H // 4, H // 4, 128)), h3=(T, (B, F_)), a1=(f32, (B, 1024)),
            yh1a=(T, (B, 1024)), yh1=(T, (B, 1024)), keep1=(f32, (B, 1024)), a2=(f32, (B, 128)), yh2=(T, (B, 128)),
            logits=(f32, (B, K)), y=(f32, (B, K)), y_lp=(T, (B, Kp)), u=(f32, (B, K)), a_pm=(f32, (B, L)), a_ps=(f32, (B, L)),
            a_t=(f32, (B, 512)), h_top=(T, (B, 512)), h5=(T, (B, F_)), keep5=(f32, (B, F_)), a_e=(f32, (B, 512)),
            he=(T, (B, 512)), hh=(T, (B, 512)), a_m=(f32, (B, L)), a_s=(f32, (B, L)), zm=(f32, (B, L)), zs=(f32, (B, L)),
            z=(f32, (B, L)), pm=(f32, (B, L)), ps=(f32, (B, L)), eps=(f32, (B, L)), kl2=(f32, (B,)), ykl=(f32, (B,)))
        self.buf = {}
        off, nb = C.c_int64(), C.c_int64()
        for name, (dt, shp) in shapes.items():
            ops.check(lib.sv_gm_encoder_buffer(h, name.encode(), C.byref(off), C.byref(nb)), "sv_gm_encoder_buffer " + name)
            self.buf[name] = self.workspace[off.value: off.value + nb.value].view(dt).view(*shp)
        self._args = _lib.GmArgs()

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.sv_gm_encoder_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def prep(self, flat):
        ops.check(self.lib.sv_gm_encoder_prep(self.handle, ops._p(flat), ops._stream()), "sv_gm_encoder_prep")

    def _fill(self, **kw):
        a = self._args
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        return a

    def forward(self, flat, in8_x, zcat, training, eps=None, u=None, keep1=None, keep5=None, seed=0, step=0, sample_offset=0):
        a = self._fill(params=flat, grads=None, in8_x=in8_x, zcat=zcat, ldz=zcat.shape[1], gz=None, ld_gz=0, eps=eps, u=u,
                       keep1=keep1, keep5=keep5, training=1 if training else 0, beta=0.0, alpha=0.0, seed=seed, step=step,
                       sample_offset=sample_offset)
        ops.check(self.lib.sv_gm_encoder_forward(self.handle, C.byref(a), ops._stream()), "sv_gm_encoder_forward")

    def backward(self, flat, grad_flat, in8_x, gz, beta, alpha):
        a = self._fill(params=flat, grads=grad_flat, in8_x=in8_x, gz=gz, ld_gz=gz.shape[1], beta=float(beta), alpha=float(alpha))
        ops.check(self.lib.sv_gm_encoder_backward(self.handle, C.byref(a), ops._stream()), "sv_gm_encoder_backward")

    def y_kl_only(self):
        ops.check(self.lib.sv_gm_encoder_y_kl(self.handle, ops._stream()), "sv_gm_encoder_y_kl")


class LGGMVae(LGVae):
    """vae/model.py:221-246: LGGMVae(global_latent_dims, local_latent_dims, image_shape, y_size, tau)."""

    def __init__(self, global_latent_dims, local_latent_dims, image_shape, y_size, tau, variational=True, type='conv',
                 dtype='bf16', device=None, seed=0, dropout_in_training=False):
        super().__init__(global_latent_dims, local_latent_dims, image_shape, variational, type, dtype, device, seed)
        self.y_size, self.tau = y_size, tau
        # False: tensorflow 2.0.0 (pinned) -- `training` never reaches encoder_x's Dropout layers; True: tensorflow >= 2.1
        self.dropout_in_training = bool(dropout_in_training)
        self.alpha = 40.0                                  # vae/main.py:29 (--alpha)
        self._enc, self._enc_py = {}, {}
        self.gm_table = GMEncoder(1, self.H, self.W, global_latent_dims, y_size, tau, self.dtype, self.device).table
        self.gm_n_params = self.gm_table[-1][1] + (int(np.prod(self.gm_table[-1][2])) + 3) // 4 * 4
        self.gm_flat = torch.zeros(self.gm_n_params, dtype=torch.float32, device=self.device)
        self.gm_grad_flat = torch.zeros_like(self.gm_flat)
        self._init_gm(seed + 1)
        # the plan's own encoder_x slots are unused in this model: keep them at zero
        for name, off, shape in self.param_table:
            if name.startswith("encoder_x/"):
                self.flat[off:off + int(np.prod(shape))].zero_()

    def _init_gm(self, seed):
        """Keras defaults; z_prior_sig / z_sig biases are constant(1) (vae/model.py:68,:78)."""
        rng = np.random.Generator(np.random.PCG64(seed))
        host = np.zeros(self.gm_n_params, np.float32)
        for name, off, shp in self.gm_table:
            n = int(np.prod(shp))
            if name.endswith("kernel"):
                fan_in = int(np.prod(shp[:-1]))
                fan_out = int(np.prod(shp[:-2])) * shp[-1] if len(shp) == 4 else shp[-1]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                host[off:off + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
            elif name in ("encoder_x/z_prior_sig/bias", "encoder_x/z_sig/bias"):
                host[off:off + n] = 1.0
        self.gm_flat.copy_(torch.from_numpy(host))

    # ---------------------------------------------------------------- variables (54 arrays)
    def _gm_views(self, flat):
        return [flat[off:off + int(np.prod(shp))].view(*shp) for (_, off, shp) in self.gm_table]

    @property
    def trainable_variables(self):
        return self._gm_views(self.gm_flat) + self._views(self.flat)[10:]

    @property
    def gradients(self):
        return self._gm_views(self.gm_grad_flat) + self._views(self.grad_flat)[10:]

    def keras_names(self):
        # same convention as LGVae.keras_names: <sublayer>/<attr>/<kernel|bias>:0
        return [n + ":0" for n, _, _ in self.gm_table] + [n + ":0" for n, _, _ in self.param_table][10:]

    KERAS_MODEL_NAME = "lggm_vae"

    def _keras_kind(self, name):
        parts = name.split("/")
        if parts[0] != "encoder_x":
            return super()._keras_kind(name)
        attr = parts[1]
        if attr == "h_block":                              # Sequential of three Conv2D (vae/model.py:49-52)
            return "conv2d", None, "h_block"
        if attr == "y_block":                              # Sequential(Dense, Dropout, Dense) (:54-58)
            return "dense", None, "y_block"
        if attr in ("y_dense", "z_prior_mean", "z_prior_sig"):      # layers given explicit names (:60,:66,:68)
            return "dense", attr, None
        return "dense", None, None                         # h_top_dense, e1, z_mean, z_sig: auto-named Dense

    def summary(self):
        total = 0
        for n, v in zip(self.keras_names(), self.trainable_variables):
            print("%-40s %s" % (n, tuple(v.shape)))
            total += v.numel()
        print("Total params: {:,}".format(total))

    def set_weights(self, arrays):
        assert len(arrays) == 54
        for v, a in zip(self.trainable_variables, arrays):
            v.copy_(torch.as_tensor(np.asarray(a), dtype=torch.float32).to(self.device).reshape(v.shape))

    def get_weights(self):
        return [v.detach().cpu().numpy().copy() for v in self.trainable_variables]

    # ---------------------------------------------------------------- plans
    def plan(self, B, beta=None):
        beta = self.beta if beta is None else beta
        key = (int(B), float(beta))
        if key not in self._plans:
            self._plans[key] = ops.LGVaePlan(B, self.H, self.W, self.global_latent_dims, self.local_latent_dims, beta=beta,
                                             dtype=self.dtype, device=self.device, external_global_encoder=True)
        return self._plans[key]

    def encoder(self, B):
        """The natively sequenced encoder (SV_GM_PYTHON=1: the per-layer Python sequence, kept as the readable restatement
        of the launch order and for A/B)."""
        if B not in self._enc:
            cls = GMEncoder if os.environ.get("SV_GM_PYTHON") == "1" else NativeGMEncoder
            self._enc[B] = cls(B, self.H, self.W, self.global_latent_dims, self.y_size, self.tau, self.dtype, self.device)
        return self._enc[B]

    def _py_encoder(self, n):
        if n not in self._enc_py:
            self._enc_py[n] = GMEncoder(n, self.H, self.W, self.global_latent_dims, self.y_size, self.tau, self.dtype, self.device)
        return self._enc_py[n]

    def _forward(self, inputs, training, eps, noise, want_loss, plan_kw):
        """noise = (u, keep1, keep5) pins the Gumbel uniforms / dropout masks (parity tests)."""
        B = inputs.shape[0]
        plan, enc = self.plan(B), self.encoder(B)
        ex, eh = (None, None) if eps is None else eps
        u, k1, k5 = (None, None, None) if noise is None else noise
        kw = dict(params=self.flat, images6=inputs.contiguous(), eps_x_hat=eh, seed=self.seed, step=self._calls)
        kw.update(plan_kw)
        plan.step(PHASE_PREP | PHASE_FWD_ENCODERS, **kw)
        enc.prep(self.gm_flat)
        Lc = self.global_latent_dims + self.local_latent_dims
        enc.forward(self.gm_flat, plan.buffer("in8_x", self.dtype, (B, self.H, self.W, 8)),
                    plan.buffer("zcat", self.dtype, (B, Lc)), bool(training) and self.dropout_in_training, eps=ex, u=u,
                    keep1=k1, keep5=k5, seed=self.seed,
                    step=self._calls, sample_offset=kw.get("sample_offset", 0))
        plan.step(PHASE_FWD_DECODERS | (PHASE_LOSS if want_loss else 0), **kw)
        return plan, enc, kw

    def __call__(self, inputs, training=False, eps=None, noise=None, copy=True):
        """vae/model.py:236-246 -> the 14-tuple (the LGVae 10-tuple, then y, y_logits, z_prior_mean, z_prior_sig)."""
        B = inputs.shape[0]
        plan, enc, _ = self._forward(inputs, training, eps, noise, False, {})
        self._calls += 1
        o = list(self._outputs(plan, B, copy))
        b = enc.buf
        o[2], o[3], o[4] = b["z"].clone(), b["zm"].clone(), b["zs"].clone()
        return tuple(o) + (b["y"].clone(), b["logits"].clone(), b["pm"].clone(), b["ps"].clone())

    call = __call__

    def encode(self, inputs, eps=None):
        out = self(inputs, eps=eps)
        return out[2], out[5]

    def encode_y(self, y, rescale=True):
        """vae/model.py:263-265 / :137-140: prior mean and sig of a given y [n, y_size] (fp32)."""
        n = y.shape[0]
        enc = self._py_encoder(n)                          # two Dense layers and a head: the per-layer objects
        enc.prep(self.gm_flat)
        b = enc.buf
        b["y_lp"].zero_(); b["y_lp"][:, :self.y_size] = y.to(self.dtype)
        enc.conv["pm"].fwd(b["y_lp"], enc._kb(self.gm_flat, "pm")[1], out=b["a_pm"])
        enc.conv["ps"].fwd(b["y_lp"], enc._kb(self.gm_flat, "ps")[1], out=b["a_ps"])
        dummy = torch.zeros((n, self.global_latent_dims), dtype=torch.float32, device=self.device)
        zc = torch.zeros((n, self.global_latent_dims), dtype=self.dtype, device=self.device)
        ops.gm_head_fwd(b["a_pm"], b["a_ps"], b["a_pm"], b["a_ps"], b["zm"], b["zs"], b["z"], b["pm"], b["ps"], zc, 0, b["kl2"],
                        eps=dummy)
        return b["pm"].clone(), b["ps"].clone()

    def get_y(self, x):
        """vae/model.py:267-270: (y, y_logits) of encoder_x for x [n,H,W,3] (a 6-channel batch uses its x half)."""
        x = x[..., :3]
        out = self(torch.cat([x, x], dim=-1).contiguous())
        return out[10], out[11]


LOSS_KEYS = ["x_recon_loss", "x_kl_loss", "x_hat_recon_loss", "x_hat_kl_loss", "y_kl_loss", "total_loss"]


def _metrics(model, plan, enc, B):
    """The five Mean metrics of vae/trainer.py:169-173 (+ total) from the per-image terms, as a [6] fp32 device tensor."""
    nll_x = plan.buffer("nll_x", torch.float32, (B,))
    nll_h = plan.buffer("nll_xh", torch.float32, (B,))
    kl_h = plan.buffer("kl_xh", torch.float32, (B,))
    per = torch.stack([nll_x, enc.buf["kl2"], nll_h, kl_h, enc.buf["ykl"]])          # [5, B] views -> batch means
    m = per.mean(dim=1)
    total = m[0] + m[2] + model.beta * (m[1] + m[3]) + model.alpha * m[4]
    return torch.cat([m, total[None]])


def train_step_lg_gm_vae(model, images, optimizer, eps=None, noise=None, sample_offset=0):
    """train_step_lg_gm_vae (vae/trainer.py:146-173): total = recon_x + recon_x_hat + beta*(KL(q_x || p_y) + KL(q_xh || N(0,1)))
    + alpha * KL(softmax(y_logits) || uniform); gradients of the 54 variables; Adam; returns the [6] metric tensor."""
    if not isinstance(model, LGGMVae):
        raise NotImplementedError("train_step_lg_gm_vae needs an LGGMVae")
    with ops.hold_stream():
        return _train_step_lg_gm_vae(model, images, optimizer, eps, noise, sample_offset)


def _train_step_lg_gm_vae(model, images, optimizer, eps, noise, sample_offset):
    B = images.shape[0]
    m, v = optimizer.slots(model.flat)
    gm_m, gm_v = optimizer.slots(model.gm_flat)
    lr = optimizer.lr()
    optimizer.iterations += 1
    t = optimizer.iterations
    kw = dict(grads=model.grad_flat, adam_m=m, adam_v=v, sample_offset=sample_offset, lr=lr, beta1=optimizer.beta_1,
              beta2=optimizer.beta_2, adam_eps=optimizer.epsilon, t=t, accumulate_metrics=False)
    plan, enc, kw = model._forward(images, True, eps, noise, True, kw)
    model._calls += 1
    plan.step(PHASE_BWD_DECODERS, **kw)
    Lc = model.global_latent_dims + model.local_latent_dims
    model.gm_grad_flat.zero_()
    enc.backward(model.gm_flat, model.gm_grad_flat, plan.buffer("in8_x", model.dtype, (B, model.H, model.W, 8)),
                 plan.buffer("gz_x", torch.float32, (B, Lc)), model.beta, model.alpha)
    plan.step(PHASE_BWD_ENC_HEADS | PHASE_BWD_ENC_CONVS, **kw)
    metrics = _metrics(model, plan, enc, B)
    plan.step(PHASE_ADAM, **kw)
    ops.adam_step(model.gm_flat, model.gm_grad_flat, gm_m, gm_v, t, lr, optimizer.beta_1, optimizer.beta_2, optimizer.epsilon)
    return metrics


def test_step_lg_gm_vae(model, images, eps=None, noise=None):
    """Evaluation counterpart: same loss terms with training=False (no dropout), no update."""
    B = images.shape[0]
    with ops.hold_stream():
        plan, enc, _ = model._forward(images, False, eps, noise, True, {})
        model._calls += 1
        enc.y_kl_only()
        return _metrics(model, plan, enc, B)
