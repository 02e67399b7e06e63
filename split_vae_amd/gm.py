"""SPLIT-GMVAE: LGGMVae (vae/model.py:221-275) and its training step (vae/trainer.py:146-173) -- SURVEY 8a row A9.

LGGMVae = LGVae whose global encoder is Encoder(type='gmvae') (vae/model.py:48-79, call_gmvae :116-135).
encoder_x_hat, both decoders, the two reconstruction terms and the N(0,1) KL of the local latent are the
LGVae step plan, built with `external_global_encoder` (the plan then skips its own encoder_x).  The GMVAE
global encoder runs on the same HIP kernels -- its convs and dense layers are sv_conv2d_* calls (dense = 1x1 conv on a 1x1 grid),
the glue between them is gm_pointwise.hip -- sequenced natively by csrc/gm_encoder.hip (sv_gm_encoder_*: one C call per phase).
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
                   PHASE_FWD_ENCODERS, PHASE_INPUTS_STAGED, PHASE_LOSS, PHASE_PREP)

# the global (GM) encoder and the local encoder on two HIP streams, forward (when the augmentation staged the step's inputs) and backward: SV_GM_STREAMS=0 keeps
# everything on the caller's stream (A/B, the equivalence test).  Measured: profiles/r06_gm_streams.txt
_GM_STREAMS = int(os.environ.get("SV_GM_STREAMS", "3"))        # bit 0: the forward, bit 1: the backward
_SIDE_STREAMS = {}                                              # device index -> the process's one side stream (LGGMVae._side_stream)
from .model import LGVae

GM_RATE = 0.2      # Dropout(rate=0.2): y_block (vae/model.py:56) and do5 (:72)


def _r8(v):
    return (v + 7) // 8 * 8


class _Dense:
    """Dense(in -> out) as a 1x1 conv on a 1x1 grid; fp32 pre-activation out (bias included)."""

    def __init__(self, B, cin, cout, dtype):
        self.cin, self.cout = cin, cout
        self.conv = ops.Conv2D(B, 1, 1, cin, cout, 1, 1, act=None, dtype=dtype, y_f32=True)


GM_LAYER_NAMES = ["encoder_x/h_block/conv2d", "encoder_x/h_block/conv2d_1", "encoder_x/h_block/conv2d_2",
                  "encoder_x/y_block/dense", "encoder_x/y_block/dense_1", "encoder_x/y_dense", "encoder_x/h_top_dense",
                  "encoder_x/z_prior_mean", "encoder_x/z_prior_sig", "encoder_x/e1", "encoder_x/z_mean", "encoder_x/z_sig"]
GM_LAYERS = ["c1", "c2", "c3", "d1", "d2", "yd", "ht", "pm", "ps", "e1", "zm", "zs"]   # variable order (kernel, bias each)


def gm_param_table(H, W, latent, y_size):
    """(name, offset, shape) of the 24 variables of Encoder(type='gmvae') (vae/model.py:48-79) in the reference's layer-tracking
    order, each 16-B aligned in the flat buffer (the layout csrc/gm_encoder.hip's build_params uses)."""
    F_, K, L = (H // 8) * (W // 8) * 128, y_size, latent
    shapes = [(6, 6, 3, 128), (128,), (6, 6, 128, 128), (128,), (4, 4, 128, 128), (128,),
              (F_, 1024), (1024,), (1024, 128), (128,), (128, K), (K,), (K, 512), (512,),
              (K, L), (L,), (K, L), (L,), (F_, 512), (512,), (512, L), (L,), (512, L), (L,)]
    table, off = [], 0
    for i, shp in enumerate(shapes):
        table.append((GM_LAYER_NAMES[i // 2] + ("/kernel" if i % 2 == 0 else "/bias"), off, shp))
        off += (int(np.prod(shp)) + 3) // 4 * 4
    return table


class PriorHead:
    """The two Dense layers encode_y needs (z_prior_mean / z_prior_sig of a GIVEN y, vae/model.py:137-140) for n rows: inference only.
    (The training sequence of the whole encoder is native: NativeGMEncoder / csrc/gm_encoder.hip.)"""

    def __init__(self, n, H, W, latent, y_size, dtype, device):
        self.table = gm_param_table(H, W, latent, y_size)
        self.conv = {k: _Dense(n, y_size, latent, dtype).conv for k in ("pm", "ps")}
        f32, Kp = torch.float32, _r8(y_size)
        z = lambda *s, dt=dtype: torch.zeros(s, dtype=dt, device=device)
        self.buf = dict(y_lp=z(n, Kp), a_pm=z(n, latent, dt=f32), a_ps=z(n, latent, dt=f32), zm=z(n, latent, dt=f32), zs=z(n, latent, dt=f32),
                        z=z(n, latent, dt=f32), pm=z(n, latent, dt=f32), ps=z(n, latent, dt=f32), kl2=z(n, dt=f32))

    def _kb(self, flat, name):
        i = GM_LAYERS.index(name)
        (_, ko, ks), (_, bo, bs) = self.table[2 * i], self.table[2 * i + 1]
        k = flat[ko:ko + int(np.prod(ks))].view(1, 1, *ks)       # dense kernel [in,out] == HWIO with H=W=1
        return k, flat[bo:bo + int(np.prod(bs))]

    def prep(self, flat):
        for n in ("pm", "ps"):
            self.conv[n].prep(self._kb(flat, n)[0], dgrad=False)


class NativeGMEncoder:
    """The same encoder sequenced natively (csrc/gm_encoder.hip, include/splitvae.h `sv_gm_encoder_*`): one C call each
    for weight preparation, forward and backward instead of ~110 ctypes calls per step.  `buf[name]` are views of
    its workspace."""

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
        self.gm_table = gm_param_table(self.H, self.W, global_latent_dims, y_size)
        self.gm_n_params = self.gm_table[-1][1] + (int(np.prod(self.gm_table[-1][2])) + 3) // 4 * 4
        self.gm_flat = torch.zeros(self.gm_n_params, dtype=torch.float32, device=self.device)
        self.gm_grad_flat = torch.zeros_like(self.gm_flat)
        self._metrics_buf = torch.zeros(8, dtype=torch.float32, device=self.device)      # sv_gm_metrics writes [0:6]
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
        """The natively sequenced encoder (csrc/gm_encoder.hip) for batch B."""
        if B not in self._enc:
            self._enc[B] = NativeGMEncoder(B, self.H, self.W, self.global_latent_dims, self.y_size, self.tau, self.dtype, self.device)
        return self._enc[B]

    def _py_encoder(self, n):
        if n not in self._enc_py:
            self._enc_py[n] = PriorHead(n, self.H, self.W, self.global_latent_dims, self.y_size, self.dtype, self.device)
        return self._enc_py[n]

    def _forward(self, inputs, training, eps, noise, want_loss, plan_kw):
        """noise = (u, keep1, keep5) pins the Gumbel uniforms / dropout masks (parity tests)."""
        B = inputs.shape[0]
        plan, enc = self.plan(B), self.encoder(B)
        ex, eh = (None, None) if eps is None else eps
        u, k1, k5 = (None, None, None) if noise is None else noise
        kw = dict(params=self.flat, images6=inputs.contiguous(), eps_x_hat=eh, seed=self.seed, step=self._calls)
        kw.update(plan_kw)
        Lc = self.global_latent_dims + self.local_latent_dims

        def gm_forward():
            enc.prep(self.gm_flat)
            enc.forward(self.gm_flat, plan.buffer("in8_x", self.dtype, (B, self.H, self.W, 8)),
                        plan.buffer("zcat", self.dtype, (B, Lc)), bool(training) and self.dropout_in_training, eps=ex, u=u,
                        keep1=k1, keep5=k5, seed=self.seed,
                        step=self._calls, sample_offset=kw.get("sample_offset", 0))
        # Augmentator.augment(..., plan=plan) left this batch's padded inputs in in8_x / in8_xh (still current: generation, version counter): the global (GM)
        # encoder then depends on nothing the local encoder's launches produce -- the two encoders of vae/model.py:236-240 run on two HIP streams
        staged = getattr(inputs, "_sv_staged_plan", None) is plan and inputs._sv_staged_gen == plan.in8_gen and inputs._sv_staged_version == inputs._version
        if getattr(inputs, "_sv_staged_plan", None) is plan:
            inputs._sv_staged_plan = None                     # one step per staging
        if staged and (_GM_STREAMS & 1):
            cur, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(cur)                            # (the side stream starts from the state BEFORE the local encoder's phase)
            # the local encoder first: ONE library call enqueues its ~10 launches in ~60 us of host time, and they run while the host is still issuing the GM
            # encoder's ~40 (the other order left the compute stream idle for 370 us of a 1.77 ms step: profiles/r06_j_gm_timeline.txt)
            plan.step(PHASE_PREP | PHASE_FWD_ENCODERS | PHASE_INPUTS_STAGED, **kw)
            with torch.cuda.stream(side), ops.hold_stream():
                gm_forward()
            cur.wait_stream(side)
        else:
            plan.step(PHASE_PREP | PHASE_FWD_ENCODERS | (PHASE_INPUTS_STAGED if staged else 0), **kw)
            gm_forward()
        plan.step(PHASE_FWD_DECODERS | (PHASE_LOSS if want_loss else 0), **kw)
        return plan, enc, kw

    def _side_stream(self):
        # ONE side stream per process and device, whatever the number of models: a stream per model left a second model's stream on the hardware queue of
        # the first one's compute stream, and its cross-stream waits cost 3x the step (profiles/r06_gm_streams.txt: 1.19 -> 3.84 ms)
        # ... and it is the LIBRARY's shared side stream 1 (sv_side_stream: the plan's second weight-gradient stream, idle at the shard sizes this model trains at)
        dev = torch.cuda.current_device()
        s = _SIDE_STREAMS.get(dev)
        if s is None:
            h = C.c_void_p()
            ops.check(_lib.load().sv_side_stream(1, C.byref(h)), "sv_side_stream")
            s = _SIDE_STREAMS[dev] = torch.cuda.ExternalStream(h.value, device=dev)
        return s

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
    out = model._metrics_buf                                 # one native launch (sv_gm_metrics): no ATen stack / mean / cat in the step
    ops.gm_metrics(nll_x, enc.buf["kl2"], nll_h, kl_h, enc.buf["ykl"], model.beta, model.alpha, out)
    return out[:6].clone()


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

    def gm_backward():
        model.gm_grad_flat.zero_()
        enc.backward(model.gm_flat, model.gm_grad_flat, plan.buffer("in8_x", model.dtype, (B, model.H, model.W, 8)),
                     plan.buffer("gz_x", torch.float32, (B, Lc)), model.beta, model.alpha)
    if _GM_STREAMS & 2:
        # the two encoders' adjoints read disjoint column blocks of dz and write disjoint gradient buffers: the global (GM) encoder's on a second stream
        # beside the local encoder's phases (vae/trainer.py:167's tape.gradient has no order between them either)
        cur, side = torch.cuda.current_stream(), model._side_stream()
        side.wait_stream(cur)                                # (behind the decoders' backward, not behind the local encoder's)
        plan.step(PHASE_BWD_ENC_HEADS | PHASE_BWD_ENC_CONVS, **kw)       # one call, issued first: it runs while the host issues the GM encoder's ~60 launches
        with torch.cuda.stream(side), ops.hold_stream():
            gm_backward()
        cur.wait_stream(side)
    else:
        gm_backward()
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
