"""The SPLIT-SPAIR train / test step as ONE native launch sequence (include/splitvae.h: sv_tape_*; csrc/tape.hip).

`NativeStep(model, config, B)` walks the model objects of spair.py ONCE -- the same classes, layer order and branches as
spair/spair.py (SPAIR :19-49, LGSPAIR :52-106, Encoder :368-496, ObjEncoder :246-273, ObjDecoder :341-366, Decoder :500-532,
Image{En,De}coder[Dense] :110-202, Renderer :534-579) and the loss assembly of spair/trainer.py:136-228 -- and records them as
tape nodes over fp32 2-D tensors in one workspace.  `step()` is then a single C call: forward, losses, the hand-written adjoint
of every node, Adam.  No torch autograd, no library GEMM, no ATen kernel between the launches; torch only owns the memory.

Tensors are [rows, cols | row pitch]: per-cell quantities have rows = B*16, images rows = B*H*W; the returned tuple entries are
zero-copy (strided) views of the workspace, valid until the next step.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (TAPE_CLAMP, TAPE_CONV, TAPE_COPY, TAPE_DENSE, TAPE_LOGITNOISE, TAPE_LOSS, TAPE_NOISE, TAPE_RELU, TAPE_RENDER, TAPE_SAMPLE,
                   TAPE_SIGMOID, TAPE_SOFTPLUS, TAPE_STN, TAPE_UNARY, TAPE_UPSAMPLE, TAPE_ZPRES, TapeNode, TapeRunArgs, check)
from .ops import _p, _stream

N_WHERE, N_DEPTH, N_PRES, N_PASS = 4, 1, 1, 8
CELLS = 16                                            # the reference hard-codes the 4x4 grid (spair/spair.py:408, :411)


def _r4(v):
    return (v + 3) // 4 * 4


def _r8(v):
    return (v + 7) // 8 * 8


class T:
    """A tape tensor: id + its 2-D shape."""

    def __init__(self, id_, rows, cols, ld):
        self.id, self.rows, self.cols, self.ld = id_, rows, cols, ld


_NOISE_ON_LANE1 = os.environ.get("SV_SPAIR_NOISE_LANE", "1") != "0"      # (A/B: the NOISE nodes on their consumer's lane)


class NativeStep:
    def __init__(self, model, config, B, training=True):
        self.lib = _lib.load()
        self.model, self.cfg, self.B, self.training = model, config, B, training
        # config.dtype 'bf16': the spatial convolutions take bf16 operands (fp32 accumulation, master weights and activations); Dense
        # layers, STN, Renderer and losses are fp32 either way
        bf16 = (getattr(model, "dtype", "f32") or "f32") == "bf16"
        h = C.c_void_p()
        check(self.lib.sv_tape_create(C.byref(h), B, _lib.SV_BF16 if bf16 else _lib.SV_F32), "sv_tape_create")
        self.h = h
        self.store = model.store
        self.device = model.store.flat.device
        self.n_nodes = 0
        self._group = 0
        self._lane = 0                                # sv_tape_node.lane of the nodes being recorded (include/splitvae.h): 0 = the caller's stream
        self._views = {}
        self.noise = {}                               # name -> (T, kind, std)
        self.out = {}                                 # name -> (T, column offset, columns, shape)
        self._build()
        check(self.lib.sv_tape_finalize(h), "sv_tape_finalize")
        nbytes = self.lib.sv_tape_workspace_bytes(h)
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        check(self.lib.sv_tape_bind(h, _p(self.ws), nbytes, _stream()), "sv_tape_bind")
        self.wsf = self.ws.view(torch.float32)
        oo, mo, nl = C.c_int64(), C.c_int64(), C.c_int32()
        check(self.lib.sv_tape_loss_info(h, C.byref(oo), C.byref(mo), C.byref(nl)), "sv_tape_loss_info")
        self.loss_out = self.wsf[oo.value // 4: oo.value // 4 + 1 + 2 * _lib.TAPE_MAX_LOSS]
        self.metric = self.wsf[mo.value // 4: mo.value // 4 + _lib.TAPE_MAX_LOSS + 2]
        self.grads = torch.zeros_like(self.store.flat)
        self._norm_ws = None
        self._calls = 0
        self._images_flat = self._tview(self.images).view(-1)
        self._loss_list = [self.loss_out[1 + i] for i in range(self.n_loss)]
        self._grad_views = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.sv_tape_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------ recording helpers
    def tensor(self, rows, cols, ld=None, grad=True):
        ld = cols if ld is None else ld
        i = self.lib.sv_tape_tensor(self.h, rows, cols, ld, 1 if grad else 0)
        if i < 0:
            check(i, "sv_tape_tensor")
        return T(i, rows, cols, ld)

    def view(self, t, rows, cols, ld=None):
        ld = cols if ld is None else ld
        i = self.lib.sv_tape_view(self.h, t.id, rows, cols, ld)
        if i < 0:
            check(i, "sv_tape_view")
        return T(i, rows, cols, ld)

    def add(self, kind, **kw):
        n = TapeNode()
        for f in ("x", "y", "t2", "t3", "t4", "t5", "t6"):
            setattr(n, f, -1)
        n.w_off = n.b_off = -1
        n.dyn_idx = n.loss_idx = -1
        n.rep = 1
        n.kind = kind
        n.lane = self._lane
        for k, v in kw.items():
            setattr(n, k, v.id if isinstance(v, T) else v)
        check(self.lib.sv_tape_add(self.h, C.byref(n)), "sv_tape_add(kind %d)" % kind)
        self.n_nodes += 1

    def lane(self, k):
        """with self.lane(k): the nodes recorded inside run on the tape's HIP stream k (an independent branch of the model beside the object pipeline);
        the tape orders every conflicting tensor access across lanes itself, so this is a scheduling hint only."""
        step = self

        class _Lane:
            def __enter__(self_):
                self_.prev, step._lane = step._lane, k

            def __exit__(self_, *exc):
                step._lane = self_.prev
        return _Lane()

    def _offs(self, layer):
        o = self.store.offsets
        return o[layer.ik], o[layer.ib]

    def unary(self, op, x, y, n, xo=0, yo=0, rep=1, p0=0.0, p1=0.0, group=0):
        self.add(TAPE_UNARY, op=op, x=x, y=y, xo=xo, yo=yo, n=n, rep=rep, p0=p0, p1=p1, group=group)

    def new_group(self):
        """Consecutive UNARY nodes of one group are independent of each other and run as one launch (a concat's column blocks ...)."""
        self._group += 1
        return self._group

    def dense(self, layer, x):
        """tf.keras.layers.Dense: y = act(x . kernel + bias); relu rides in the GEMM's epilogue, softplus / sigmoid follow in place."""
        shp = self.store.spec[layer.ik][1]
        K, N = shp[-2], shp[-1]                          # Dense [in, out]; the backbone's 1x1 convolutions [1, 1, in, out]
        assert x.cols == K and (len(shp) == 2 or shp[0] == shp[1] == 1), (layer.name, x.cols, shp)
        y = self.tensor(x.rows, N, _r4(N))
        w, b = self._offs(layer)
        act = layer.activation
        self.add(TAPE_DENSE, x=x, y=y, w_off=w, b_off=b, act=_lib.SV_ACT_RELU if act == "relu" else _lib.SV_ACT_NONE)
        if act == "softplus":
            self.unary(TAPE_SOFTPLUS, y, y, N)
        elif act == "sigmoid":
            self.unary(TAPE_SIGMOID, y, y, N)
        return y

    def conv(self, layer, x, B, H, W):
        """Conv2D(padding='same') on NHWC rows [B*H*W, Cin | pitch r8(Cin)] -> [B*OH*OW, Cout | pitch r8(Cout)] (+ relu in the epilogue)."""
        k, _, cin, cout = self.store.spec[layer.ik][1]
        s = layer.strides
        assert x.ld == _r8(cin) and x.rows == B * H * W, (layer.name, x.ld, x.rows)
        OH, OW = (H + s - 1) // s, (W + s - 1) // s
        y = self.tensor(B * OH * OW, cout, _r8(cout))
        w, b = self._offs(layer)
        relu = layer.activation == "relu"
        self.add(TAPE_CONV, x=x, y=y, w_off=w, b_off=b, B=B, H=H, W=W, C=cin, Cout=cout, k=k, stride=s,
                 act=_lib.SV_ACT_RELU if relu else _lib.SV_ACT_NONE)
        if layer.activation == "sigmoid":
            self.unary(TAPE_SIGMOID, y, y, cout)
        return y, OH, OW

    def upsample(self, x, B, H, W):
        y = self.tensor(B * 4 * H * W, x.cols, x.ld)
        self.add(TAPE_UPSAMPLE, x=x, y=y, B=B, H=H, W=W)
        return y

    def sample(self, mean, mo, sig, so, eps, n, ld=None):
        """Sampling.call (spair/utils.py:19-24)."""
        z = self.tensor(mean.rows, n, n if ld is None else ld)
        self.add(TAPE_SAMPLE, x=mean, xo=mo, t2=sig, o2=so, t3=eps, o3=0, y=z, yo=0, n=n)
        return z

    def draw(self, name, rows, cols, kind, std=1.0, stream_id=0):
        t = self.tensor(rows, cols, cols, grad=False)
        # the Philox draws depend on nothing: always on lane 1, off the object pipeline's dependent chain (their consumers wait on an event)
        with self.lane(1 if _NOISE_ON_LANE1 else self._lane):
            self.add(TAPE_NOISE, y=t, op=0 if kind == "normal" else 1, p0=std, stream_id=stream_id)
        self.noise[name] = t
        return t

    def concat(self, parts, rows):
        """tf.concat(axis=-1) of column blocks (tensor, column offset, columns[, rep])."""
        n = sum(p[2] for p in parts)
        y = self.tensor(rows, n, _r4(n))
        o, g = 0, self.new_group()
        for p in parts:
            self.unary(TAPE_COPY, p[0], y, p[2], xo=p[1], yo=o, rep=p[3] if len(p) > 3 else 1, group=g)
            o += p[2]
        return y

    def loss(self, idx, mode, a, ao, b, bo, R, n, p0=0.0, p1=1.0, dyn=-1):
        self.add(TAPE_LOSS, loss_idx=idx, mode=mode, x=a, xo=ao, t2=b, o2=bo, R=R, n=n, p0=p0, p1=p1, dyn_idx=dyn)

    # ------------------------------------------------------------------------------------------------ the model, recorded
    def _image_encoder(self, enc, x3, x8, eps_name, L, sid):
        """ImageEncoder.call :124-132 / ImageEncoderDense.call :145-154 -> (z, head) with head columns z_mean at 0."""
        B = self.B
        H, W, Cc = self.model.image_size
        if enc.dense:
            h = self.dense(enc.e2, self.dense(enc.e1, self.view(x3, B, H * W * Cc)))
        else:
            a, h1, w1 = self.conv(enc.e1, x8, B, H, W)
            a, h2, w2 = self.conv(enc.e2, a, B, h1, w1)
            a, h3, w3 = self.conv(enc.e3, a, B, h2, w2)
            h = self.view(a, B, h3 * w3 * 128)
        z_mean, z_sig = self.dense(enc.z_mu, h), self.dense(enc.z_sigma, h)
        eps = self.draw(eps_name, B, L, "normal", 1.0, sid)
        z = self.sample(z_mean, 0, z_sig, 0, eps, L, _r4(L))
        return z, z_mean, z_sig

    def _image_decoder(self, dec, z):
        """ImageDecoder.call :171-182 / ImageDecoderDense.call :196-202 -> packed [B*H*W, C]."""
        B = self.B
        H, W, Cc = self.model.image_size
        if dec.dense:
            y = self.dense(dec.d3, self.dense(dec.d2, self.dense(dec.d1, z)))            # d3's sigmoid in place on [B, H*W*C]
            return self.view(y, B * H * W, Cc)
        h = self.view(self.dense(dec.d1, z), B * (H // 8) * (W // 8), 128)
        a, _, _ = self.conv(dec.d2, h, B, H // 8, W // 8)
        a, _, _ = self.conv(dec.d3, self.upsample(a, B, H // 8, W // 8), B, H // 4, W // 4)
        a, _, _ = self.conv(dec.d4, self.upsample(a, B, H // 4, W // 4), B, H // 2, W // 2)
        a, _, _ = self.conv(dec.d5, self.upsample(a, B, H // 2, W // 2), B, H, W)          # sigmoid in place on the 8-channel pitch
        out = self.tensor(B * H * W, Cc)
        self.unary(TAPE_COPY, a, out, Cc)
        return out

    def _build(self):
        m, cfg, B = self.model, self.cfg, self.B
        H, W, Cc = m.image_size
        lg = cfg.model == "lg_spair"
        n = B * CELLS
        L, S = m.encoder.latent_size, m.encoder.object_size
        Cimg = 6 if lg else Cc
        self.images = self.tensor(B * H * W, Cimg, Cimg, grad=False)
        g0 = self.new_group()                                                        # the input split / pad copies: one launch
        if lg:
            x3 = self.tensor(B * H * W, Cc, Cc, grad=False)
            self.unary(TAPE_COPY, self.images, x3, Cc, group=g0)
        else:
            x3 = self.images
        x8 = self.tensor(B * H * W, Cc, 8, grad=False)
        self.unary(TAPE_COPY, self.images, x8, Cc, group=g0)
        O = self.out
        z_l = None
        if lg:                                                                       # LGSPAIR.call :84-90
            xh3 = xh8 = None
            if m.x_hat_encoder.dense:
                xh3 = self.tensor(B * H * W, Cc, Cc, grad=False)
                self.unary(TAPE_COPY, self.images, xh3, Cc, xo=3, group=g0)
            else:
                xh8 = self.tensor(B * H * W, Cc, 8, grad=False)
                self.unary(TAPE_COPY, self.images, xh8, Cc, xo=3, group=g0)

        def image_encoders():
            # the x-hat and background networks are independent of the object pipeline until the renderer / the losses: their own lanes (HIP streams)
            with self.lane(1):
                zl = self._image_encoder(m.x_hat_encoder, xh3, xh8, "eps_l", m.Ll, 6)
            with self.lane(2):
                zb = self._image_encoder(m.bg_encoder, x3, x8, "eps_bg", m.Lbg, 5)
            return zl, zb
        # Tape order = the order the host issues the launches in.  The backbone's three convolutions (the longest launches of lane 0) are recorded BEFORE the image
        # encoders, so that the compute stream has work while the host issues the other lane's ~16 launches.  Measured: no difference in steady state (2.444 against
        # 2.44-2.46 ms: the host runs a step ahead of the GPU; profiles/r06_spair_lanes.txt).  SV_SPAIR_BRANCH_FIRST=1: the reference's source order (spair/spair.py:84-90).
        branch_first = os.environ.get("SV_SPAIR_BRANCH_FIRST", "0") == "1"
        if lg and branch_first:
            (z_l, zl_mean, zl_sig), (z_bg, zb_mean, zb_sig) = image_encoders()
        # ---- Encoder.call :403-496
        e = m.encoder
        a, h1, w1 = self.conv(e.conv1, x8, B, H, W)
        a, h2, w2 = self.conv(e.conv2, a, B, h1, w1)
        a, Hc, Wc = self.conv(e.conv3, a, B, h2, w2)
        assert Hc * Wc == CELLS, (Hc, Wc)
        if lg and not branch_first:
            (z_l, zl_mean, zl_sig), (z_bg, zb_mean, zb_sig) = image_encoders()
        fv = self.dense(e.z3, self.dense(e.z2, self.dense(e.z1, a)))                    # 1x1 convolutions = Dense over the cells
        Fv = fv.cols
        if lg and m.concat_backbone:
            zl16 = self.dense(e.dense_z_l[1], self.dense(e.dense_z_l[0], z_l))
            fv = self.concat([(fv, 0, Fv), (zl16, 0, 16, CELLS)], n)                    # tf.tile(z_l[:, newaxis, :], [1, 16, 1]) :408
            Fv = fv.cols
        # box network :424-437 -- o = [mean 0:4 | sig 4:8 | f1 8:16], activations in place on their column blocks
        o = self.dense(e.dense_z_where[2], self.dense(e.dense_z_where[1], self.dense(e.dense_z_where[0], fv)))
        g = self.new_group()
        self.unary(TAPE_SOFTPLUS, o, o, N_WHERE, xo=N_WHERE, yo=N_WHERE, p0=-1.0, group=g)
        self.unary(TAPE_RELU, o, o, N_PASS, xo=2 * N_WHERE, yo=2 * N_WHERE, group=g)
        zw = self.sample(o, 0, o, N_WHERE, self.draw("eps_where", n, N_WHERE, "normal", 1.0, 1), N_WHERE)
        # attribute network :440-441
        glimpses = self.tensor(n * S * S, Cc)
        self.add(TAPE_STN, x=x3, t2=zw, y=glimpses, B=B, H=H, W=W, C=Cc, Ho=S, Wo=S, Hc=Hc, Wc=Wc, inverse=0)
        g8 = self.tensor(n * S * S, Cc, 8)
        self.unary(TAPE_COPY, glimpses, g8, Cc)
        oe = e.obj_encoder
        a, s1, _ = self.conv(oe.conv1, g8, n, S, S)
        a, s2, _ = self.conv(oe.conv2, a, n, s1, s1)
        hh = self.dense(oe.dense1, self.view(a, n, s2 * s2 * 64))
        zt_mean, zt_sig = self.dense(oe.z_what_mu, hh), self.dense(oe.z_what_sigma, hh)
        zt = self.sample(zt_mean, 0, zt_sig, 0, self.draw("eps_what", n, L, "normal", 1.0, 2), L, _r4(L))
        # depth network :455-461 -- o2 = [mean 0 | sig 1 | f2 2:10]
        din = self.concat([(fv, 0, Fv), (o, 2 * N_WHERE, N_PASS), (zw, 0, N_WHERE), (zt, 0, L)], n)
        o2 = self.dense(e.dense_z_depth[1], self.dense(e.dense_z_depth[0], din))
        g = self.new_group()
        self.unary(TAPE_SOFTPLUS, o2, o2, N_DEPTH, xo=N_DEPTH, yo=N_DEPTH, group=g)
        self.unary(TAPE_RELU, o2, o2, N_PASS, xo=2 * N_DEPTH, yo=2 * N_DEPTH, group=g)
        zd = self.sample(o2, 0, o2, N_DEPTH, self.draw("eps_depth", n, N_DEPTH, "normal", 1.0, 3), N_DEPTH)
        # presence network :464-467 + concrete_binary_pre_sigmoid_sample (spair/utils.py:14-17)
        pin = self.concat([(fv, 0, Fv), (o2, 2 * N_DEPTH, N_PASS), (zw, 0, N_WHERE), (zt, 0, L), (zd, 0, N_DEPTH)], n)
        po = self.dense(e.dense_z_pres[1], self.dense(e.dense_z_pres[0], pin))
        logits = self.tensor(n, 1)
        self.unary(TAPE_CLAMP, po, logits, 1, p0=-10.0, p1=10.0)
        pre = self.tensor(n, 1)
        self.add(TAPE_LOGITNOISE, x=logits, t2=self.draw("u_pres", n, N_PRES, "uniform", 1.0, 4), y=pre, n=1, p0=float(e.tau))
        pres = self.tensor(n, 1)
        self.unary(TAPE_SIGMOID, pre, pres, 1)
        # ---- LGSPAIR.call :91-104 / SPAIR.call :37-44
        zin, Lw = zt, L
        bg = None
        if lg:
            with self.lane(1):
                xh = self._image_decoder(m.x_hat_decoder, z_l)
            with self.lane(2):
                if m.concat_z_bg:
                    z_bg = self.concat([(z_bg, 0, m.Lbg), (z_l, 0, m.Ll)], B)
                bg = self._image_decoder(m.bg_decoder, z_bg)
            if m.concat_z_what:
                zin = self.concat([(zt, 0, L), (z_l, 0, m.Ll, CELLS)], n)
                Lw = L + m.Ll
        elif m.bg_model is not None:
            with self.lane(2):
                z_bg, zb_mean, zb_sig = self._image_encoder(m.bg_model.enc, x3, x8, "eps_bg", m.bg_latent_size, 5)
                bg = self._image_decoder(m.bg_model.dec, z_bg)
        if bg is None:
            bg = self.tensor(B * H * W, Cc, Cc, grad=False)                               # bg_recon = 0.0 (:39): never written, stays zero
        # Decoder.call :514-532
        od = m.decoder.obj_decoder
        a = self.view(self.dense(od.d1, self.dense(od.d0, zin)), n * (S // 4) * (S // 4), 32)
        a, _, _ = self.conv(od.d2, a, n, S // 4, S // 4)
        a, _, _ = self.conv(od.d3, self.upsample(a, n, S // 4, S // 4), n, S // 2, S // 2)
        a, _, _ = self.conv(od.d5, self.upsample(a, n, S // 2, S // 2), n, S, S)
        obj = self.tensor(n * S * S, Cc + 1)                                               # sigmoid(rgb) | sigmoid(alpha), packed
        self.unary(TAPE_SIGMOID, a, obj, Cc + 1)
        full = self.tensor(n * H * W, Cc + 1)
        bbox = self.tensor(n, 4, 4, grad=False)
        self.add(TAPE_STN, x=obj, t2=zw, y=full, t3=bbox, B=B, H=S, W=S, C=Cc + 1, Ho=H, Wo=W, Hc=Hc, Wc=Wc, inverse=1)
        rn = self.draw("render", n * H * W, Cc, "normal", 0.01, 7) if self.training else None
        x_recon = self.tensor(B * H * W, Cc)
        self.add(TAPE_RENDER, x=full, t2=bg, t3=zd, t4=pres, t5=logits, t6=rn if rn is not None else -1, y=x_recon, B=B, H=H, W=W, C=Cc, R=CELLS,
                 training=1 if self.training else 0)
        # ---- losses (spair/trainer.py:148-160, :170-216): per-image sums, one node each
        self.loss(0, 0, self.images, 0, x_recon, 0, H * W, Cc)                               # x_recon_loss
        self.loss(1, 2, o, 0, o, N_WHERE, CELLS, 2, p1=0.5, dyn=1)                           # zoom_kl against N(dyn[1], 0.5)
        self.loss(2, 1, zt_mean, 0, zt_sig, 0, CELLS, L)                                     # what_kl
        self.loss(3, 1, o, 2, o, N_WHERE + 2, CELLS, 2)                                      # where_kl
        self.loss(4, 1, o2, 0, o2, N_DEPTH, CELLS, N_DEPTH)                                  # depth_kl
        self.add(TAPE_ZPRES, loss_idx=5, x=pres, t2=logits, t3=pre, R=CELLS, p0=float(cfg.tau), dyn_idx=0)
        self.n_loss = 6
        if lg:
            self.loss(6, 1, zb_mean, 0, zb_sig, 0, 1, m.Lbg)
            self.loss(7, 1, zl_mean, 0, zl_sig, 0, 1, m.Ll)
            self.loss(8, 0, self.images, 3, xh, 0, H * W, Cc)
            self.n_loss = 9
        elif m.bg_model is not None:
            self.loss(6, 1, zb_mean, 0, zb_sig, 0, 1, m.bg_latent_size)
            self.n_loss = 7
        # reported list = the `losses` of train_step (:158-160, :208-216, :227)
        R = [[0.0] * _lib.TAPE_MAX_LOSS for _ in range(_lib.TAPE_MAX_LOSS)]
        for i in range(self.n_loss):
            R[i][i] = 1.0
        if lg and not cfg.split_z_l and m.concat_z_bg:
            R[6][7] = 1.0                                                                  # kl(concat(z_bg, z_l)) = kl_bg + kl_l (:178-181)
        flat = (C.c_float * (_lib.TAPE_MAX_LOSS * self.n_loss))(*[v for row in R[:self.n_loss] for v in row])
        check(self.lib.sv_tape_set_report(self.h, flat, self.n_loss), "sv_tape_set_report")
        # ---- the returned tuple (spair/spair.py:45-49, :105-106): (tensor, column offset, columns, shape)
        c4 = (B, 4, 4)
        O["x_recon"] = (x_recon, 0, Cc, (B, H, W, Cc))
        O["z_what"] = (zin, 0, Lw, c4 + (Lw,))
        O["z_what_mean"], O["z_what_sigma"] = (zt_mean, 0, L, c4 + (L,)), (zt_sig, 0, L, c4 + (L,))
        O["z_where"], O["z_where_mean"], O["z_where_sigma"] = (zw, 0, 4, c4 + (4,)), (o, 0, 4, c4 + (4,)), (o, 4, 4, c4 + (4,))
        O["z_depth"], O["z_depth_mean"], O["z_depth_sigma"] = (zd, 0, 1, c4 + (1,)), (o2, 0, 1, c4 + (1,)), (o2, 1, 1, c4 + (1,))
        O["z_pres"], O["z_pres_logits"], O["z_pres_pre_sigmoid"] = (pres, 0, 1, c4 + (1,)), (logits, 0, 1, c4 + (1,)), (pre, 0, 1, c4 + (1,))
        O["all_glimpses"] = (glimpses, 0, Cc, (B, CELLS, S, S, Cc))
        O["obj_recon_unnorm"], O["obj_recon_alpha"] = (obj, 0, Cc, (B, CELLS, S, S, Cc)), (obj, Cc, 1, (B, CELLS, S, S, 1))
        O["obj_full_recon_unnorm"] = (full, 0, Cc + 1, (B, CELLS, H, W, Cc + 1))
        O["obj_bbox_mask"] = (bbox, 0, 4, (B, CELLS, 4))
        if lg or m.bg_model is not None:
            Lb = z_bg.cols
            O["z_bg"], O["z_bg_mean"], O["z_bg_sig"] = (z_bg, 0, Lb, (B, Lb)), (zb_mean, 0, zb_mean.cols, (B, zb_mean.cols)), (zb_sig, 0, zb_sig.cols, (B, zb_sig.cols))
        if lg:
            O["x_hat_recon"] = (xh, 0, Cc, (B, H, W, Cc))
            O["z_l"], O["z_l_mean"], O["z_l_sig"] = (z_l, 0, m.Ll, (B, m.Ll)), (zl_mean, 0, m.Ll, (B, m.Ll)), (zl_sig, 0, m.Ll, (B, m.Ll))

    # ------------------------------------------------------------------------------------------------ running
    def _tview(self, t, grad=False):
        key = (t.id, grad)
        v = self._views.get(key)
        if v is None:
            off, goff = C.c_int64(), C.c_int64()
            check(self.lib.sv_tape_tensor_info(self.h, t.id, C.byref(off), C.byref(goff)), "sv_tape_tensor_info")
            o = (goff.value if grad else off.value) // 4
            v = self._views[key] = torch.as_strided(self.wsf, (t.rows, t.cols), (t.ld, 1), o)
        return v

    def output(self, name):
        """One entry of the model's returned tuple: a view of the workspace when its columns are contiguous, else a copy."""
        t, co, nc, shape = self.out[name]
        return self._tview(t)[:, co:co + nc].reshape(shape)

    def outputs(self, names):
        """The model's returned tuple, materialised entry by entry on access (a training loop that ignores it pays nothing)."""
        return LazyOutputs(self, list(names))

    def loss_weights(self, sc):
        """total = sum_i w_i * loss_i (spair/trainer.py:165-167, :176-207, :222-226) for the step's annealed scalars."""
        cfg, m = self.cfg, self.model
        w = [0.0] * self.n_loss
        lg = cfg.model == "lg_spair"
        beta = cfg.beta if lg else sc["annealed_beta"]
        w[0] = cfg.reconstruction_weight
        w[1] = w[3] = w[4] = w[5] = beta
        w[2] = beta * cfg.z_what_beta
        if lg:
            w[6] = cfg.z_bg_beta
            w[8] = 1.0
            if cfg.split_z_l:
                w[7] = cfg.z_l_beta
            else:
                w[7] = (cfg.z_bg_beta if m.concat_z_bg else 0.0) + (beta * cfg.z_what_beta * CELLS if m.concat_z_what else 0.0)
        elif m.bg_model is not None:
            w[6] = cfg.z_bg_beta
        return w

    def run(self, images, sc, optimizer=None, noise=None, backward=True, accumulate_metrics=False):
        """One step on `images` [B,H,W,C]: forward + losses (+ adjoint (+ Adam when an optimizer is given)).  sc: spair_trainer.step_scalars."""
        self._images_flat.copy_(images.reshape(-1))
        if noise:
            for k, t in self.noise.items():
                self._tview(t).view(-1).copy_(noise[k].reshape(-1).to(torch.float32))
        st = self.store
        a = TapeRunArgs()
        a.params, a.grads = _p(st.flat).value, _p(self.grads).value
        w = self.loss_weights(sc)
        wa = (C.c_float * len(w))(*w)
        a.loss_weights, a.n_weights = wa, len(w)
        a.dyn[0], a.dyn[1] = float(sc["prior_prob"]), float(sc["zoom_mean"])
        # the Philox step of the NOISE nodes comes from a MODEL-level counter shared by every NativeStep of the model (other batch sizes, the
        # last partial batch, train / test): a per-instance counter starting at 0 replayed the same eps / u_pres / render-noise streams
        # (ADVICE r03).  A caller that resumes a run sets model.noise_step from its saved step count.
        a.seed, a.step = int(getattr(self.model, "seed", 0)) + 1, int(getattr(self.model, "noise_step", 0))
        self.model.noise_step = a.step + 1
        self._calls += 1
        a.pinned_noise = 1 if noise else 0
        a.phases = _lib.TAPE_PHASE_FORWARD | (_lib.TAPE_PHASE_BACKWARD if backward else 0)
        a.accumulate_metrics = 1 if accumulate_metrics else 0
        a.n_params = st.flat.numel()
        if optimizer is not None and backward:
            m_, v_ = optimizer.slots(st.flat)
            optimizer.iterations += 1
            a.phases |= _lib.TAPE_PHASE_ADAM
            a.adam_m, a.adam_v = _p(m_).value, _p(v_).value
            a.lr, a.beta1, a.beta2, a.adam_eps, a.t = float(optimizer.learning_rate), optimizer.beta_1, optimizer.beta_2, optimizer.epsilon, optimizer.iterations
            if optimizer.clip_in_apply:
                if self._norm_ws is None:
                    self._norm_ws = torch.empty((len(st.spec) * 256,), dtype=torch.float32, device=self.device)
                a.clipnorm, a.tensor_off, a.n_tensors, a.norm_ws = float(optimizer.clipnorm), _p(st.tensor_off).value, len(st.spec), _p(self._norm_ws).value
        check(self.lib.sv_tape_run(self.h, C.byref(a), _stream()), "sv_tape_run")
        return self.loss_out

    def grad_views(self):
        import math
        if self._grad_views is None:
            o = self.store.offsets
            self._grad_views = [self.grads[o[i]:o[i] + math.prod(shp)].view(shp) for i, (_, shp) in enumerate(self.store.spec)]
        return self._grad_views


class LazyOutputs:
    """Sequence over the native step's returned tensors; each entry is built (a view, or a copy when its columns are strided) on access."""

    def __init__(self, ns, names):
        self.ns, self.names = ns, names

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return LazyOutputs(self.ns, self.names[i])
        return self.ns.output(self.names[i])

    def __iter__(self):
        return (self.ns.output(k) for k in self.names)

    def __add__(self, other):
        return tuple(self) + tuple(other)
