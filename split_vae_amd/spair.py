"""SPAIR / SPLIT-SPAIR models (config 5; spair/spair.py of 51616/split-vae) assembled from the device operators.

Same surface as the reference module: `get_model(config)` -> `SPAIR` | `LGSPAIR`; `model(images, training)` returns the
reference's output tuple (spair/spair.py:35-49, :84-106); `model.trainable_variables` lists (name, tensor) in the reference's
layer-tracking order.  What runs where:
  * every Conv2D (backbone 4x4 stride 2/2/3 + 1x1, object encoder / decoder, the conv image encoders / decoders):
    the MFMA conv kernels through `split_vae::conv2d_*` (torch_ops.conv2d; fused 2x resize where the kernel covers the geometry);
  * the spatial transformers (glimpse gather, inverse paste) and the Renderer: stn.hip / spair_render.hip through
    `split_vae::stn_sample_*` / `split_vae::spair_render_*`;
  * Dense layers: plain library GEMMs (rocBLAS through torch.addmm, fp32) -- they are ordinary [N, in] x [in, out] products;
  * the pointwise glue between them (softplus / sigmoid / concat / sampling arithmetic): torch elementwise ops, autograd-paired.
The first assembled step: correct and differentiable end to end on the device, not launch-optimised (SPLIT-SPAIR).
fp32 throughout, like the reference.  No CPU path: the operators exist for HIP tensors only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, torch_ops as T
from .utils import dotdict

N_WHERE, N_DEPTH, N_PRES, N_PASS = 4, 1, 1, 8          # spair/spair.py:374-376, :390


def _r8(v):
    return (v + 7) // 8 * 8


class VarStore:
    """The model's variables as views of ONE flat fp32 device buffer (what the clipnorm Adam kernel sweeps), in creation order."""

    def __init__(self):
        self.spec = []            # (name, shape)
        self.flat = None
        self.convs = []           # the Conv2D layers (their compute dtype is a model-wide switch)

    def add(self, name, shape):
        self.spec.append((name, tuple(shape)))
        return len(self.spec) - 1

    def finalize(self, device, seed):
        rng = np.random.Generator(np.random.PCG64(seed))
        offs, chunks, o = [], [], 0
        for name, shp in self.spec:
            n = int(np.prod(shp))
            if name.endswith("/bias"):
                chunks.append(np.zeros((n,), np.float32))
            else:                                                    # Keras default: glorot_uniform
                rf = shp[0] * shp[1] if len(shp) == 4 else 1
                lim = math.sqrt(6.0 / (rf * shp[-2] + rf * shp[-1]))
                chunks.append(rng.uniform(-lim, lim, size=n).astype(np.float32))
            offs.append(o)
            pad = (-n) % 4                                           # every variable starts 16-B aligned (the kernels read 16-B pieces)
            if pad:
                chunks.append(np.zeros((pad,), np.float32))          # the gaps stay zero: zero gradient, zero Adam moments, no update
            o += n + pad
        offs.append(o)
        self.offsets = offs
        self.n_params = sum(int(np.prod(shp)) for _, shp in self.spec)
        self.flat = torch.from_numpy(np.concatenate(chunks)).to(device)
        self.tensor_off = torch.tensor(offs, dtype=torch.int64, device=device)
        # leaves sharing the flat storage: autograd returns one gradient per variable, the optimizer updates the flat buffer in place
        self.vars = [self.flat[offs[i]:offs[i] + int(np.prod(shp))].view(shp).detach().requires_grad_(True) for i, (_, shp) in enumerate(self.spec)]


class _Layer:
    def __init__(self, store, name, kshape):
        self.store, self.name = store, name
        self.ik = store.add(name, kshape)
        self.ib = store.add(name + "/bias", (kshape[-1],))

    @property
    def kernel(self):
        return self.store.vars[self.ik]

    @property
    def bias(self):
        return self.store.vars[self.ib]


def _activation(y, act):
    if act == "relu":
        return F.relu(y)
    if act == "softplus":
        return F.softplus(y)
    if act == "sigmoid":
        return torch.sigmoid(y)
    return y


class Dense(_Layer):
    """tf.keras.layers.Dense on the last axis."""

    def __init__(self, store, name, n_in, n_out, activation=None):
        super().__init__(store, name, (n_in, n_out))
        self.activation = activation

    def __call__(self, x):
        y = torch.addmm(self.bias, x.reshape(-1, x.shape[-1]), self.kernel).reshape(*x.shape[:-1], -1)
        return _activation(y, self.activation)


class Conv2D(_Layer):
    """Conv2D(padding='same'); `resize_in`: the layer reads the 2x bilinear resize of its input (tf.image.resize before it)."""

    def __init__(self, store, name, k, cin, cout, strides=1, activation=None, resize_in=False):
        super().__init__(store, name, (k, k, cin, cout))
        self.cin, self.cout, self.strides, self.activation, self.resize_in = cin, cout, strides, activation, resize_in
        self.fused = resize_in                   # cleared when the fused-resize kernel does not cover this geometry
        store.convs.append(self)
        self.compute_dtype = torch.float32       # torch.bfloat16: bf16 MFMA operands, fp32 accumulation and fp32 master weights

    def __call__(self, x, out_lp=False):
        """out_lp: with bf16 compute, hand the bf16 activation to the caller as it is (the next layer is another convolution that
        would cast it straight back: saves the two casts and their adjoints per pair of layers)."""
        if self.kernel.shape[0] == 1 and self.strides == 1 and not self.resize_in:
            # a 1x1 convolution IS a Dense layer over the pixels (the backbone's z1 / z2 / z3 on the 4x4 cell grid): a plain library GEMM
            y = torch.addmm(self.bias, x.reshape(-1, self.cin), self.kernel.view(self.cin, self.cout)).reshape(*x.shape[:-1], self.cout)
            return _activation(y, self.activation)
        if x.shape[-1] != _r8(self.cin):                             # the kernels read 8-channel pixel pitches (pad channels zero)
            x = F.pad(x, (0, _r8(self.cin) - x.shape[-1]))
        if x.dtype != self.compute_dtype:
            x = x.to(self.compute_dtype)                             # (autograd casts the input gradient back)
        act = "relu" if self.activation == "relu" else None
        y = None
        if self.fused:
            try:
                y = T.conv2d(x, self.kernel, self.bias, self.strides, act, True, False)
            except _lib.SplitVaeError as e:
                if "SV_E_UNSUPPORTED" not in str(e):
                    raise
                self.fused = False
        if y is None:
            if self.resize_in:
                x = T.upsample2x(x)
            y = T.conv2d(x, self.kernel, self.bias, self.strides, act, False, False)
        if y.shape[-1] != self.cout:
            y = y[..., :self.cout]
        if y.dtype != torch.float32 and not (out_lp and (act or self.activation is None)):
            y = y.float()
        return y if act else _activation(y, self.activation)


def _sample(mean, sig, eps):
    """Sampling.call (spair/utils.py:19-24)."""
    return mean + sig * eps


class _Noise:
    """The forward's random draws: pinned tensors from the caller (tests) or torch's device generator."""

    def __init__(self, given, device, generator=None):
        self.given, self.device, self.generator = given or {}, device, generator

    def normal(self, name, shape, std=1.0):
        if name in self.given:
            return self.given[name]
        return torch.randn(shape, device=self.device, generator=self.generator) * std

    def uniform(self, name, shape):
        if name in self.given:
            return self.given[name]
        return torch.rand(shape, device=self.device, generator=self.generator)


class ImageEncoder:
    """spair/spair.py:110-132 (conv) / ImageEncoderDense :135-154."""

    def __init__(self, store, image_size, latent_size, name, dense=False, mu="z_mu", sigma="z_sigma"):
        H, W, C = image_size
        self.dense = dense
        if dense:
            self.e1 = Dense(store, name + "/e1", H * W * C, 1024, "relu")
            self.e2 = Dense(store, name + "/e2", 1024, 500, "relu")
            feat = 500
        else:
            self.e1 = Conv2D(store, name + "/e1", 3, C, 32, 2, "relu")
            self.e2 = Conv2D(store, name + "/e2", 3, 32, 64, 2, "relu")
            self.e3 = Conv2D(store, name + "/e3", 3, 64, 128, 2, "relu")
            feat = ((H + 7) // 8) * ((W + 7) // 8) * 128
        self.z_mu = Dense(store, f"{name}/{mu}", feat, latent_size)
        self.z_sigma = Dense(store, f"{name}/{sigma}", feat, latent_size, "softplus")

    def __call__(self, x, eps):
        B = x.shape[0]
        h = self.e2(self.e1(x.reshape(B, -1))) if self.dense else self.e3(self.e2(self.e1(x, out_lp=True), out_lp=True)).reshape(B, -1)
        z_mean, z_sig = self.z_mu(h), self.z_sigma(h)
        return _sample(z_mean, z_sig, eps), z_mean, z_sig


class ImageDecoder:
    """spair/spair.py:157-182 (conv; the three resizes are fused into the convs that read them) / ImageDecoderDense :185-202."""

    def __init__(self, store, image_size, n_in, name, dense=False):
        H, W, C = image_size
        self.image_size, self.dense = image_size, dense
        if dense:
            self.d1 = Dense(store, name + "/d1", n_in, 500, "relu")
            self.d2 = Dense(store, name + "/d2", 500, 1024, "relu")
            self.d3 = Dense(store, name + "/d3", 1024, H * W * C, "sigmoid")
        else:
            self.d1 = Dense(store, name + "/d1", n_in, H // 8 * W // 8 * 128, "relu")
            self.d2 = Conv2D(store, name + "/d2", 3, 128, 128, 1, "relu")
            self.d3 = Conv2D(store, name + "/d3", 3, 128, 64, 1, "relu", resize_in=True)
            self.d4 = Conv2D(store, name + "/d4", 3, 64, 32, 1, "sigmoid", resize_in=True)
            self.d5 = Conv2D(store, name + "/d5", 3, 32, C, 1, "sigmoid", resize_in=True)

    def __call__(self, z):
        H, W, C = self.image_size
        if self.dense:
            return self.d3(self.d2(self.d1(z))).reshape(-1, H, W, C)
        x = self.d1(z).reshape(-1, H // 8, W // 8, 128)
        return self.d5(self.d4(self.d3(self.d2(x, out_lp=True), out_lp=True)))      # d4's sigmoid is a torch op: fp32 from there


class BackgroundModel:
    """spair/spair.py:205-244."""

    def __init__(self, store, image_size, bg_latent_size, name="bg_model"):
        self.enc = ImageEncoder(store, image_size, bg_latent_size, name, False, "z_bg_mu", "z_bg_sigma")
        self.dec = ImageDecoder(store, image_size, bg_latent_size, name, False)

    def __call__(self, x, eps):
        z_bg, z_bg_mean, z_bg_sig = self.enc(x, eps)
        return self.dec(z_bg), z_bg, z_bg_mean, z_bg_sig


class ObjEncoder:
    """spair/spair.py:246-273."""

    def __init__(self, store, latent_size, object_size, channels, name="encoder/obj_encoder"):
        self.conv1 = Conv2D(store, name + "/conv1", 3, channels, 32, 2, "relu")
        self.conv2 = Conv2D(store, name + "/conv2", 3, 32, 64, 2, "relu")
        self.dense1 = Dense(store, name + "/dense1", (object_size // 4) ** 2 * 64, latent_size * 2, "relu")
        self.z_what_mu = Dense(store, name + "/z_what_mu", latent_size * 2, latent_size)
        self.z_what_sigma = Dense(store, name + "/z_what_sigma", latent_size * 2, latent_size, "softplus")

    def __call__(self, glimpses, eps):
        g = glimpses.reshape(-1, *glimpses.shape[2:])
        x = self.conv2(self.conv1(g, out_lp=True))
        h = self.dense1(x.reshape(x.shape[0], -1))
        mean, sig = self.z_what_mu(h), self.z_what_sigma(h)
        return _sample(mean, sig, eps), mean, sig


class ObjDecoder:
    """spair/spair.py:341-366."""

    def __init__(self, store, object_size, num_channel, latent_size, n_in, name="decoder/obj_decoder"):
        self.object_size, self.num_channel = object_size, num_channel
        self.d0 = Dense(store, name + "/d0", n_in, latent_size * 2, "relu")
        self.d1 = Dense(store, name + "/d1", latent_size * 2, object_size // 4 * object_size // 4 * 32, "relu")
        self.d2 = Conv2D(store, name + "/d2", 3, 32, 64, 1, "relu")
        self.d3 = Conv2D(store, name + "/d3", 3, 64, 32, 1, "relu", resize_in=True)
        self.d5 = Conv2D(store, name + "/d5", 3, 32, num_channel + 1, 1, None, resize_in=True)      # recon + alpha channel

    def __call__(self, z_what):
        S = self.object_size
        x = self.d1(self.d0(z_what)).reshape(-1, S // 4, S // 4, 32)
        x = self.d5(self.d3(self.d2(x, out_lp=True), out_lp=True))
        return torch.sigmoid(x[..., :self.num_channel]), torch.sigmoid(x[..., self.num_channel:])


class Encoder:
    """spair/spair.py:368-496 (glimpse_local=False: the reference's LGGlimpseSPAIR is referenced but never defined)."""

    def __init__(self, store, object_size, latent_size, tau, channels, concat=False, local_latent_size=None, name="encoder"):
        self.tau, self.latent_size, self.object_size = tau, latent_size, object_size
        self.conv1 = Conv2D(store, name + "/conv1", 4, channels, 128, 2, "relu")
        self.conv2 = Conv2D(store, name + "/conv2", 4, 128, 128, 2, "relu")
        self.conv3 = Conv2D(store, name + "/conv3", 4, 128, 128, 3, "relu")
        self.z1 = Conv2D(store, name + "/z1", 1, 128, 128, 1, "relu")
        self.z2 = Conv2D(store, name + "/z2", 1, 128, 128, 1, "relu")
        self.z3 = Conv2D(store, name + "/z3", 1, 128, 100, 1, "relu")
        fv = 100 + (16 if concat else 0)
        L = latent_size
        self.dense_z_where = [Dense(store, name + "/dense_z_where/0", fv, 128, "relu"), Dense(store, name + "/dense_z_where/1", 128, 64, "relu"),
                              Dense(store, name + "/dense_z_where/2", 64, 2 * N_WHERE + N_PASS)]
        self.dense_z_depth = [Dense(store, name + "/dense_z_depth/0", fv + N_PASS + N_WHERE + L, 64, "relu"),
                              Dense(store, name + "/dense_z_depth/1", 64, 2 * N_DEPTH + N_PASS)]
        self.dense_z_pres = [Dense(store, name + "/dense_z_pres/0", fv + N_PASS + N_WHERE + L + N_DEPTH, 64, "relu"),
                             Dense(store, name + "/dense_z_pres/1", 64, N_PRES)]
        self.obj_encoder = ObjEncoder(store, latent_size, object_size, channels)
        self.dense_z_l = [Dense(store, name + "/dense_z_l/0", local_latent_size, 16, "relu"),
                          Dense(store, name + "/dense_z_l/1", 16, 16, "relu")] if concat else None

    @staticmethod
    def _seq(layers, x):
        for l in layers:
            x = l(x)
        return x

    def __call__(self, inputs, noise, training=False):
        x, z_l = inputs if isinstance(inputs, (list, tuple)) else (inputs, None)
        B = x.shape[0]
        z = self.z3(self.z2(self.z1(self.conv3(self.conv2(self.conv1(x, out_lp=True), out_lp=True)))))
        Hc, Wc = z.shape[1], z.shape[2]
        n = B * Hc * Wc
        fv = z.reshape(n, z.shape[-1])
        if z_l is not None:
            zl = self._seq(self.dense_z_l, z_l)
            zl = zl[:, None, :].expand(-1, 16, -1).reshape(-1, zl.shape[-1])          # tf.tile(z_l[:,newaxis,:],[1,16,1]) :408
            fv = torch.cat([fv, zl], dim=-1)
        # box network (:424-437)
        o = self._seq(self.dense_z_where, fv)
        zw_mean, zw_sig, f1 = torch.split(o, [N_WHERE, N_WHERE, N_PASS], dim=-1)
        zw_sig = F.softplus(zw_sig - 1.0)
        zw = _sample(zw_mean, zw_sig, noise.normal("eps_where", (n, N_WHERE)))
        f1 = F.relu(f1)
        z_where = zw.reshape(B, Hc, Wc, N_WHERE)
        # attribute network (:440-441): glimpses cut by the spatial transformer, encoded per cell
        glimpses, _ = T.stn_sample(x.contiguous(), z_where.contiguous(), self.object_size, self.object_size, False)
        zt, zt_mean, zt_sig = self.obj_encoder(glimpses, noise.normal("eps_what", (n, self.latent_size)))
        prog = torch.cat([zw, zt], dim=1)
        # depth network (:455-461)
        o = self._seq(self.dense_z_depth, torch.cat([fv, f1, prog], dim=1))
        zd_mean, zd_sig, f2 = torch.split(o, [N_DEPTH, N_DEPTH, N_PASS], dim=-1)
        zd_sig = F.softplus(zd_sig)
        zd = _sample(zd_mean, zd_sig, noise.normal("eps_depth", (n, N_DEPTH)))
        prog = torch.cat([prog, zd], dim=1)
        f2 = F.relu(f2)
        # presence network (:464-467) + concrete_binary_pre_sigmoid_sample (spair/utils.py:14-17)
        logits = torch.clamp(self._seq(self.dense_z_pres, torch.cat([fv, f2, prog], dim=1)), -10.0, 10.0)
        u = noise.uniform("u_pres", (n, N_PRES))
        pre = (logits + (torch.log(u + 1e-8) - torch.log(1.0 - u + 1e-8))) / self.tau
        pres = torch.sigmoid(pre)
        r = lambda t: t.reshape(B, Hc, Wc, -1)
        return (r(zt), r(zt_mean), r(zt_sig), r(zw), r(zw_mean), r(zw_sig), r(zd), r(zd_mean), r(zd_sig), r(pres), r(logits), r(pre),
                glimpses)


class Decoder:
    """spair/spair.py:500-532."""

    def __init__(self, store, image_size, object_size, latent_size, n_in):
        self.image_size, self.object_size, self.num_channel = image_size, object_size, image_size[2]
        self.obj_decoder = ObjDecoder(store, object_size, self.num_channel, latent_size, n_in)

    def __call__(self, z_what, z_where):
        B, Hc, Wc, _ = z_where.shape
        S, C = self.object_size, self.num_channel
        rgb, alpha = self.obj_decoder(z_what)
        rgb = rgb.reshape(B, Hc * Wc, S, S, C)
        alpha = alpha.reshape(B, Hc * Wc, S, S, 1)
        full, bbox = T.stn_sample(torch.cat([rgb, alpha], dim=4).contiguous(), z_where.contiguous(), self.image_size[0], self.image_size[1], True)
        return rgb, alpha, full, bbox


class Renderer:
    """spair/spair.py:534-579 (spair_render.hip)."""

    def __init__(self, num_channel):
        self.num_channel = num_channel

    def __call__(self, full, bg, z_depth, z_pres, z_pres_logits, noise, training=False):
        if training:
            return T.spair_render(full.contiguous(), bg.contiguous(), z_depth.contiguous(), z_pres.contiguous(),
                                  noise.normal("render", tuple(full.shape[:-1]) + (self.num_channel,), 0.01))
        return torch.ops.split_vae.spair_render_fwd(full.contiguous(), bg.contiguous(), z_depth.contiguous(), None,
                                                    z_pres_logits.contiguous(), None, False)


class _Model:
    def _finish(self, store, device, seed, dtype="f32"):
        self.store = store
        store.finalize(device, seed)
        # dtype 'bf16': the spatial convolutions run on the bf16 MFMA kernels (fp32 accumulation, fp32 master weights, fp32
        # activations between layers); everything else stays fp32.  'f32' (default) is the reference's precision.
        self.dtype = dtype
        for c in store.convs:
            c.compute_dtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.device = torch.device(device)
        self.seed = seed
        self.generator = torch.Generator(device=device).manual_seed(seed + 1)
        self._native = {}                        # (batch, training, baked config fields) -> spair_native.NativeStep
        self.noise_step = 0                      # Philox step of the native NOISE nodes: one stream position per step of the MODEL

    # config fields a recorded tape bakes in (ZPRES / e.tau, the report matrix of split_z_l / concat_*, the conv dtype ...): part of the cache key,
    # so a changed or different config object records a new tape instead of silently replaying the old values
    _TAPE_FIELDS = ("tau", "dtype", "split_z_l", "concat_z_what", "concat_z_bg", "concat_backbone", "dense_local", "dense_bg", "model",
                    "latent_size", "bg_latent_size", "local_latent_size", "patch_size", "bg_model")

    def native(self, B, config, training=True):
        """The model recorded as a native launch sequence for batch B (spair_native.NativeStep), built once per (batch, mode, baked config)."""
        key = (B, bool(training)) + tuple(repr(config.get(k) if hasattr(config, "get") else getattr(config, k, None)) for k in self._TAPE_FIELDS)
        if key not in self._native:
            from .spair_native import NativeStep
            self._native[key] = NativeStep(self, config, B, training)
        return self._native[key]

    @property
    def trainable_variables(self):
        return [(n, v) for (n, _), v in zip(self.store.spec, self.store.vars)]

    def count_params(self):
        return int(self.store.n_params)

    def set_weights(self, weights):
        """{name: array-like} (any subset), e.g. another implementation's variables by name."""
        with torch.no_grad():
            for (n, shp), v in zip(self.store.spec, self.store.vars):
                if n in weights:
                    v.copy_(torch.as_tensor(weights[n], dtype=torch.float32).reshape(shp))

    def get_weights(self):
        return {n: v.detach().cpu().numpy().copy() for (n, _), v in zip(self.store.spec, self.store.vars)}

    def keras_h5_layers(self):
        """[(top-level layer, [weight names])]: the variables grouped by the Keras sub-model they belong to (spair/spair.py: encoder,
        decoder, bg_model | bg_encoder, bg_decoder, x_hat_encoder, x_hat_decoder), in variable order."""
        groups = []
        for n, _ in self.store.spec:
            top = n.split("/")[0]
            wn = (n + ":0") if n.endswith("/bias") else (n + "/kernel:0")
            if groups and groups[-1][0] == top:
                groups[-1][1].append(wn)
            else:
                groups.append((top, [wn]))
        return groups

    def save_weights(self, path):
        """spair/trainer.py:424 model.save_weights('models/<run>.h5'): a Keras HDF5 weights file (h5io.py: layer_names / weight_names
        attribute layout) when the path ends in .h5 / .hdf5 / .keras, else an .npz keyed by the variable names.  Keras layouts either
        way (conv HWIO, dense [in,out]).  Returns the path written."""
        path = str(path)
        if path.endswith((".h5", ".hdf5", ".keras")):
            from . import h5io
            it = iter(v.detach().cpu().numpy() for v in self.store.vars)
            return h5io.save_keras_weights(path, [(ln, [(wn, next(it)) for wn in wns]) for ln, wns in self.keras_h5_layers()])
        path = path if path.endswith(".npz") else path + ".npz"
        np.savez(path, **self.get_weights())
        return path

    def load_weights(self, path):
        """Inverse of save_weights; HDF5 files are read BY ORDER like Keras' load_weights_from_hdf5_group, shapes checked."""
        path = str(path)
        if path.endswith((".h5", ".hdf5", ".keras")):
            from . import h5io
            arrs = [a for _, ws in h5io.load_keras_weights(path) for _, a in ws]
            if [tuple(a.shape) for a in arrs] != [shp for _, shp in self.store.spec]:
                raise ValueError("weights file does not match this model's variables (count / shapes, by order)")
            with torch.no_grad():
                for v, a in zip(self.store.vars, arrs):
                    v.copy_(torch.from_numpy(a))
            return
        with np.load(path) as f:
            self.set_weights({k: f[k] for k in f.files})

    def summary(self):
        for n, s in self.store.spec:
            print(f"{n:48s} {s}")
        print("Total params:", self.count_params())


class SPAIR(_Model):
    """spair/spair.py:19-49: model 'spair' | 'bg_spair'."""

    def __init__(self, config, device="cuda", seed=0):
        image_size, C = list(config.image_size), config.image_size[2]
        self.model, self.image_size = config.model, image_size
        store = VarStore()
        self.encoder = Encoder(store, config.object_size, config.latent_size, config.tau, C)
        self.decoder = Decoder(store, image_size, config.object_size, config.latent_size, config.latent_size)
        self.bg_model = BackgroundModel(store, image_size, config.bg_latent_size) if config.model == "bg_spair" else None
        self.bg_latent_size = config.bg_latent_size
        self.renderer = Renderer(C)
        self._finish(store, device, seed, config.dtype or "f32")

    def __call__(self, inputs, training=False, noise=None):
        nz = _Noise(noise, self.device, self.generator)
        enc = self.encoder(inputs, nz, training)
        z_what, z_where, z_depth, z_pres, z_pres_logits = enc[0], enc[3], enc[6], enc[9], enc[10]
        rgb, alpha, full, bbox = self.decoder(z_what, z_where)
        if self.bg_model is not None:
            bg, z_bg, z_bg_mean, z_bg_sig = self.bg_model(inputs, nz.normal("eps_bg", (inputs.shape[0], self.bg_latent_size)))
            x_recon = self.renderer(full, bg, z_depth, z_pres, z_pres_logits, nz, training)
            return (x_recon, *enc, rgb, alpha, full, bbox, z_bg, z_bg_mean, z_bg_sig)
        x_recon = self.renderer(full, torch.zeros_like(inputs), z_depth, z_pres, z_pres_logits, nz, training)    # bg_recon = 0.0 (:39)
        return (x_recon, *enc, rgb, alpha, full, bbox)


class LGSPAIR(_Model):
    """spair/spair.py:52-106: SPLIT-SPAIR.  inputs [B,H,W,6] = x | scrambled x (augmentation.py)."""

    def __init__(self, config, device="cuda", seed=0):
        image_size, C = list(config.image_size), config.image_size[2]
        L, Ll, Lbg = config.latent_size, config.local_latent_size, config.bg_latent_size
        self.image_size = image_size
        self.concat_z_what, self.concat_backbone, self.concat_z_bg = bool(config.concat_z_what), bool(config.concat_backbone), bool(config.concat_z_bg)
        self.Ll, self.Lbg = Ll, Lbg
        store = VarStore()
        self.encoder = Encoder(store, config.object_size, L, config.tau, C, concat=self.concat_backbone, local_latent_size=Ll)
        self.decoder = Decoder(store, image_size, config.object_size, L, L + (Ll if self.concat_z_what else 0))
        self.renderer = Renderer(C)
        self.bg_encoder = ImageEncoder(store, image_size, Lbg, "bg_encoder", bool(config.dense_bg))
        self.bg_decoder = ImageDecoder(store, image_size, Lbg + (Ll if self.concat_z_bg else 0), "bg_decoder", bool(config.dense_bg))
        self.x_hat_encoder = ImageEncoder(store, image_size, Ll, "x_hat_encoder", bool(config.dense_local))
        self.x_hat_decoder = ImageDecoder(store, image_size, Ll, "x_hat_decoder", bool(config.dense_local))
        self._finish(store, device, seed, config.dtype or "f32")

    def __call__(self, inputs, training=False, noise=None):
        nz = _Noise(noise, self.device, self.generator)
        B = inputs.shape[0]
        x, x_hat = inputs[..., :3].contiguous(), inputs[..., 3:].contiguous()
        z_l, z_l_mean, z_l_sig = self.x_hat_encoder(x_hat, nz.normal("eps_l", (B, self.Ll)))
        z_bg, z_bg_mean, z_bg_sig = self.bg_encoder(x, nz.normal("eps_bg", (B, self.Lbg)))
        enc = list(self.encoder([x, z_l] if self.concat_backbone else x, nz, training))
        x_hat_recon = self.x_hat_decoder(z_l)
        if self.concat_z_bg:
            z_bg = torch.cat([z_bg, z_l], dim=-1)
        bg_recon = self.bg_decoder(z_bg)
        if self.concat_z_what:
            enc[0] = torch.cat([enc[0], z_l[:, None, None, :].expand(-1, 4, 4, -1)], dim=-1)
        z_what, z_where, z_depth, z_pres, z_pres_logits = enc[0], enc[3], enc[6], enc[9], enc[10]
        rgb, alpha, full, bbox = self.decoder(z_what, z_where)
        x_recon = self.renderer(full, bg_recon, z_depth, z_pres, z_pres_logits, nz, training)
        return (x_recon, *enc, rgb, alpha, full, bbox, z_bg, z_bg_mean, z_bg_sig, x_hat_recon, z_l, z_l_mean, z_l_sig)


def get_model(config, device="cuda", seed=0):
    """spair/spair.py:8-17."""
    config = config if isinstance(config, dotdict) else dotdict(config)
    if config.model == "lg_spair":
        return LGSPAIR(config, device, seed)
    if config.model in ("spair", "bg_spair"):
        return SPAIR(config, device, seed)
    raise NotImplementedError("Model type not implemented")          # incl. 'lg_glimpse_spair': LGGlimpseSPAIR is undefined upstream
