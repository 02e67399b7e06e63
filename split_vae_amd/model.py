"""Host-side mirror of vae/model.py's SPLIT-VAE surface: LGVae (vae/model.py:174-218).

Same constructor arguments, call/encode/decode signatures, 10-tuple order and trainable-variable
order as the reference; all arithmetic runs in libsplitvae_hip.so through a native step plan
(one per batch size).  Parameters live in ONE flat fp32 device buffer in Keras creation order and
Keras layouts (conv HWIO, dense [in,out], flatten order h,w,c), so `trainable_variables` are
zero-copy views and `save_weights` emits the 40 arrays a TF user would expect.
"""
import math

import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import PHASE_FWD_DECODERS, PHASE_INFER, PHASE_PREP


class LGVae:
    def __init__(self, global_latent_dims, local_latent_dims, image_shape=None, variational=True, type='conv',
                 dtype='bf16', device=None, seed=0):
        if not variational:
            raise NotImplementedError('Determiistic LG-AE not implemented')   # vae/model.py:202
        if type != 'conv':
            raise NotImplementedError("only the 'conv' encoder is ever instantiated (vae/model.py:182-183)")
        if not torch.cuda.is_available():
            raise _lib.SplitVaeError("split_vae_amd needs a HIP device (MI355X); there is no CPU path")
        self.global_latent_dims = global_latent_dims
        self.local_latent_dims = local_latent_dims
        self.variational = variational
        self.image_shape = image_shape
        self.H, self.W = int(image_shape[1]), int(image_shape[2])
        self.dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32}.get(dtype, dtype)
        self.device = torch.device(device or "cuda")
        self.seed = seed
        self._calls = 0
        self._plans = {}
        self.beta = 1.0     # KL weight (vae/main.py:19); set by the trainer from config.beta
        desc = _lib.LGVaeDesc(1, self.H, self.W, global_latent_dims, local_latent_dims, ops.sv_dtype(self.dtype), 1.0)
        self.param_table = ops.param_table(desc)
        import ctypes as C
        self.n_params = _lib.load().sv_lgvae_param_count(C.byref(desc))
        if self.n_params < 0:
            raise _lib.SplitVaeError("unsupported LGVae geometry H=%d W=%d latents=%d/%d" %
                                     (self.H, self.W, global_latent_dims, local_latent_dims))
        self.flat = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grad_flat = torch.zeros_like(self.flat)
        self._init_glorot(seed)

    # ---------------------------------------------------------------- variables
    def _init_glorot(self, seed):
        """Keras defaults [TF-2.0 semantics]: glorot_uniform kernels, zero biases."""
        rng = np.random.Generator(np.random.PCG64(seed))
        host = np.zeros(self.n_params, np.float32)
        for name, off, shape in self.param_table:
            if name.endswith("bias"):
                continue
            if len(shape) == 4:
                rf = shape[0] * shape[1]
                fan_in, fan_out = rf * shape[2], rf * shape[3]
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            n = int(np.prod(shape))
            host[off:off + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        self.flat.copy_(torch.from_numpy(host))

    def _views(self, flat):
        return [flat[off:off + int(np.prod(shape))].view(*shape) for (_, off, shape) in self.param_table]

    @property
    def trainable_variables(self):
        """40 tensors, creation order: encoder_x{e1,e2,e3,e4_mean,e4_sd}, encoder_x_hat{..},
        decoder_x{d1..d5}, decoder_x_hat{..}; kernel then bias."""
        return self._views(self.flat)

    @property
    def gradients(self):
        return self._views(self.grad_flat)

    def set_weights(self, arrays):
        for v, a in zip(self.trainable_variables, arrays):
            v.copy_(torch.as_tensor(np.asarray(a, np.float32)).to(self.device))

    def get_weights(self):
        return [v.cpu().numpy() for v in self.trainable_variables]

    def keras_names(self):
        # vae/model.py layer attribute names under the model: <sublayer>/<attr>/<kernel|bias>:0
        return [n + ":0" for (n, _, _) in self.param_table]

    # ---------------------------------------------------------------- checkpoints (vae/trainer.py:421)
    KERAS_MODEL_NAME = "lg_vae"             # keras to_snake_case(class name): the outer name scope of every variable

    def _keras_kind(self, name):
        """Keras class of the layer that owns variable `name` (<sublayer>/<attr>/<kernel|bias>) -> (class uid stem,
        explicit layer name or None, enclosing Sequential attr or None)."""
        attr = name.split("/")[1]
        return ("dense" if attr in ("e4_mean", "e4_sd", "d1") else "conv2d"), None, None

    def keras_h5_layers(self):
        """[(layer name, [weight name, ...]), ...] as Keras names them for model.save_weights(*.h5) [TF-2.0 semantics:
        global per-class uid counters in creation order (conv2d, conv2d_1, ..., dense, dense_1, ...), variables scoped
        <model>/<layer>/<inner layer>/<kernel|bias>:0; sub-models are named encoder, encoder_1, decoder, decoder_1].
        Aligned with trainable_variables order.  Only cosmetic for a round trip: loading goes by order."""
        uid, seq_uid, layers, sub_names, sub_uid, inner = {}, {}, [], {}, {}, {}
        names = [n[:-2] if n.endswith(":0") else n for n in self.keras_names()]
        for n in names:
            sub, attr, kind = n.split("/")[0], n.split("/")[1:-1], n.split("/")[-1]
            if sub not in sub_names:
                stem = "encoder" if sub.startswith("encoder") else "decoder"
                k = sub_uid.get(stem, 0)
                sub_uid[stem] = k + 1
                sub_names[sub] = stem if k == 0 else "%s_%d" % (stem, k)
                layers.append((sub_names[sub], []))
            key = (sub, tuple(attr))
            if key not in inner:
                cls, explicit, seq = self._keras_kind(n)
                if explicit:
                    lname = explicit
                else:
                    k = uid.get(cls, 0)
                    uid[cls] = k + 1
                    lname = cls if k == 0 else "%s_%d" % (cls, k)
                if seq:
                    if (sub, seq) not in seq_uid:
                        k = len(seq_uid)
                        seq_uid[(sub, seq)] = "sequential" if k == 0 else "sequential_%d" % k
                    lname = seq_uid[(sub, seq)] + "/" + lname
                inner[key] = lname
            layers[-1][1].append("%s/%s/%s/%s:0" % (self.KERAS_MODEL_NAME, sub_names[sub], inner[key], kind))
        return layers

    def save_weights(self, path):
        """vae/trainer.py:421 model.save_weights('models/<run>.h5'): a Keras HDF5 weights file (h5io.py: the layer_names /
        weight_names attribute layout of keras/saving/hdf5_format.py) when the path ends in .h5 / .hdf5 / .keras;
        otherwise an .npz with the same arrays under <sublayer>/<attr>/<kernel|bias>:0 names.  Keras layouts either way
        (conv HWIO, dense [in,out], flatten order h,w,c).  Returns the path written."""
        path = str(path)
        ws = self.get_weights()
        if path.endswith((".h5", ".hdf5", ".keras")):
            from . import h5io
            it = iter(ws)
            return h5io.save_keras_weights(path, [(ln, [(wn, next(it)) for wn in wns]) for ln, wns in self.keras_h5_layers()])
        path = path if path.endswith(".npz") else path + ".npz"
        np.savez(path, **{n: w for n, w in zip(self.keras_names(), ws)})
        return path

    def load_weights(self, path):
        """Inverse of save_weights; HDF5 files are read BY ORDER like Keras' load_weights_from_hdf5_group (so a file the
        reference wrote loads whatever uids its layer names carry), shapes checked."""
        path = str(path)
        if path.endswith((".h5", ".hdf5", ".keras")):
            from . import h5io
            arrs = [a for _, ws in h5io.load_keras_weights(path) for _, a in ws]
            want = [tuple(v.shape) for v in self.trainable_variables]
            if [tuple(a.shape) for a in arrs] != want:
                raise ValueError("weight file %s holds %d arrays with shapes %s; this model expects %d with shapes %s" %
                                 (path, len(arrs), [tuple(a.shape) for a in arrs][:4], len(want), want[:4]))
            self.set_weights(arrs)
            return
        z = np.load(path if path.endswith(".npz") else path + ".npz")
        self.set_weights([z[n] for n in self.keras_names()])

    def summary(self):
        total = sum(int(np.prod(s)) for (_, _, s) in self.param_table)
        for n, _, s in self.param_table:
            print("%-32s %s" % (n, tuple(s)))
        print("Total params: {:,}".format(total))

    # ---------------------------------------------------------------- plans
    def plan(self, B, beta=None):
        beta = self.beta if beta is None else beta
        key = (int(B), float(beta))
        if key not in self._plans:
            self._plans[key] = ops.LGVaePlan(B, self.H, self.W, self.global_latent_dims, self.local_latent_dims,
                                             beta=beta, dtype=self.dtype, device=self.device)
            if os.environ.get("SV_GRAPH", "0") == "1":      # opt-in hipGraph replay of repeated steps (non-default streams only;
                                                            # measured: no gain, the step is GPU-bound at every batch size)
                self._plans[key].graph_enable(True)
        return self._plans[key]

    def _outputs(self, plan, B, copy):
        L_g, L_l = self.global_latent_dims, self.local_latent_dims
        o6x = plan.buffer("out6_x", torch.float32, (B, self.H, self.W, 6))
        o6h = plan.buffer("out6_xh", torch.float32, (B, self.H, self.W, 6))
        outs = (o6x[..., :3], o6x[..., 3:],
                plan.buffer("z_x", torch.float32, (B, L_g)), plan.buffer("z_mean_x", torch.float32, (B, L_g)),
                plan.buffer("z_sig_x", torch.float32, (B, L_g)), plan.buffer("z_xh", torch.float32, (B, L_l)),
                o6h[..., :3], o6h[..., 3:],
                plan.buffer("z_mean_xh", torch.float32, (B, L_l)), plan.buffer("z_sig_xh", torch.float32, (B, L_l)))
        return tuple(t.clone() for t in outs) if copy else outs

    # ---------------------------------------------------------------- reference surface
    def __call__(self, inputs, training=False, eps=None, copy=True):
        """LGVae.call (vae/model.py:189-200): inputs[B,H,W,6] fp32 -> (x_mean, x_log_scale, z_x,
        z_mean_x, z_sig_x, z_x_hat, x_hat_mean, x_hat_log_scale, z_mean_x_hat, z_sig_x_hat).
        eps=(eps_x, eps_x_hat) pins the Sampling noise (vae/model.py:12 draws it unseeded)."""
        B = inputs.shape[0]
        plan = self.plan(B)
        ex, eh = (None, None) if eps is None else eps
        plan.step(PHASE_INFER, params=self.flat, images6=inputs.contiguous(), eps_x=ex, eps_x_hat=eh,
                  seed=self.seed, step=self._calls)
        self._calls += 1
        return self._outputs(plan, B, copy)

    call = __call__

    def encode(self, inputs, eps=None):
        """vae/model.py:204-209 -> (z_x, z_x_hat), sampled."""
        out = self(inputs, eps=eps)
        return out[2], out[5]

    def decode(self, z_x, z_x_hat, rescale=True):
        """vae/model.py:211-218: decoder_x(concat[z_x, z_x_hat]), decoder_x_hat(z_x_hat); with
        rescale the means are mapped to [0,1] and log_scale is dropped."""
        B = z_x.shape[0]
        plan = self.plan(B)
        zcat = plan.buffer("zcat", self.dtype, (B, self.global_latent_dims + self.local_latent_dims))
        zcat.copy_(torch.cat([z_x, z_x_hat], dim=1).to(self.dtype))
        plan.step(PHASE_PREP | PHASE_FWD_DECODERS, params=self.flat)
        x_mean = plan.buffer("out6_x", torch.float32, (B, self.H, self.W, 6))[..., :3].clone()
        x_hat_mean = plan.buffer("out6_xh", torch.float32, (B, self.H, self.W, 6))[..., :3].clone()
        if rescale:
            return torch.clamp((x_mean + 1) * 0.5, 0., 1.), torch.clamp((x_hat_mean + 1) * 0.5, 0., 1.)
        return x_mean, x_hat_mean
