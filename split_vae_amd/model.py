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

    def save_weights(self, path):
        """vae/trainer.py:421 (Keras HDF5 there; h5py is unavailable, so an .npz with the same 40
        arrays under Keras-style names in Keras layouts)."""
        arrs = {n: w for n, w in zip(self.keras_names(), self.get_weights())}
        np.savez(path if str(path).endswith(".npz") else str(path) + ".npz", **arrs)

    def load_weights(self, path):
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
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
