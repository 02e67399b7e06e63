"""Probe of tests/test_gpu_spair_model.py::test_lanes_compute_the_single_stream_step: a few native SPLIT-SPAIR train steps (lg_spair, Multi-Bird-Hard flags, batch 8);
prints every step's reported losses and, at the end, the parameters' and the last gradients' norms + the tape's lane count.
SV_TAPE_LANES=0 in the environment: everything on the caller's stream.  usage: python tests/spair_lanes_probe.py <out.npz> [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from split_vae_amd import spair, spair_main, spair_trainer        # noqa: E402
from split_vae_amd.augmentation import Augmentator                  # noqa: E402

out, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = 8
cfg = spair_main.default_config(dtype="f32", model="lg_spair", split_z_l=True, concat_z_what=True, dense_local=True, dense_bg=True, latent_size=64,
                                bg_latent_size=64, local_latent_size=64, patch_size=8, z_bg_beta=1.0, z_what_beta=0.5)
model = spair.get_model(cfg, seed=0)
x, _ = spair_main.synthetic_canvases(B, seed=1)
images = Augmentator("scramble", size=cfg.patch_size, seed=2).augment(x)
opt = spair_trainer.ClipnormAdam(cfg.learning_rate)
hist = []
for i in range(steps):
    _, losses = spair_trainer.train_step(model, images, opt, i, cfg)
    torch.cuda.synchronize()
    hist.append(np.array([float(v) for v in losses], dtype=np.float64))
ns = model.native(B, cfg)
np.savez(out, losses=np.stack(hist), params=model.store.flat.cpu().numpy(), grads=ns.grads.cpu().numpy())
print("SPAIR_LANES_PROBE ok", steps)
