"""SPLIT-SPAIR (README.md:107's lg_spair model = BASELINE config 5, Multi-Bird-Hard flags; SPAIR_FLAGS=easy: README.md:93; batch 32) train step through the native launch sequence (spair_native.NativeStep):
ms per step and images/s; SPAIR_PROFILE=1 runs fewer steps (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from split_vae_amd import spair, spair_main, spair_trainer
from split_vae_amd.augmentation import Augmentator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
FLAGS = {"hard": dict(latent_size=64, bg_latent_size=64, local_latent_size=64, patch_size=8, z_bg_beta=1.0, z_what_beta=0.5),
         "easy": dict(latent_size=64, bg_latent_size=4, local_latent_size=4, patch_size=8, z_bg_beta=10.0)}[os.environ.get("SPAIR_FLAGS", "hard")]
cfg = spair_main.default_config(dtype=DT, model="lg_spair", split_z_l=True, concat_z_what=True, dense_local=True, dense_bg=True, **FLAGS)
model = spair.get_model(cfg, seed=0)
x, _ = spair_main.synthetic_canvases(B, seed=1)
images = Augmentator("scramble", size=cfg.patch_size, seed=2).augment(x)
opt = spair_trainer.ClipnormAdam(cfg.learning_rate)
steps = 20 if os.environ.get("SPAIR_PROFILE") else 200
for i in range(10):
    spair_trainer.train_step(model, images, opt, i, cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    spair_trainer.train_step(model, images, opt, 10 + i, cfg)
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / steps
print({"batch": B, "dtype": DT, "ms_per_step": round(1e3 * t, 4), "host_ms_per_step": round(1e3 * t_host, 4), "images_per_s": round(B / t, 1),
       "tape_nodes": model.native(B, cfg).n_nodes})
