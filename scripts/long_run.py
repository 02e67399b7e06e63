"""Stability run: N training steps of SPLIT-VAE (CelebA-64 shapes, eight synthetic batches in turn) -- loss terms every 500 steps.  usage: long_run.py [N] [bf16|f32] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
DT = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype=DT, device=torch.device("cuda"), seed=3)
m.beta = 120.0
opt = Adam(learning_rate=1e-4)
aug = Augmentator("scramble", size=8, seed=1)
xs = [data.synthetic_images(B, 64, 64, seed=s, device="cuda") for s in range(8)]
t0 = time.time()
for i in range(N):
    plan = trainer.train_step(m, aug.augment(xs[i % 8], plan=m.plan(B)), opt, keep_recon=False)
    if i % 500 == 0 or i == N - 1:
        l = trainer.last_losses(plan)
        print(i, {k: round(v, 2) for k, v in l.items()}, "finite params:", bool(torch.isfinite(m.flat).all()), flush=True)
print("%s B=%d: %d steps in %.1f s (%.3f ms per step incl. the augmentation)" % (DT, B, N, time.time() - t0, 1e3 * (time.time() - t0) / N))
