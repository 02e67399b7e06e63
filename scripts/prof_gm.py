import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
from split_vae_amd.optimizer import Adam
B, H = 64, 32
x = data.synthetic_images(B, H, H, seed=0, device="cuda")
aug = Augmentator("scramble", size=4, seed=1)
m = LGGMVae(128, 128, [-1, H, H, 3], 30, 0.4, dtype="bf16", device="cuda", seed=3); m.beta, m.alpha = 40.0, 40.0
opt = Adam(learning_rate=1e-4)
for _ in range(5): train_step_lg_gm_vae(m, aug.augment(x), opt)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): train_step_lg_gm_vae(m, aug.augment(x), opt)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
