import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
B = int(sys.argv[2])
dev = torch.device("cuda")
model = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=dev, seed=3); model.beta = 120.0
opt = Adam(learning_rate=1e-4); aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(B, 64, 64, seed=100, device=dev)
plan = trainer.train_step(model, aug.augment(x), opt)
torch.cuda.synchronize()
out = {}
for name, shp in (("gu4_", (B, 64, 64, 32)), ("g4_", (B, 32, 32, 32)), ("gu3_", (B, 32, 32, 64)), ("g3_", (B, 16, 16, 64)), ("gu2_", (B, 16, 16, 128)), ("g2_", (B, 8, 8, 128)), ("g5_", (B, 64, 64, 8))):
    for sfx in ("x", "xh"):
        out[name + sfx] = plan.buffer(name + sfx, torch.bfloat16, shp).float().cpu().numpy()
np.savez(sys.argv[1], **out)
