"""One eager train step at the bench configuration (for SV_TC_VERBOSE=1 / SV_WT_VERBOSE=1 plan dumps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=torch.device("cuda"), seed=3)
model.beta = 120.0
x = data.synthetic_images(B, 64, 64, seed=0, device="cuda")
trainer.train_step(model, Augmentator("scramble", size=8, seed=1).augment(x), Adam(learning_rate=1e-4))
torch.cuda.synchronize()
