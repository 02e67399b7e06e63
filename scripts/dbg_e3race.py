import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from split_vae_amd import data, ops
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
H, B = 32, 16
m = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="bf16", device=torch.device("cuda"), seed=11); m.beta = 40.0
x = data.synthetic_images(B, H, H, seed=0, device="cuda")
images = Augmentator("scramble", size=4, seed=1).augment(x)
plan = m.plan(B)
outs = []
for rep in range(3):
    G = torch.zeros_like(m.flat)
    plan.step(PHASE_ALL & ~PHASE_ADAM, params=m.flat, grads=G, images6=images, seed=1, step=1)
    torch.cuda.synchronize()
    outs.append(G.cpu().numpy())
print("self-consistent:", [bool(np.array_equal(outs[0], o)) for o in outs[1:]])
np.save(sys.argv[1], outs[0])
