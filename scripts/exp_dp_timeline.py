"""The data-parallel step on one rank (SV_DIST_FORCE) for a kernel-trace timeline: 30 steps with the bucketed all-reduce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SV_DIST_FORCE"] = "1"
import torch
from split_vae_amd import data, dist as svdist, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
svdist.init_from_env()
model = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=torch.device("cuda"), seed=3)
model.beta = 120.0
opt = Adam(learning_rate=1e-4)
aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(512, 64, 64, seed=0, device="cuda")
red = svdist.make_reducer(model.param_table, model.n_params)
for _ in range(30):
    trainer.train_step(model, aug.augment(x), opt, reducer=red, keep_recon=False)
torch.cuda.synchronize()
torch.distributed.destroy_process_group()
