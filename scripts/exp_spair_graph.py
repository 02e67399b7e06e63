"""Feasibility: the SPLIT-SPAIR train step captured into one hipGraph (torch.cuda.graph), constants frozen."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import spair, spair_main, spair_trainer
from split_vae_amd.augmentation import Augmentator
cfg = spair_main.default_config(model="lg_spair", latent_size=64, bg_latent_size=4, local_latent_size=4, patch_size=8, split_z_l=True,
                                concat_z_what=True, dense_local=True, dense_bg=True)
model = spair.get_model(cfg, seed=0)
model.generator = None
x, _ = spair_main.synthetic_canvases(32, seed=1)
images = Augmentator("scramble", size=8, seed=2).augment(x)
opt = spair_trainer.ClipnormAdam(1e-4)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        spair_trainer.train_step(model, images, opt, i, cfg)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res, losses = spair_trainer.train_step(model, images, opt, 3, cfg)
torch.cuda.synchronize()
print("captured")
for k in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(30):
    g.replay()
torch.cuda.synchronize()
print("replay ms/step", (time.perf_counter() - t0) / 30 * 1e3, [float(l) for l in losses][:3])
