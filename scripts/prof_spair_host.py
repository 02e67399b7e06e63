"""Host-side profile (cProfile) of the SPLIT-SPAIR train step: where the Python / dispatch time of the ~700 launches goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import spair, spair_main, spair_trainer
from split_vae_amd.augmentation import Augmentator
cfg = spair_main.default_config(model="lg_spair", latent_size=64, bg_latent_size=4, local_latent_size=4, patch_size=8, split_z_l=True,
                                concat_z_what=True, dense_local=True, dense_bg=True)
model = spair.get_model(cfg, seed=0)
x, _ = spair_main.synthetic_canvases(32, seed=1)
images = Augmentator("scramble", size=8, seed=2).augment(x)
opt = spair_trainer.ClipnormAdam(1e-4)
for i in range(5):
    spair_trainer.train_step(model, images, opt, i, cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    spair_trainer.train_step(model, images, opt, 5 + i, cfg)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
