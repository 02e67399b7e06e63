"""Per-variable gradient error of the SPAIR step vs the fp64 oracle (debug aid of tests/test_gpu_spair_model.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import spair_model_ref as R
from split_vae_amd import spair, spair_trainer
from split_vae_amd.utils import dotdict
sys.path.insert(0, "tests")
from test_gpu_spair_model import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "spair"
dt = torch.float64 if len(sys.argv) < 3 else torch.float32
cfg = R.default_config(**CONFIGS[name])
B, step = 3, 41
chans = 6 if cfg.model == "lg_spair" else 3
images = torch.rand(B, 48, 48, chans, generator=torch.Generator().manual_seed(11))
p = R.init_params(cfg, seed=5, dtype=dt)
noise = R.draw_noise(cfg, B, seed=7, dtype=dt)
for v in p.values():
    v.requires_grad_(True)
ref = R.forward(p, cfg, images.to(dt), noise, training=True)
total_ref, _ = R.losses(cfg, images.to(dt), ref, step)
gref = torch.autograd.grad(total_ref, list(p.values()), allow_unused=True)
model = spair.get_model(dotdict(cfg), seed=0)
model.set_weights({k: v.detach().numpy() for k, v in p.items()})
dn = {k: v.float().cuda() for k, v in noise.items()}
opt = spair_trainer.ClipnormAdam(1e-3)
res, losses, total, grads = spair_trainer.train_step(model, images.cuda(), opt, step, dotdict(cfg), noise=dn, return_grads=True)
for (n, _), ga, gb in zip(model.trainable_variables, grads, gref):
    e = float((ga.double().cpu() - gb.double()).norm() / gb.double().norm().clamp_min(1e-12))
    print(f"{n:44s} {e:.3e}  |g|={float(gb.norm()):.3e}")
