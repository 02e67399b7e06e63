"""SPLIT-SPAIR / SPAIR train step (config 5: 48x48 canvases, 4x4 cells, batch 32; README.md:93) on one MI355X.
Usage: python scripts/bench_spair.py [batch ...]     (one JSON line per model)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from split_vae_amd import spair, spair_main, spair_trainer
from split_vae_amd.augmentation import Augmentator

MODELS = {
    "lg_spair (README.md:93)": dict(model="lg_spair", latent_size=64, bg_latent_size=4, local_latent_size=4, patch_size=8, z_bg_beta=10.0,
                                    split_z_l=True, concat_z_what=True, dense_local=True, dense_bg=True),
    "bg_spair (README.md:87)": dict(model="bg_spair", latent_size=64, bg_latent_size=4, dense_bg=True, z_bg_beta=10.0),
    "spair (defaults)": dict(model="spair"),
}


def run(name, kw, B, steps=30, warmup=5):
    cfg = spair_main.default_config(dtype=os.environ.get("SPAIR_DTYPE", "f32"), **kw)
    model = spair.get_model(cfg, seed=0)
    x, _ = spair_main.synthetic_canvases(B, seed=1)
    images = Augmentator("scramble", size=cfg.patch_size, seed=2).augment(x) if cfg.model == "lg_spair" else x
    opt = spair_trainer.ClipnormAdam(cfg.learning_rate, clipnorm=1.0)
    graphed = not os.environ.get("SPAIR_EAGER")
    step_fn = spair_trainer.GraphedTrainStep(model, opt, cfg, images) if graphed else (lambda im, st: spair_trainer.train_step(model, im, opt, st, cfg))
    for i in range(warmup):
        step_fn(images, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(images, warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"what": "SPAIR train step (fwd+losses+bwd+clipnorm Adam)", "launch": "hipGraph replay" if graphed else "eager", "model": name, "device": "MI355X", "dtype": cfg.dtype, "batch": B,
                      "params": model.count_params(), "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}), flush=True)


def cpu(name, kw, B, steps=3):
    """The fp32 torch-CPU restatement (oracle/spair_model_ref.py: forward + losses + autograd + clipnorm Adam) on the host cores."""
    from oracle import spair_model_ref as R
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = R.default_config(**kw)
    p = R.init_params(cfg, 0, torch.float32)
    for v in p.values():
        v.requires_grad_(True)
    images = torch.rand(B, 48, 48, 6 if cfg.model == "lg_spair" else 3)
    m = [torch.zeros_like(v) for v in p.values()]
    vv = [torch.zeros_like(v) for v in p.values()]
    ts = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        o = R.forward(p, cfg, images, R.draw_noise(cfg, B, i, torch.float32), training=True)
        total, _ = R.losses(cfg, images, o, i)
        g = torch.autograd.grad(total, list(p.values()), allow_unused=True)
        with torch.no_grad():
            R.clipnorm_adam_(list(p.values()), [x if x is not None else torch.zeros_like(v) for x, v in zip(g, p.values())], m, vv, i + 1)
        ts.append(time.perf_counter() - t0)
    dt = min(ts[1:])
    print(json.dumps({"what": "SPAIR train step, torch-CPU restatement (oracle)", "model": name, "device": "host CPU", "cores": cores, "dtype": "f32",
                      "batch": B, "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:]] or [32]:
        for name, kw in MODELS.items():
            if os.environ.get("SPAIR_ONLY", "") in name:
                run(name, kw, B)
    if os.environ.get("SPAIR_CPU"):
        for name, kw in MODELS.items():
            if os.environ.get("SPAIR_ONLY", "") in name:
                cpu(name, kw, 32)
