"""Eager launches vs hipGraph replay of the full train step (sv_lgvae_graph_enable) on the launch-bound
configurations.  Usage: python scripts/bench_graph.py [size batch]...   (default: 32 64, 64 64, 64 512)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam


def run(H, B, use_graph, steps=200, warmup=10):
    os.environ["SV_GRAPH"] = "1" if use_graph else "0"
    beta, patch = (120.0, 8) if H == 64 else (40.0, 1)
    model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="bf16", device=torch.device("cuda"), seed=3)
    model.beta = beta
    opt = Adam(learning_rate=1e-4)
    aug = Augmentator("scramble", size=patch, seed=1)
    x = data.synthetic_images(B, H, H, seed=0, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            plan = trainer.train_step(model, aug.augment(x), opt)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            plan = trainer.train_step(model, aug.augment(x), opt)
        side.synchronize()
        dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3, plan.graph_count()


if __name__ == "__main__":
    cfgs = [(int(a), int(b)) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(32, 64), (64, 64), (64, 512)]
    for H, B in cfgs:
        for g in (False, True, False, True):
            ips, ms, n = run(H, B, g)
            print("size %d batch %4d  %-6s %10.0f images/s  %7.3f ms/step  graphs=%d" % (H, B, "graph" if g else "eager", ips, ms, n), flush=True)
