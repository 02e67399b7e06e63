"""The data-parallel step on one rank (SV_DIST_FORCE=1), eager launches against hipGraph replay of its phases (SV_GRAPH=1: single-stream captures),
on a created stream.  Usage: python scripts/r04_dp_graph_probe.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SV_DIST_FORCE"] = "1"
import torch
import split_vae_amd
from split_vae_amd import data, trainer, dist as svdist
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
split_vae_amd.configure_hw_queues()
svdist.init_from_env()
dev = torch.device("cuda", 0)


def run(B, use_graph, dp, steps=300, warmup=20):
    os.environ["SV_GRAPH"] = "1" if use_graph else "0"
    m = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=dev, seed=3); m.beta = 120.0
    opt = Adam(learning_rate=1e-4); aug = Augmentator("scramble", size=8, seed=1)
    x = data.synthetic_images(B, 64, 64, seed=0, device=dev)
    red = svdist.make_reducer(m.param_table, m.n_params) if dp else None
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            plan = trainer.train_step(m, aug.augment(x, plan=m.plan(B)), opt, reducer=red, keep_recon=False)
        side.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            plan = trainer.train_step(m, aug.augment(x, plan=m.plan(B)), opt, reducer=red, keep_recon=False)
        side.synchronize(); dt = time.perf_counter() - t0
    return dt / steps * 1e3, plan.graph_count()


for B in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    for dp in (False, True):
        for g in (False, True, False, True):
            ms, n = run(B, g, dp)
            print("B=%4d %-6s %-6s %7.3f ms/step  graphs=%d" % (B, "dp" if dp else "single", "graph" if g else "eager", ms, n), flush=True)
