"""Cost of issuing the step as the data-parallel trainer does (four phase groups) vs one call, on one GPU without
communication.  Usage: python scripts/bench_split.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data, trainer
from split_vae_amd._lib import (PHASE_ALL, PHASE_PREP, PHASE_FORWARD, PHASE_LOSS, PHASE_BWD_DECODERS, PHASE_BWD_ENC_HEADS,
                                PHASE_BWD_ENC_CONVS, PHASE_ADAM)
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae

B, H = 512, 64
model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="bf16", device=torch.device("cuda"), seed=3)
model.beta = 120.0
plan = model.plan(B)
x = data.synthetic_images(B, H, H, seed=0, device="cuda")
aug = Augmentator("scramble", size=8, seed=1)
P, G = model.flat, model.grad_flat
M, V = torch.zeros_like(P), torch.zeros_like(P)
groups = [PHASE_PREP | PHASE_FORWARD | PHASE_LOSS | PHASE_BWD_DECODERS, PHASE_BWD_ENC_HEADS, PHASE_BWD_ENC_CONVS, PHASE_ADAM]


def run(split, steps=60, warmup=8):
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        img = aug.augment(x)
        kw = dict(params=P, grads=G, adam_m=M, adam_v=V, images6=img, seed=3, step=it, lr=1e-4, t=it + 1)
        for ph in (groups if split else [PHASE_ALL]):
            plan.step(ph, **kw)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


for rep in range(3):
    print("one call %9.0f images/s   four phase groups %9.0f images/s" % (run(False), run(True)), flush=True)
