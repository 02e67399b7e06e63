"""Eager launches vs hipGraph replay (single-stream capture) vs replay of a capture that FORKS to the weight-gradient side streams (SV_GRAPH_SIDE=1):
ms/step and a hash of the parameters after the run (the bf16 step and the default fp32 step are bit-reproducible: equal hashes = the replay computed the same step).
usage: python scripts/r06_graph_side.py <mode: eager|graph|graph_side|early_side> <dtype> <batch> [steps]      (one mode per process: a bad capture must not take the others down)"""
import hashlib
import os
import sys
import time

mode, dtype, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
os.environ["SV_GRAPH"] = "1" if mode in ("graph", "graph_side") else "0"
if mode == "early_side":            # the decoders' weight images + the gradient zero fill on side stream 0 beside the encoders' forward (opt-in)
    os.environ["SV_EARLY_SIDE"] = "1"
if mode == "graph_side":
    os.environ["SV_GRAPH_SIDE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import split_vae_amd  # noqa: E402
split_vae_amd.configure_hw_queues()
import torch  # noqa: E402
from split_vae_amd import data, trainer  # noqa: E402
from split_vae_amd.augmentation import Augmentator  # noqa: E402
from split_vae_amd.model import LGVae  # noqa: E402
from split_vae_amd.optimizer import Adam  # noqa: E402

H = 64
model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dtype, device=torch.device("cuda"), seed=3)
model.beta = 120.0
opt = Adam(learning_rate=1e-4)
aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(B, H, H, seed=0, device="cuda")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(10):
        plan = trainer.train_step(model, aug.augment(x, plan=model.plan(B)), opt, keep_recon=False)
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan = trainer.train_step(model, aug.augment(x, plan=model.plan(B)), opt, keep_recon=False)
    side.synchronize()
    dt = time.perf_counter() - t0
h = hashlib.sha256(model.flat.cpu().numpy().tobytes()).hexdigest()[:16]
losses = plan.buffer("losses", torch.float32, (8,)).cpu().numpy()
print("%-10s %s B=%d  %.4f ms/step  graphs=%d  params %s  loss %.4f" % (mode, dtype, B, dt / steps * 1e3, plan.graph_count(), h, float(losses[0])), flush=True)
