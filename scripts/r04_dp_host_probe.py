"""Host-side cost of the data-parallel step's parts on one rank (SV_DIST_FORCE=1): seconds of CPU time per call, GPU running asynchronously."""
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
rank, local_rank, world = svdist.init_from_env()
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=dev, seed=3); m.beta = 120.0
opt = Adam(learning_rate=1e-4); aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(B, 64, 64, seed=0, device=dev)
red = svdist.make_reducer(m.param_table, m.n_params)
print("backend", os.environ.get("SV_DIST_BACKEND"), type(red).__name__, {k: len(v) for k, v in red.buckets.items()})
T = {}
def timed(name, fn):
    t0 = time.perf_counter(); r = fn(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
orig_launch, orig_wait = red.launch, red.wait
red.launch = lambda flat, b: timed("reducer.launch(" + b + ")", lambda: orig_launch(flat, b))
red.wait = lambda: timed("reducer.wait", orig_wait)
plan = m.plan(B)
orig_step = plan.step
def step(ph, **kw): return timed("plan.step(%d)" % ph, lambda: orig_step(ph, **kw))
plan.step = step
for it in range(60):
    if it == 10:
        torch.cuda.synchronize(); T.clear(); t_all = time.perf_counter()
    img = timed("augment", lambda: aug.augment(x, plan=plan))
    timed("train_step total", lambda: trainer.train_step(m, img, opt, reducer=red, keep_recon=False))
torch.cuda.synchronize()
el = time.perf_counter() - t_all
print("B=%d: %.1f us per step wall" % (B, el / 50 * 1e6))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-28s %7.1f us per step (host)" % (k, v / 50 * 1e6))
