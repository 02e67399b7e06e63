"""Where the data-parallel step's overhead comes from on ONE rank: plain step vs the 4-call phase split without any all-reduce vs the
split with the bucketed all-reduce over a world of one (nccl = RCCL, or sv_comm).  usage: python scripts/exp_dp_overhead.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SV_DIST_FORCE"] = "1"
import torch
from split_vae_amd import data, dist as svdist, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
svdist.init_from_env()
H = 64
model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="bf16", device=torch.device("cuda"), seed=3)
model.beta = 120.0
opt = Adam(learning_rate=1e-4)
aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(B, H, H, seed=0, device="cuda")


class NoReduce:
    world, force, grad_scale = 1, True, 1.0
    def launch(self, flat, bucket): pass
    def wait(self): pass


def timed(reducer, steps=100):
    for _ in range(10):
        trainer.train_step(model, aug.augment(x), opt, reducer=reducer, keep_recon=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(model, aug.augment(x), opt, reducer=reducer, keep_recon=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("plain one-call step        %.3f ms" % timed(None))
print("4-call split, no reduce    %.3f ms" % timed(NoReduce()))
red = svdist.make_reducer(model.param_table, model.n_params)
print("buckets:", {k: [(e - b) * 4 for b, e in v] for k, v in red.buckets.items()})
print("4-call split + all-reduce  %.3f ms  (%s)" % (timed(red), type(red).__name__))
red.mode = "single"
print("single all-reduce mode     %.3f ms" % timed(red))
red.mode = "overlap"
# host cost of the launches alone (no synchronisation in between)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    for k in red.buckets:
        red.launch(model.grad_flat, k)
    red.wait()
th = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
tt = (time.perf_counter() - t0) / 50 * 1e3
print("3 bucket launches + wait: host %.3f ms, with device completion %.3f ms" % (th, tt))
class Timed:
    """wraps a reducer: host time spent inside launch() / wait() while the device is busy with the step"""
    def __init__(self, r):
        self.r, self.t_launch, self.t_wait, self.world, self.force = r, 0.0, 0.0, r.world, True
    @property
    def grad_scale(self): return self.r.grad_scale
    def launch(self, flat, bucket):
        t = time.perf_counter(); self.r.launch(flat, bucket); self.t_launch += time.perf_counter() - t
    def wait(self):
        t = time.perf_counter(); self.r.wait(); self.t_wait += time.perf_counter() - t
tr = Timed(red)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    trainer.train_step(model, aug.augment(x), opt, reducer=tr, keep_recon=False)
th = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print("DP step: host enqueue %.3f ms per step, of which launch() %.3f ms, wait() %.3f ms; wall %.3f" % (th, tr.t_launch / 50 * 1e3, tr.t_wait / 50 * 1e3, (time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50):
    trainer.train_step(model, aug.augment(x), opt, reducer=None, keep_recon=False)
th = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print("plain step: host enqueue time %.3f ms per step" % th)
print("plain again                %.3f ms" % timed(None))
torch.distributed.destroy_process_group()
