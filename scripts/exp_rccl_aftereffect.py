"""Does a (1-rank) RCCL all-reduce slow the kernels that follow it?  Times an HBM-bound copy (128 MB) repeatedly after one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SV_DIST_FORCE"] = "1"
import torch
import torch.distributed as dist
from split_vae_amd import dist as svdist
svdist.init_from_env()
a = torch.empty(32 << 20, device="cuda")
b = torch.empty_like(a)
g = torch.zeros(8 << 20, device="cuda")


def copies(n=12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        b.copy_(a)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [round(ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(n)]


for _ in range(3):
    copies()
print("baseline copies (us):", copies())
for size in (1 << 10, 8 << 20):
    dist.all_reduce(g[:size])
    print("after all_reduce(%d floats):" % size, copies())
    torch.cuda.synchronize(); time.sleep(0.01)
    print("  10 ms later:", copies())
w = dist.all_reduce(g, async_op=True)
r = copies()
w.wait()
print("concurrent with async all_reduce:", r)
dist.destroy_process_group()
