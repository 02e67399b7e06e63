"""Experiment: the whole step with the MAIN stream at high priority (the weight-gradient side streams are low priority already): no effect."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from split_vae_amd import dist as svdist
import split_vae_amd
split_vae_amd.configure_hw_queues()
dev = torch.device("cuda:0")
for prio in (0, -1, 0, -1):
    st = torch.cuda.Stream(priority=prio) if prio else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        w = bench.Workload(64, 512, "bf16", dev, 0, 1, svdist.make_reducer)
        dt = w.timed(200, 10, 1, dev)
    print("priority", prio, "ms_per_step %.4f" % (dt / 200 * 1e3), flush=True)
    del w
