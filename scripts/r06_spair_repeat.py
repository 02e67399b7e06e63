"""Repeat bench.py's SPLIT-SPAIR rows in one process (row-to-row variance of the host-bound step): python scripts/r06_spair_repeat.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import split_vae_amd                     # noqa: E402
split_vae_amd.configure_hw_queues()
import torch                             # noqa: E402
import bench                             # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
for i in range(n):
    for which in ("hard", "easy"):
        r = bench.spair_row(dev, which)
        print(i, which, "f32 %.3f ms  bf16 %.3f ms" % (r["f32"]["ms_per_step"], r["bf16"]["ms_per_step"]), flush=True)
