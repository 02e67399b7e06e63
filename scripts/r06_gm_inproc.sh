#!/bin/bash
# the SPLIT-GMVAE rows as bench.py runs them: in ONE process, after other models
T=${1:-r06_gm4}; O=$GRAFT_REPO_ROOT/gpurun_out
cat > /tmp/gmb2.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, torch
dev = torch.device("cuda", 0)
w = bench.Workload(64, 64, "f32", dev, 0, 1); print("vae f32 b64", round(1e3 * w.timed(100, 10, 1, dev) / 100, 4)); del w
w = bench.Workload(64, 64, "bf16", dev, 0, 1); print("vae bf16 b64", round(1e3 * w.timed(200, 10, 1, dev) / 200, 4)); del w
for dt in ("f32", "bf16", "f32", "bf16"):
    r = bench.gm_row(dev, dtype=dt, steps=200); print("gm", dt, r["ms_per_step"])
r = bench.spair_row(dev, "hard"); print("spair hard", r["f32"]["ms_per_step"], r["bf16"]["ms_per_step"])
w = bench.Workload(64, 64, "bf16", dev, 0, 1); print("vae bf16 b64 again", round(1e3 * w.timed(200, 10, 1, dev) / 200, 4)); del w
PY
for s in 3 0; do echo "SV_GM_STREAMS=$s"; SV_GM_STREAMS=$s timeout 600 python /tmp/gmb2.py 2>/dev/null; done > $O/${T}_inproc.txt 2>&1
cat $O/${T}_inproc.txt
