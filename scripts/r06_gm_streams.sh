#!/bin/bash
# SPLIT-GMVAE step with the two encoders on one / two HIP streams (SV_GM_STREAMS bit 0: forward, bit 1: backward), a fresh process per precision
T=${1:-r06_gm}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_gm_streams.txt
cat > /tmp/gmb.py <<PY
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, torch
dev = torch.device("cuda", 0)
dt = sys.argv[1]
r = bench.gm_row(dev, dtype=dt, steps=200)
print(dt, r["ms_per_step"], r["value"])
PY
: > $OUT
for rep in 1 2; do for s in 0 3; do for dt in f32 bf16; do echo -n "SV_GM_STREAMS=$s: " >> $OUT; SV_GM_STREAMS=$s timeout 300 python /tmp/gmb.py $dt 2>/dev/null >> $OUT; done; done; done
for q in 2 4; do for dt in f32 bf16; do echo -n "GPU_MAX_HW_QUEUES=$q SV_GM_STREAMS=3: " >> $OUT; GPU_MAX_HW_QUEUES=$q SV_GM_STREAMS=3 timeout 300 python /tmp/gmb.py $dt 2>/dev/null >> $OUT; done; done
cat $OUT
