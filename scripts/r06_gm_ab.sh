#!/bin/bash
T=${1:-r06_gm6}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_gm_ab.txt
cat > /tmp/gmb.py <<PY
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, torch
r = bench.gm_row(torch.device("cuda", 0), dtype=sys.argv[1], steps=200)
print(sys.argv[1], r["ms_per_step"])
PY
: > $OUT
for rep in 1 2; do for dt in f32; do for k in A=0 SV_CONV_SPLITK_TILES=0 SV_CONV_SPLITK_TILES=32 SV_CONV_SPLITK_TILES=8; do echo -n "[$k] " >> $OUT; env $k timeout 300 python /tmp/gmb.py $dt 2>/dev/null >> $OUT; done; done; done

cat $OUT
timeout 900 python -m pytest tests/test_gpu_gm.py tests/test_gpu_kernels.py -k "gm or k_split" -m gpu -x -q 2>&1 | tail -2
