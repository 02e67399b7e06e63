#!/bin/bash
T=${1:-r06_l}; O=$GRAFT_REPO_ROOT/gpurun_out
{
for rep in 1 2; do
echo -n "lanes f32: "; timeout 300 python scripts/bench_spair_native.py 32 f32 | tail -1
echo -n "one stream f32: "; SV_TAPE_LANES=0 timeout 300 python scripts/bench_spair_native.py 32 f32 | tail -1
done
echo -n "lanes bf16: "; timeout 300 python scripts/bench_spair_native.py 32 bf16 | tail -1
echo -n "one stream bf16: "; SV_TAPE_LANES=0 timeout 300 python scripts/bench_spair_native.py 32 bf16 | tail -1
echo -n "lanes easy f32: "; SPAIR_FLAGS=easy timeout 300 python scripts/bench_spair_native.py 32 f32 | tail -1
echo -n "one stream easy f32: "; SPAIR_FLAGS=easy SV_TAPE_LANES=0 timeout 300 python scripts/bench_spair_native.py 32 f32 | tail -1
} > $O/${T}_spair_lanes.txt 2>&1
cat $O/${T}_spair_lanes.txt
timeout 900 python -m pytest tests/test_gpu_spair_model.py tests/test_gpu_spair_ops.py -m gpu -x -q 2>&1 | tail -3
