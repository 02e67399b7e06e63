#!/bin/bash
# round 6, first GPU call: the tests the advisor fixes touch, the restructured bench line, one-step timelines of config 4's shard
T=${1:-r06_a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_kernels.py -m gpu -x -q -k "dist or polyphase_forward or conv_fwd" > $O/${T}_tests.log 2>&1; tail -3 $O/${T}_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_table.txt; cut -c1-1500 $O/${T}_bench.json
bash scripts/r05_timeline.sh $T f32 64
bash scripts/r05_timeline.sh $T bf16 64
timeout 300 python bench.py --batch 64 --dtype f32 --steps 200 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision > $O/${T}_bench_b64_f32.json 2> $O/${T}_table_b64_f32.txt
