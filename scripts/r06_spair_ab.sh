#!/bin/bash
T=${1:-r06_sp}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_spair_ab.txt
: > $OUT
sp() { echo -n "spair $1 B=32 [${*:2}]: " >> $OUT; env "${@:2}" timeout 300 python scripts/bench_spair_native.py 32 $1 2>/dev/null | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT; }
for rep in 1 2; do for dt in f32 bf16; do sp $dt A=0; sp $dt SV_SPAIR_NOISE_LANE=0; sp $dt SV_TAPE_WGRAD_SIDE=0; sp $dt SV_TAPE_LANES=0; done; done
cat $OUT
timeout 900 python -m pytest tests/test_gpu_spair_model.py -m gpu -x -q 2>&1 | tail -2
