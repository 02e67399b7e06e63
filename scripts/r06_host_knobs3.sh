#!/bin/bash
T=${1:-r06_h3}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_host_knobs.txt
: > $OUT
sp() { echo -n "spair f32 B=32 [$*]: " >> $OUT; env "$@" timeout 300 python scripts/bench_spair_native.py 32 f32 2>/dev/null | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['ms_per_step'], 'host', d['host_ms_per_step'])" >> $OUT; }
for rep in 1 2; do for q in 2 3 4; do for l in 0 1 2; do sp GPU_MAX_HW_QUEUES=$q SV_TAPE_LANES=$l; done; done; done
cat $OUT
