#!/bin/bash
# stream-placement sweep of the bf16 step: bash scripts/r06_sweep_bf16.sh <tag> <batch> <steps>
T=${1:-r06_sb}; B=${2:-256}; K=${3:-300}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_sweep_bf16_b${B}.txt
run() { echo -n "$* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $B --dtype bf16 --steps $K --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
run A=0
run SV_WGRAD_MAIN=e1,e2
run SV_WGRAD_MAIN=e1,e2,e3
run SV_WGRAD_MAIN=e2
run SV_WGRAD_MAIN=e1,e3
run SV_WGRAD_MAIN=e1,e2,d1
run SV_SIDE_STREAMS=2
run SV_SIDE_STREAMS=2 SV_WGRAD_MAIN=e1,e2
run SV_SIDE_STREAMS=1
done
cat $OUT
