#!/bin/bash
T=${1:-r06_sf}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_sweep_f32_big.txt
run() { local b=$1 k=$2; shift 2; echo -n "f32 B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype f32 --steps $k --warmup 8 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
for cfg in "512 40" "256 80" "128 120"; do set -- $cfg
run $1 $2 A=0
run $1 $2 SV_SIDE_STREAMS=1
run $1 $2 SV_SIDE_STREAMS=2
run $1 $2 SV_WGRAD_MAIN=e1,e2
run $1 $2 SV_WGRAD_MAIN=e1
run $1 $2 SV_WGRAD_MAIN=e1,e2,e3,d2
done; done
cat $OUT
