#!/bin/bash
T=${1:-r06_pr}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_prio_ab.txt
run() { local dt=$1 b=$2 k=$3; shift 3; echo -n "$dt B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype $dt --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
for cfg in "f32 512 50" "f32 64 150" "bf16 512 200" "bf16 256 300" "bf16 64 300"; do set -- $cfg; run $1 $2 $3 A=0; run $1 $2 $3 SV_SIDE_PRIO_NORMAL=1; done
done
cat $OUT
