#!/bin/bash
T=${1:-r06_sk}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_svhn_knobs.txt
run() { local sz=$1 b=$2 k=$3; shift 3; echo -n "f32 size $sz B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --size $sz --batch $b --dtype f32 --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
run 32 64 300 A=0
run 32 64 300 SV_TC_MF2=c
run 32 64 300 SV_TC_MF2=bc
run 32 64 300 SV_TC_MF2=b
run 32 64 300 SV_WTF32_WGS=256
run 32 64 300 SV_TC_SMALL_WGS=400 SV_TC_TINY_WGS=200
run 32 64 300 SV_WGRAD_MAIN=e1
run 32 64 300 SV_WGRAD_MAIN=e1,e2,e3
done
cat $OUT
