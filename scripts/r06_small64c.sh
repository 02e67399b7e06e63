#!/bin/bash
T=${1:-r06_s64c}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_small64c.txt
run() { local sz=$1 b=$2 k=$3; shift 3; echo -n "f32 size $sz B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --size $sz --batch $b --dtype f32 --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
run 64 512 50 A=0
run 64 512 50 SV_TC_SMALL64_WGS=0 SV_TC_SMALL32_WGS=0
run 64 256 80 A=0
run 64 256 80 SV_TC_SMALL64_WGS=0 SV_TC_SMALL32_WGS=0
run 64 256 80 SV_TC_SMALL32_WGS=0
run 64 256 80 SV_TC_SMALL64_WGS=0
run 32 64 300 A=0
run 32 64 300 SV_TC_SMALL64_WGS=0 SV_TC_SMALL32_WGS=0
run 64 64 150 A=0
run 64 64 150 SV_TC_SMALL64_WGS=0 SV_TC_SMALL32_WGS=0
done
cat $OUT
