#!/bin/bash
T=${1:-r06_s32}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_small32.txt
run() { local sz=$1 b=$2 k=$3; shift 3; echo -n "f32 size $sz B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --size $sz --batch $b --dtype f32 --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
for th in 0 150 300; do run 32 64 300 SV_TC_SMALL32_WGS=$th; done
run 32 64 300 SV_TC_MF2=b
for th in 0 150 300 600; do run 64 64 150 SV_TC_SMALL32_WGS=$th; done
for th in 0 150 300; do run 32 256 150 SV_TC_SMALL32_WGS=$th; done
for th in 0 300 600; do run 64 128 100 SV_TC_SMALL32_WGS=$th; done
done
cat $OUT
