#!/bin/bash
# A/B of the round-6 step changes: bash scripts/r06_ab.sh <tag>
T=${1:-r06_ab}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_ab.txt
run() { local dt=$1 b=$2 k=$3; shift 3; echo -n "$dt B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype $dt --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2; do
run f32 512 60 A=0
run f32 512 60 SV_NO_EARLY_SIDE=1
run f32 512 60 SV_NO_FUSED_NLL_F32=1
run f32 512 60 SV_NO_EARLY_SIDE=1 SV_NO_FUSED_NLL_F32=1
run bf16 512 200 A=0
run bf16 512 200 SV_NO_EARLY_SIDE=1
run bf16 64 300 A=0
run bf16 64 300 SV_NO_EARLY_SIDE=1
run f32 64 200 A=0
run f32 64 200 SV_NO_EARLY_SIDE=1
run f32 256 100 A=0
run f32 256 100 SV_NO_EARLY_SIDE=1
done
cat $OUT
