#!/bin/bash
# knob sweep of the 2- and 4-rank shard sizes of config 4 (256 / 128 images) at fp32: bash scripts/r06_sweep_mid.sh <tag>
T=${1:-r06_m}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_sweep_f32_mid.txt
run() { local b=$1 k=$2; shift 2; echo -n "f32 B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype f32 --steps $k --warmup 8 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for cfg in "128 150" "256 100"; do set -- $cfg
run $1 $2 A=0
run $1 $2 A=1
run $1 $2 SV_WTF32_WGS=384
run $1 $2 SV_WTF32_WGS=768
run $1 $2 SV_WTF32_WGS=1024
run $1 $2 SV_WTF32_LDS=40000
run $1 $2 SV_WTF32_LDS=78000
run $1 $2 SV_TC_MF2=b
run $1 $2 SV_TC_MF2=c
run $1 $2 SV_TC_SMALL_WGS=400 SV_TC_TINY_WGS=200
run $1 $2 SV_TC_SMALL_WGS=800 SV_TC_TINY_WGS=400
run $1 $2 SV_TC_SMALL_WGS=100 SV_TC_TINY_WGS=50
run $1 $2 SV_TC_SMALL64_WGS=0
run $1 $2 SV_TC_SMALL64_WGS=600
run $1 $2 SV_TC_SMALL64_WGS=1200
run $1 $2 SV_TC_SMALL32_WGS=0
run $1 $2 SV_TC_SMALL32_WGS=600
run $1 $2 SV_TC_SMALL32_WGS=1200
run $1 $2 SV_CONV_SPLITK_TILES=0
run $1 $2 SV_CONV_SPLITK_TILES=64
run $1 $2 SV_SIDE_STREAMS=1
run $1 $2 SV_SIDE_STREAMS=2
run $1 $2 SV_POLYC_WGRAD_MIN=99999
run $1 $2 SV_NO_POLYC=1
run $1 $2 SV_NO_POLYD=1
run $1 $2 SV_NO_LATENT_FUSE=1
run $1 $2 SV_EARLY_SIDE=1
done
cat $OUT
