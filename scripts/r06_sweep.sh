#!/bin/bash
# knob sweep of one shard size: bash scripts/r06_sweep.sh <tag> <dtype> <batch> <steps>   (each line: ms/step of a fresh process with one knob changed)
T=${1:-r06_s}; DT=${2:-f32}; B=${3:-64}; K=${4:-150}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_sweep_${DT}_b${B}.txt
run() { echo -n "$* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $B --dtype $DT --steps $K --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
run A=0
run A=1
run SV_NO_EARLY_SIDE=1
run SV_NO_FUSED_NLL_F32=1
run SV_WTF32_WGS=256
run SV_WTF32_WGS=384
run SV_WTF32_WGS=768
run SV_WTF32_LDS=40000
run SV_WTF32_LDS=78000
run SV_TC_MF2=b
run SV_TC_MF2=c
run SV_TC_MF2=bc
run SV_TC_SMALL_WGS=400 SV_TC_TINY_WGS=200
run SV_TC_SMALL_WGS=100 SV_TC_TINY_WGS=50
run SV_SIDE_STREAMS=2
run SV_WGRAD_MAIN=e1
run SV_WGRAD_MAIN=e1,e2,e3
run SV_WGRAD_MAIN=
run SV_POLYC_WGRAD_MIN=99999
run SV_NO_POLYC=1
run SV_NO_POLYD=1
run SV_NO_POLYC=1 SV_NO_POLYD=1 SV_POLYC_WGRAD_MIN=99999
run SV_NO_POLY_F32=1
run SV_TC_NO_DMA=1
run SV_WT32_NO_DMA=1
run SV_NO_LATENT_FUSE=1
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=2
cat $OUT
