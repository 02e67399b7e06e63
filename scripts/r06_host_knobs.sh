#!/bin/bash
# runtime knobs against the launch-bound steps: bash scripts/r06_host_knobs.sh <tag>
T=${1:-r06_h}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_host_knobs.txt
: > $OUT
sp() { echo -n "spair f32 B=32 [$*]: " >> $OUT; env "$@" timeout 300 python scripts/bench_spair_native.py 32 f32 2>/dev/null | tail -1 >> $OUT; }
vb() { echo -n "vae bf16 B=64 [$*]: " >> $OUT; env "$@" timeout 200 python bench.py --batch 64 --dtype bf16 --steps 300 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
for k in A=0 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=8 AMD_DIRECT_DISPATCH=0 SV_TAPE_LANES=0; do sp $k; done
for k in A=0 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 HSA_ENABLE_INTERRUPT=0; do vb $k; done
cat $OUT
