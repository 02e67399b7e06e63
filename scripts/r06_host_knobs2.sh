#!/bin/bash
T=${1:-r06_h2}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_host_knobs.txt
: > $OUT
sp() { echo -n "spair f32 B=32 [$*]: " >> $OUT; env "$@" timeout 300 python scripts/bench_spair_native.py 32 f32 2>/dev/null | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print(d['ms_per_step'], 'host', d['host_ms_per_step'])" >> $OUT; }
vb() { local dt=$1 b=$2 k=$3; shift 3; echo -n "vae $dt B=$b [$*]: " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype $dt --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
for dd in 1 0; do for q in 1 2 3 4; do for l in 0 1; do sp AMD_DIRECT_DISPATCH=$dd GPU_MAX_HW_QUEUES=$q SV_TAPE_LANES=$l; done; done; done
for dd in 1 0; do
vb bf16 64 300 AMD_DIRECT_DISPATCH=$dd
vb f32 64 150 AMD_DIRECT_DISPATCH=$dd
vb bf16 512 150 AMD_DIRECT_DISPATCH=$dd
vb f32 512 40 AMD_DIRECT_DISPATCH=$dd
done
cat $OUT
