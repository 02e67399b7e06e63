#!/bin/bash
T=${1:-r06_pd}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_polyd_depth.txt
run() { local dt=$1 b=$2 k=$3; shift 3; echo -n "$dt B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype $dt --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
: > $OUT
for rep in 1 2 3; do
for cfg in "f32 512 50" "f32 64 150" "f32 256 80"; do set -- $cfg; run $1 $2 $3 SV_POLYD_DEPTH=2; run $1 $2 $3 SV_POLYD_DEPTH=1; done
done
cat $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "polyphase_input_gradient" 2>&1 | tail -2
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3
