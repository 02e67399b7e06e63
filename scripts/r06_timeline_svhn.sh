#!/bin/bash
T=${1:-r06_sv}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_tl && rocprofv3 --kernel-trace --output-format csv -d $O/_tl -o k -- python3 $R/bench.py --size 32 --batch 64 --dtype f32 --steps 8 --warmup 3 --no-cpu-baseline --no-rows --no-other-precision > /dev/null 2>&1 )
python3 scripts/timeline.py $(find $O/_tl -name "*kernel_trace.csv" | head -1) --gaps > $O/${T}_svhn32_f32_b64_timeline.txt
rm -rf $O/_tl
