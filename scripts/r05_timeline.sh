#!/bin/bash
# one step's kernel timeline (launch order, queue, duration) + gap accounting: bash scripts/r05_timeline.sh <tag> <dtype> <batch>
T=${1:-r05_x}; DT=${2:-f32}; B=${3:-512}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_tl && rocprofv3 --kernel-trace --output-format csv -d $O/_tl -o k -- python3 $R/bench.py --batch $B --dtype $DT --steps 8 --warmup 3 --no-cpu-baseline --no-rows --no-other-precision > /dev/null 2>&1 )
python3 scripts/timeline.py $(find $O/_tl -name "*kernel_trace.csv" | head -1) --gaps > $O/${T}_${DT}_b${B}_timeline.txt
rm -rf $O/_tl
tail -3 $O/${T}_${DT}_b${B}_timeline.txt
