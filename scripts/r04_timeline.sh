# kernel timeline of one step at a given per-GPU batch (rocprofv3 kernel trace) + gap accounting on the main queue
# usage: bash scripts/r04_timeline.sh <tag> <batch>      -> gpurun_out/<tag>_timeline_b<batch>.txt
T=${1:-r04_tl}; B=${2:-64}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof_b$B
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_b$B -o k -- python3 $R/bench.py --steps 30 --warmup 5 --batch $B --no-cpu-baseline --no-rows --no-fp32 > $O/${T}_prof_b${B}_bench.json 2>/dev/null
cd $R
python3 scripts/timeline.py $(find $O/${T}_prof_b$B -name "*kernel_trace.csv" | head -1) --gaps | tee $O/${T}_timeline_b$B.txt
