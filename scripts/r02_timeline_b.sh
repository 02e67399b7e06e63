# kernel timeline of one step at a given per-GPU batch: usage bash scripts/r02_timeline_b.sh <tag> <batch>
T=${1:-r02_tl64}; B=${2:-64}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o k -- python3 $R/bench.py --steps 30 --warmup 5 --batch $B --no-cpu-baseline --no-rows > $O/${T}_prof_bench.json 2>/dev/null
cd $R
python3 scripts/timeline.py $(find $O/${T}_prof -name "*kernel_trace.csv" | head -1) | tee $O/${T}_timeline.txt
