# kernel timeline of the data-parallel step on one rank: usage bash scripts/r02_dp_timeline.sh <tag> [backend]
T=${1:-r02_dp}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export SV_DIST_BACKEND=${2:-sv_comm}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof
rocprofv3 --kernel-trace --output-format csv -d $O/${T}_prof -o k -- python3 $R/scripts/exp_dp_timeline.py > /dev/null 2>&1
cd $R
python3 scripts/timeline.py $(find $O/${T}_prof -name "*kernel_trace.csv" | head -1) | tee $O/${T}_timeline.txt
