R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp && export SV_DIST_FORCE=1
rm -rf $O/dptl
rocprofv3 --kernel-trace --output-format csv -d $O/dptl -o k -- python3 $R/bench.py --steps 30 --warmup 5 --batch 64 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
cd $R
python3 scripts/timeline.py $(find $O/dptl -name "*kernel_trace.csv" | head -1) --gaps > $O/r04_dp_timeline_b64.txt; cat $O/r04_dp_timeline_b64.txt
