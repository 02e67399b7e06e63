# rocprofv3 kernel stats of one layer micro-benchmark.  usage: bash scripts/r02_kprof.sh <tag> <B> <layer> [ops]
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_kprof
export SV_BENCH_OPS=${4:-fwd}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kprof -o k -- python3 $R/scripts/bench_layers.py $2 $3 > /dev/null 2>&1
f=$(find $O/${T}_kprof -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | cut -c1-230 | head -8
