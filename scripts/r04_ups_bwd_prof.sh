# per-kernel time of the fp32 resize adjoint, both forms (rocprofv3 --kernel-trace --stats of a short fp32 bench)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pa /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o k -- python3 $R/bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
export SV_UPS_BWD_PLAIN=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o k -- python3 $R/bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
for d in /tmp/pa /tmp/pb; do f=$(find $d -name '*kernel_stats.csv' | head -1); echo "== $d"; grep -i "upsample" $f | cut -c1-200; done
