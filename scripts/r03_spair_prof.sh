#!/bin/bash
# rocprofv3 kernel-trace summary of the native SPLIT-SPAIR step (README.md:93's model, batch 32) -> gpurun_out/<tag>_spair_native_kernel_stats.txt
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_spn
SPAIR_PROFILE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_spn -o spn --output-format csv -- python3 $ROOT/scripts/bench_spair_native.py > /tmp/prof_spn.log 2>&1
f=$(find /tmp/prof_spn -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then
  head -60 "$f" | cut -c1-260 > $ROOT/gpurun_out/${TAG}_spair_native_kernel_stats.txt
  python3 - "$f" <<'PY' >> $ROOT/gpurun_out/${TAG}_spair_native_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# steps 30 (10 warm-up + 20 timed): %.1f launches and %.3f ms of kernel time per step" % (calls / 30.0, tot / 30.0 / 1e6))
PY
else
  echo "no kernel_stats.csv" > $ROOT/gpurun_out/${TAG}_spair_native_kernel_stats.txt; tail -20 /tmp/prof_spn.log >> $ROOT/gpurun_out/${TAG}_spair_native_kernel_stats.txt
fi
