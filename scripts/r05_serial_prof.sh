#!/bin/bash
# rocprofv3 kernel stats of the serial per-launch table (every launch alone on the chip): bash scripts/r05_serial_prof.sh <tag> [f32|bf16] [passes]
T=${1:-r05_x}; DT=${2:-f32}; N=${3:-10}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/${T}_prof_serial && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_serial -o k -- python3 $R/bench.py --dtype $DT --table-only $N > $O/${T}_prof_serial_bench.json 2>/dev/null )
python3 - <<PY
import csv, glob
f = glob.glob("$O/${T}_prof_serial/**/*kernel_stats.csv", recursive=True)
if f:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))
    with open("$O/${T}_${DT}_serial_kernel_stats.csv", "w") as out:
        out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows:
            out.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
rm -rf $O/${T}_prof_serial
