#!/bin/bash
# one SPLIT-SPAIR step's kernel timeline (start / end / queue): bash scripts/r06_spair_timeline.sh <tag>
T=${1:-r06_l}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_tl && SPAIR_PROFILE=1 rocprofv3 --kernel-trace --output-format csv -d $O/_tl -o k -- python3 $R/scripts/bench_spair_native.py 32 f32 > /dev/null 2>&1 )
python3 - $(find $O/_tl -name "*kernel_trace.csv" | head -1) > $O/${T}_spair_timeline.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "prep_table_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:60]
    q = qs.setdefault(r["Queue_Id"], len(qs))
    print("%8.1f %8.1f %7.1f q%d %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, q, n))
print("# span %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
PY
rm -rf $O/_tl; tail -1 $O/${T}_spair_timeline.txt
