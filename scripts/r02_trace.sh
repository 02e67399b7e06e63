# rocprofv3 kernel trace of the bench (GPU box): usage bash scripts/r02_trace.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
rm -rf $O/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o k -- python3 $R/bench.py --no-cpu-baseline --no-rows "$@" > $O/${T}_prof_bench.json 2>/dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$O/${T}_prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
out = open("$O/${T}_kernel_stats.csv", "w")
out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
for r in rows:
    out.write('"%s",%s,%s,%s,%s\n' % (r["Name"][:150], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
for r in rows[:28]:
    print("%-100s n=%5s avg %9.1f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
