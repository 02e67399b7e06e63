# kernel stats of the SPLIT-SPAIR step: usage bash scripts/r02_spair_prof.sh <tag> [batch]
T=${1:-r02_spair}; B=${2:-32}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof
export SPAIR_ONLY=lg_spair
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o k -- python3 $R/scripts/bench_spair.py $B > $O/${T}_bench.json 2>/dev/null
cd $R
cat $O/${T}_bench.json
F=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY' | tee $O/${T}_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%6.2f%% %8d calls %9.1f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
