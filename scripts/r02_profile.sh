# One state's evidence set (GPU box): bench line + per-launch table, rocprofv3 kernel stats, PMC traffic, fidelity run.
# usage: bash scripts/r02_profile.sh <tag>   -> gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
T=${1:-r02_a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof $O/trafR $O/trafW
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o k -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-rows --no-fp32 > $O/${T}_prof_bench.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/trafR -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/trafW -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$O/${T}_prof/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
with open("$O/${T}_kernel_stats.csv", "w") as out:
    out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows:
        out.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
python3 scripts/traffic_summary.py $O/trafR $O/trafW > $O/${T}_traffic.json
mkdir -p profiles && cp $O/${T}_traffic.json profiles/${T}_traffic.json      # the bench line below cites this state's traffic
python bench.py > $O/${T}_bench.json 2> $O/${T}_table.txt
timeout 600 python -m pytest tests/test_gpu_fidelity.py -x -q -s 2>&1 | grep "bf16 fidelity\|passed\|failed" | tee $O/${T}_fidelity.txt
cp $O/bf16_fidelity.json $O/${T}_fidelity.json 2>/dev/null
cut -c1-300 $O/${T}_bench.json; head -12 $O/${T}_table.txt; head -12 $O/${T}_kernel_stats.csv | cut -c1-160
python3 -c "
import json; d=json.load(open('$O/${T}_traffic.json')); print('total traffic per step (GB):', d.get('total_bytes_per_step', 0)/1e9)"
