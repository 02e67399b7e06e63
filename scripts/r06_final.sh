#!/bin/bash
# round-6 evidence set, ONE gpurun call: bash scripts/r06_final.sh <tag>      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
#   PMC traffic of the fp32 and the bf16 step (FETCH_SIZE / WRITE_SIZE in separate passes) FIRST (bench.py cites the newest profiles/<tag>_*traffic.json);
#   the bench line (fp32 headline with direct / issued fractions + bf16 block + rows + cpu_baseline) with its per-launch tables; rocprofv3 kernel stats of both steps,
#   in the step and serial; the per-kernel counter mix (matrix-pipe utilisation) of the fp32 serial table; one-step timelines of the 512- and 64-image fp32 steps;
#   the 64-image shard at both precisions; SPLIT-SPAIR / SPLIT-GMVAE kernel stats; the GPU test suite.
T=${1:-r06_f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
prof() {   # prof <out.csv> <bench args...>
  local out=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_prof && rocprofv3 --kernel-trace --stats --output-format csv -d $O/_prof -o k -- python3 $R/bench.py "$@" > /dev/null 2>&1 )
  python3 - <<PY
import csv, glob
f = glob.glob("$O/_prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))
    with open("$out", "w") as out:
        out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows:
            out.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
  rm -rf $O/_prof
}
traffic() {   # traffic <dtype> <out.json>
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_tR $O/_tW
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/_tR -o r -- python3 $R/bench.py --dtype $1 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-other-precision > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/_tW -o w -- python3 $R/bench.py --dtype $1 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-other-precision > /dev/null 2>&1 )
  python3 scripts/traffic_summary.py $O/_tR $O/_tW > $2
  rm -rf $O/_tR $O/_tW
}
mkdir -p profiles
traffic f32 $O/${T}_f32_traffic.json; cp $O/${T}_f32_traffic.json profiles/${T}_f32_traffic.json
traffic bf16 $O/${T}_traffic.json; cp $O/${T}_traffic.json profiles/${T}_traffic.json
python bench.py > $O/${T}_bench.json 2> $O/${T}_table.txt
prof $O/${T}_f32_kernel_stats.csv --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision
prof $O/${T}_f32_serial_kernel_stats.csv --dtype f32 --table-only 20
prof $O/${T}_bf16_kernel_stats.csv --dtype bf16 --steps 50 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision
prof $O/${T}_bf16_serial_kernel_stats.csv --dtype bf16 --table-only 30
bash scripts/r06_pmc_mix.sh $T f32 > /dev/null 2>&1
bash scripts/r05_timeline.sh $T f32 512 > /dev/null 2>&1
bash scripts/r05_timeline.sh $T f32 64 > /dev/null 2>&1
bash scripts/r05_timeline.sh $T bf16 64 > /dev/null 2>&1
for dt in f32 bf16; do timeout 300 python bench.py --batch 64 --dtype $dt --steps 200 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision > $O/${T}_bench_b64_$dt.json 2> $O/${T}_table_b64_$dt.txt; done
timeout 400 bash scripts/r03_spair_prof.sh $T > /dev/null 2>&1
GM_DTYPE=f32 timeout 400 bash scripts/r03_gm_prof.sh $T > /dev/null 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $O/${T}_gputests.log 2>&1; grep -E "passed|failed" $O/${T}_gputests.log | tail -1 > $O/${T}_gputests_tail.txt; grep -E "^FAILED" $O/${T}_gputests.log >> $O/${T}_gputests_tail.txt
cat $O/${T}_gputests_tail.txt; cut -c1-600 $O/${T}_bench.json | tail -2
