# rocprofv3 kernel stats of a python script: usage bash scripts/r02_kprof_script.sh <tag> <script> [args...]
T=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_kprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_kprof -o k -- python3 $R/$@ > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/${T}_kprof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
