# One state's evidence set (GPU box): bench line + hipEvent table, rocprofv3 kernel stats, PMC traffic, the other configs.
# usage: bash scripts/final_profile.sh <tag>   -> gpurun_out/<tag>_*
T=${1:-r01x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o k -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/${T}_prof_bench.json 2>/dev/null
cd $R
bash scripts/traffic.sh > /dev/null 2>&1
cp gpurun_out/traffic.json $O/${T}_traffic.json
cp $O/${T}_traffic.json profiles/$(echo $T | sed 's/r01/r01_/')_traffic.json 2>/dev/null   # so that the bench line below cites this state's traffic
python bench.py --steps 30 --warmup 5 --profile-all > $O/${T}_bench.json 2> $O/${T}_table.txt
python bench.py --steps 30 --warmup 5 --batch 256 --no-cpu-baseline > $O/${T}_bench_b256.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --batch 1024 --no-cpu-baseline > $O/${T}_bench_b1024.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --dtype f32 --no-cpu-baseline > $O/${T}_bench_f32.json 2>/dev/null
python bench.py --steps 50 --warmup 5 --size 32 --batch 64 --no-cpu-baseline > $O/${T}_bench_svhn32_b64.json 2>/dev/null
for f in $O/${T}_bench*.json; do echo $f; cut -c1-230 $f; done
