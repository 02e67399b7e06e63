# Round-2 baseline on the GPU box: device tests, then the small-shard configs VERDICT r01 asked for.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02a}
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${T}_gputests.txt 2>&1; tail -3 $O/${T}_gputests.txt
for b in 64 128 512; do
  python bench.py --steps 100 --warmup 10 --batch $b --no-cpu-baseline --profile-all > $O/${T}_bench_b$b.json 2> $O/${T}_table_b$b.txt
  cut -c1-200 $O/${T}_bench_b$b.json
done
python bench.py --steps 100 --warmup 10 --size 32 --batch 64 --no-cpu-baseline > $O/${T}_bench_svhn32_b64.json 2>/dev/null; cut -c1-200 $O/${T}_bench_svhn32_b64.json
