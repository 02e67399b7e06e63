R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02t}
cd $R
for b in 64 128; do
  python bench.py --steps 200 --warmup 10 --batch $b --no-cpu-baseline --no-rows > $O/${T}_bench_b$b.json 2> $O/${T}_table_b$b.txt
  cut -c1-200 $O/${T}_bench_b$b.json
done
timeout 600 python -m pytest tests/test_gpu_fidelity.py -x -q -s 2>&1 | tail -5
