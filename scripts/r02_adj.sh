R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02s}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_ring or adjoint or d5_input" > $O/${T}_tests.txt 2>&1; tail -15 $O/${T}_tests.txt
for l in d5 d4 d3; do
  SV_BENCH_OPS=dgrad python scripts/bench_layers.py 1024 $l
  SV_NO_ROWCONV=1 SV_BENCH_OPS=dgrad python scripts/bench_layers.py 1024 $l
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_layers.txt
python bench.py --steps 100 --no-cpu-baseline --no-rows > $O/${T}_bench.json 2> $O/${T}_table.txt; cut -c1-250 $O/${T}_bench.json; head -16 $O/${T}_table.txt
SV_NO_FUSED_ADJOINT=1 python bench.py --steps 100 --no-cpu-baseline --no-rows 2>/dev/null | cut -c1-200
