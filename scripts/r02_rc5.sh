R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02o}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_ring" > $O/${T}_tests.txt 2>&1; tail -5 $O/${T}_tests.txt
for l in d4 d3; do
  SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 1024 $l
  SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 128 $l
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_layers.txt
