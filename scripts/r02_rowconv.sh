# row-ring kernel: numerics + per-layer timing A/B against the tile kernel (GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02c}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_ring" > $O/${T}_tests.txt 2>&1; tail -15 $O/${T}_tests.txt
for l in d3 d4; do
  SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 1024 $l
  SV_NO_ROWCONV=1 SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 1024 $l
  SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 128 $l
  SV_NO_ROWCONV=1 SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 128 $l
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_layers.txt
