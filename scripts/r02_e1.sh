R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02u}
cd $R
for v in "X=1" "SV_NO_FUSED_ADJOINT=1" "SV_NO_ROWCONV=1"; do
  echo "== $v"; env $v python bench.py --steps 50 --batch 64 --no-cpu-baseline --no-rows 2>&1 >/dev/null | grep "wgrad.e1 \|dgrad.e2\|wgrad.e2 \|adam"
done
SV_BENCH_OPS=wgrad python scripts/bench_layers.py 128 e1 e2
