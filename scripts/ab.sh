export SV_BENCH_OPS=wgrad
for d in 0 1 2 4 8 6 14 15; do echo "--- SV_WT_DBG=$d"; SV_WT_DBG=$d python scripts/bench_layers.py 512 d5 d4 d3 e2 2>&1 | grep -v amdgpu; done
