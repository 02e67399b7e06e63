export SV_BENCH_OPS=fwd,dgrad
for d in 0 1 2 4 8 16 24 25 31; do echo "--- SV_TC_DBG=$d"; SV_TC_DBG=$d python scripts/bench_layers.py 512 d5 d4 2>&1 | grep -v amdgpu; done
