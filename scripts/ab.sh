for d in 0 1 2 4 6 8 14; do echo "--- WT_DBG=$d"; SV_WT_DBG=$d SV_BENCH_OPS=wgrad python scripts/bench_layers.py 512 d5 d4 d3 d2 e2; done
