export SV_BENCH_OPS=wgrad
L="d5 e1 e2"
echo "--- cur"; python scripts/bench_layers.py 512 $L
echo "--- SLAB_ALL"; SV_WT_SLAB_ALL=1 python scripts/bench_layers.py 512 $L
for d in 1 2 4 8; do echo "--- DBG=$d"; SV_WT_DBG=$d python scripts/bench_layers.py 512 d5 e1; done
