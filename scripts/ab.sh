export SV_BENCH_OPS=dgrad
echo "--- default"; python scripts/bench_layers.py 512 d4
echo "--- PAD64"; SV_TC_PAD64=1 python scripts/bench_layers.py 512 d4
echo "--- default"; python scripts/bench_layers.py 512 d4
echo "--- PAD64"; SV_TC_PAD64=1 python scripts/bench_layers.py 512 d4
