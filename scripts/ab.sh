export SV_BENCH_OPS=fwd,dgrad
echo "--- N split"; python scripts/bench_layers.py 512 e3 d2
echo "--- no N split"; SV_TC_NO_NSPLIT=1 python scripts/bench_layers.py 512 e3 d2
echo "--- N split"; python scripts/bench_layers.py 512 e3 d2
echo "--- no N split"; SV_TC_NO_NSPLIT=1 python scripts/bench_layers.py 512 e3 d2
