export SV_BENCH_OPS=wgrad
L="d3 d2 e1 e2"
echo "--- NG2 (2356)"; python scripts/bench_layers.py 512 $L
echo "--- NG1"; SV_WT_NG1=1 python scripts/bench_layers.py 512 $L
echo "--- NG2 (2356)"; python scripts/bench_layers.py 512 $L
echo "--- NG1"; SV_WT_NG1=1 python scripts/bench_layers.py 512 $L
