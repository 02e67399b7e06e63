export SV_BENCH_OPS=wgrad
L="d5 d4 d3 d2 e1 e2"
echo "--- contiguous"; python scripts/bench_layers.py 512 $L
echo "--- strided"; SV_WT_STRIDED=1 python scripts/bench_layers.py 512 $L
echo "--- contiguous"; python scripts/bench_layers.py 512 $L
echo "--- strided"; SV_WT_STRIDED=1 python scripts/bench_layers.py 512 $L
