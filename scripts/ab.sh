export SV_BENCH_OPS=${SV_BENCH_OPS:-wgrad}
echo "--- pairx"; python scripts/bench_layers.py 512 e1
echo "--- no pairx"; SV_WT_NO_PAIRX=1 python scripts/bench_layers.py 512 e1
