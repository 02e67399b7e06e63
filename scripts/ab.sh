export SV_BENCH_OPS=${SV_BENCH_OPS:-wgrad}
echo "--- packx"; python scripts/bench_layers.py 512 d5
echo "--- no packx"; SV_NO_PACKX=1 python scripts/bench_layers.py 512 d5
for d in 1 2 4 8; do echo "--- packx DBG=$d"; SV_WT_DBG=$d python scripts/bench_layers.py 512 d5; done
