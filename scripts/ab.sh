# same-box A/B of build variants / knobs (GPU box)
export SV_BENCH_OPS=${SV_BENCH_OPS:-fwd,dgrad}
L="d5 d4 d3 d2 e1 e2 e3"
echo "--- prev";  SV_LIB_NAME=libsplitvae_prev.so python scripts/bench_layers.py 512 $L
echo "--- cur"; python scripts/bench_layers.py 512 $L
echo "--- cur DBG=7 (floor)"; SV_TC_DBG=7 python scripts/bench_layers.py 512 $L
