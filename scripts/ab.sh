export SV_BENCH_OPS=${SV_BENCH_OPS:-fwd,wgrad}
L="d5 d4 d3"
echo "--- np (previous)";  SV_LIB_NAME=libsplitvae_np.so python scripts/bench_layers.py 512 $L
echo "--- cur"; python scripts/bench_layers.py 512 $L
echo "--- np (previous)";  SV_LIB_NAME=libsplitvae_np.so python scripts/bench_layers.py 512 $L
echo "--- cur"; python scripts/bench_layers.py 512 $L
