export SV_BENCH_OPS=${SV_BENCH_OPS:-fwd,dgrad}
L="d5 d4 d3 d2 e1 e2 e3"
echo "--- np (previous loop)";  SV_LIB_NAME=libsplitvae_np.so python scripts/bench_layers.py 512 $L
echo "--- cur"; python scripts/bench_layers.py 512 $L
echo "--- np (previous loop)";  SV_LIB_NAME=libsplitvae_np.so python scripts/bench_layers.py 512 $L
echo "--- cur"; python scripts/bench_layers.py 512 $L
