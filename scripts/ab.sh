export SV_BENCH_OPS=${SV_BENCH_OPS:-fwd}
for lib in libsplitvae_p8.so libsplitvae_p16.so libsplitvae_hip.so; do
echo "--- $lib packx / unpacked"; SV_LIB_NAME=$lib python scripts/bench_layers.py 512 d5;  SV_LIB_NAME=$lib SV_NO_PACKX=1 python scripts/bench_layers.py 512 d5; done
