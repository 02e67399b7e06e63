export SV_BENCH_OPS=fwd,dgrad
for lib in libsplitvae_hip.so libsplitvae_p32.so libsplitvae_hip.so libsplitvae_p32.so; do
echo "--- $lib"; SV_LIB_NAME=$lib python scripts/bench_layers.py 512 d5 d4; done
