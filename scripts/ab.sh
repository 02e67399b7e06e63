# Same-box A/B of a tuning knob or build variant on the per-layer microbenchmark (run on the GPU box:
#   gpurun -- 'bash scripts/ab.sh').  Boxes differ by a few per cent, so every comparison in DESIGN.md was made
# inside one invocation, alternating the arms.  Edit KNOB / LAYERS / OPS for the experiment at hand; knobs are the
# SV_* environment variables documented next to their getenv() in split_vae_amd/csrc (e.g. SV_NO_PACKX,
# SV_TC_NO_PLANAR, SV_WT_NG1, SV_TC_NW8=abc, SV_SPLITK_SMALL=1) or a second library built with
#   SV_LIB_NAME=libsplitvae_x.so SV_OBJ_TAG=x SV_EXTRA_FLAGS="-DSV_TC_PPS32=16" python split_vae_amd/build.py
KNOB=${KNOB:-SV_WT_NG1=1}
LAYERS=${LAYERS:-"d5 d4 d3 d2 e1 e2 e3"}
export SV_BENCH_OPS=${OPS:-fwd,dgrad,wgrad}
for rep in 1 2; do
  echo "--- default";  python scripts/bench_layers.py 512 $LAYERS
  echo "--- $KNOB";    env $KNOB python scripts/bench_layers.py 512 $LAYERS
done
