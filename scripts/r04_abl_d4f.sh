# phase ablation + in-kernel stamps of the d4 forward (debug-knob build: SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py)
export SV_LIB_NAME=libsplitvae_dbg.so SV_BENCH_OPS=fwd
for v in "" "SV_RC_NO_MB=1"; do
  echo "#### ${v:-MB on}"
  for d in 0 1 2 4 8 9 13 3 7 15; do echo -n "dbg=$d "; env $v SV_RC_DBG=$d timeout 120 python scripts/bench_layers.py 1024 d4 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done
  env $v SV_RC_STAMP=1 timeout 120 python scripts/bench_layers.py 1024 d4 2>&1 | grep -v amdgpu | tail -4
done
