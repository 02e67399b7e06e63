R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02n}
cd $R
SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py 2>&1 | grep -i "error" | head
export SV_LIB_NAME=libsplitvae_dbg.so
for d in 0 4 1; do
  echo "dbg=$d "; SV_RC_STAMP=1 SV_RC_DBG=$d SV_BENCH_OPS=dgrad python scripts/bench_layers.py 1024 d4
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_stamps.txt
