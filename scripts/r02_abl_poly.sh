R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02_polyabl}
cd $R
SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py > /dev/null 2>&1
export SV_LIB_NAME=libsplitvae_dbg.so SV_BENCH_OPS=fwd
for v in "SV_X=0" "SV_PF_DBG=7" "SV_PF_DBG=1" "SV_PF_DBG=2" "SV_PF_DBG=4" "SV_PF_DBG=7 SV_TC_DBG=1" "SV_PF_DBG=7 SV_TC_DBG=2" "SV_PF_DBG=7 SV_TC_DBG=4" "SV_PF_DBG=7 SV_TC_DBG=7" "SV_PF_DBG=7 SV_TC_DBG=8" "SV_PF_DBG=7 SV_TC_DBG=24"; do
  echo -n "$v: "; env $v python scripts/bench_layers.py 1024 d5 2>&1 | grep -v amdgpu
done | tee $O/${T}.txt
