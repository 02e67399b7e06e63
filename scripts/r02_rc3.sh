R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02g}
cd $R
SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py > /dev/null 2>&1
export SV_LIB_NAME=libsplitvae_dbg.so
for l in d4; do
  for d in 0 128 12 140; do
    echo -n "dbg=$d "; SV_RC_DBG=$d SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 1024 $l
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_abl.txt
