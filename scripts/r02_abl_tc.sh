# ablation of the LDS-tile conv kernel phases (debug-knob build): usage bash scripts/r02_abl_tc.sh <tag> <layers...>
# SV_TC_DBG bits: 1 skip input staging, 2 skip the K loop, 4 skip the store, 8 skip weight streaming, 16 skip the per-step barrier, 32 return at once
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
cd $R
SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py > /dev/null 2>&1
export SV_LIB_NAME=libsplitvae_dbg.so
for l in "$@"; do
  for d in ${DBGS:-0 1 2 4 3 5 6 7 8 24}; do
    echo -n "dbg=$d "; SV_NO_ROWCONV=${NOROW:-} SV_TC_DBG=$d SV_BENCH_OPS=${OPS:-fwd,dgrad} python scripts/bench_layers.py ${BB:-1024} $l
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_abl.txt
