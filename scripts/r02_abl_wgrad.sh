# ablation of the wgrad tile kernel phases (debug-knob build).  SV_WT_DBG bits: 1 skip flush(+reduce), 2 skip input staging, 4 skip dY staging, 8 skip the MFMA loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02_wabl}; shift
cd $R
SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py > /dev/null 2>&1
export SV_LIB_NAME=libsplitvae_dbg.so SV_BENCH_OPS=wgrad
for d in 0 1 2 4 8 6 14 15; do echo -n "dbg=$d "; SV_WT_DBG=$d python scripts/bench_layers.py 1024 ${@:-d5 d4} 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done | tee $O/${T}.txt
