# phase ablation of the tile weight-gradient kernel on the small-grid layers (debug-knob build, made in the container and shipped with the snapshot).
# SV_WT_DBG bits: 1 skip flush (+reduce), 2 skip input staging, 4 skip dY staging, 8 skip the MFMA loop     -> gpurun_out/<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04_abl_wt}; shift
cd $R
export SV_LIB_NAME=libsplitvae_dbg.so SV_BENCH_OPS=wgrad
for d in 0 1 2 4 8 6 7 14 15; do echo -n "dbg=$d "; SV_WT_DBG=$d python scripts/bench_layers.py 1024 ${@:-d2 e3 d3} 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done | tee $O/${T}.txt
