R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02e}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_ring" > $O/${T}_tests.txt 2>&1; tail -5 $O/${T}_tests.txt
SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py > /dev/null 2>&1
export SV_LIB_NAME=libsplitvae_dbg.so
for l in d4 d3; do
  for d in 0 1 2 3 4 8; do
    echo -n "dbg=$d "; SV_RC_DBG=$d SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 1024 $l
  done
  echo -n "B=128 "; SV_BENCH_OPS=fwd,dgrad python scripts/bench_layers.py 128 $l
done 2>&1 | grep -v amdgpu.ids | tee $O/${T}_abl.txt
