R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02y}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_upsample or row_ring or adjoint" 2>&1 | tail -4
for l in d3 d4 d5; do SV_BENCH_OPS=fwd,wgrad python scripts/bench_layers.py 1024 $l; done 2>&1 | grep -v amdgpu
SV_LIB_NAME=libsplitvae_old.so SV_OBJ_TAG=_old SV_EXTRA_FLAGS=-DSV_STAGE_BLOCKS python split_vae_amd/build.py > /dev/null 2>&1
for l in d3 d4 d5; do SV_LIB_NAME=libsplitvae_old.so SV_BENCH_OPS=fwd,wgrad python scripts/bench_layers.py 1024 $l; done 2>&1 | grep -v amdgpu
python bench.py --steps 100 --no-cpu-baseline --no-rows 2>/dev/null | cut -c1-200
SV_LIB_NAME=libsplitvae_old.so python bench.py --steps 100 --no-cpu-baseline --no-rows 2>/dev/null | cut -c1-200
