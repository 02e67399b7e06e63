# the native SPLIT-SPAIR step with / without the weight-gradient workspace (LDS-tile kernels + fixed-order slabs vs im2col + atomics)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "three_by_three" 2>&1 | tail -3
for r in 1 2 3; do for dt in f32 bf16; do for v in BASE=1 SV_TAPE_NO_WGRAD_WS=1; do echo -n "$dt $v: "; env $v python scripts/bench_spair_native.py 32 $dt 2>/dev/null | tail -1; done; done; done
SV_TRACE_DISPATCH=1 SPAIR_PROFILE=1 python scripts/bench_spair_native.py 32 f32 2>&1 | grep "wgrad_tile_f32" | sort | uniq -c
