# the object encoder's first layer (8-channel input, 3 x 3) on the fp32 LDS-tile weight gradient
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "three_by_three" 2>&1 | grep -E "passed|failed|Error" | tail -3
python -m pytest tests/test_gpu_spair_model.py -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2 3; do echo -n "f32: "; python scripts/bench_spair_native.py 32 f32 2>/dev/null | tail -1; done
SV_TRACE_DISPATCH=1 SPAIR_PROFILE=1 python scripts/bench_spair_native.py 32 f32 2>&1 | grep "wgrad_tile_f32" | sort | uniq -c
