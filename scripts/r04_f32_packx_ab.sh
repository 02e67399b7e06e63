# the x-packed head (pixel pairs: 16 columns = 2 x (6 + 2), 42 taps) at fp32 -- forward on tile_conv<float>, weight gradient on wgrad_tile_f32<11, 1> with the folded
# reduce -- against the direct 6-of-16-column form (SV_NO_PACKX=1)     -> gpurun_out/<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04_f32_packx_ab}
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -q -x -k "float32 or fp32 or f32 or step" 2>&1 | tail -4
{
for v in "BASE=1" "SV_NO_PACKX=1"; do echo "== $v"; env $v python scripts/f32probe.py d5 2>&1 | tail -1; env $v python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms'], r['frac']) for r in d['roofline']['table'][:12]])"; done
} 2>&1 | tee $O/${T}.txt
