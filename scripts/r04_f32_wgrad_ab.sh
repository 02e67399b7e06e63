# the fp32 tile weight gradient (wgrad_tile_f32.hip) against the im2col kernel (SV_NO_WGRAD_TILE_F32=1): parity tests, per-layer and whole fp32 step   -> gpurun_out/<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04_f32_wgrad_ab}
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -q -x 2>&1 | tail -4
{
for v in "BASE=1" "SV_NO_WGRAD_TILE_F32=1"; do echo "== $v"; env $v python scripts/f32probe.py 2>&1 | tail -1; env $v python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms'], r['frac']) for r in d['roofline']['table'][:12]])"; done
} 2>&1 | tee $O/${T}.txt
