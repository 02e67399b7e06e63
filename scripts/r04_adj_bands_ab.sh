# row bands for the fused-adjoint input gradients (d5 / d4 / d3) at small batches (SV_RC_ADJ_BANDS=0: whole images per workgroup)   -> gpurun_out/<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04_adj_bands_ab}
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_dist.py -q -x -k "adjoint or row_ring or step or d5_input or dist or matrix_pipe" 2>&1 | tail -3
{
for B in 64 128 256; do for r in 1 2; do for v in "BASE=1" "SV_RC_ADJ_BANDS=0"; do echo -n "B=$B $v: "; env $v python bench.py --batch $B --steps 200 --warmup 20 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms']) for r in d['roofline']['table'][:12] if 'dgrad.d' in r['kernel']])"; done; done; done
for v in "BASE=1" "SV_RC_ADJ_BANDS=0"; do echo -n "B=512 $v: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
} 2>&1 | tee $O/${T}.txt
