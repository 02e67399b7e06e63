R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02_poly}
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "polyphase or fused_upsample or conv_fwd_dgrad" 2>&1 | tail -15 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
SV_BENCH_OPS=fwd python scripts/bench_layers.py 1024 d5 2>&1 | grep -v amdgpu
SV_NO_POLY=1 SV_BENCH_OPS=fwd python scripts/bench_layers.py 1024 d5 2>&1 | grep -v amdgpu
GREP="fwd.d5" bash scripts/r02_ab.sh ${T} "SV_WGRAD_MAIN=e1,e2,d5" "SV_WGRAD_MAIN=e1,e2,d5 SV_NO_POLY=1"
