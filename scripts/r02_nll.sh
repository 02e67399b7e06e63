R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02_nll}
cd $R
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu 2>&1 | tail -15 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
GREP="fwd.d5\|dlogistic" bash scripts/r02_ab.sh ${T} "SV_X=1" "SV_NO_FUSED_NLL=1"
