# fp32 step: the heads' / d1's weight gradients on dense_f32.hip vs the im2col conv kernel (SV_NO_DENSE_F32_WGRAD=1)
# (experiment of round 4, NOT kept: the routing and its SV_NO_DENSE_F32_WGRAD knob were removed again -- LAB_NOTES 4k; kept for the record of how profiles/r04_j_dense_f32_wgrad_ab.txt was made)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_step.py -q -x -k "fp32 or f32 or float32" 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2 3; do for v in BASE=1 SV_NO_DENSE_F32_WGRAD=1; do echo -n "f32 $v: "; env $v python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done; done
