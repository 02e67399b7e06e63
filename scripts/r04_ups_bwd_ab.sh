# the fp32 resize adjoint: register-window rows kernel vs one thread per output (SV_UPS_BWD_PLAIN=1)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "resize_adjoint or upsample" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for r in 1 2 3; do for v in BASE=1 SV_UPS_BWD_PLAIN=1; do echo -n "f32 $v: "; env $v python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done; done
