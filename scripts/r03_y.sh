timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_ring or adjoint or d5 or dgrad or upsample" 2>&1 | grep -E "passed|failed"
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed"
run() { echo -n "$1 $2  "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-rows $2 2>gpurun_out/y_tbl_$3.txt | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; grep -E "^dgrad.d[345]" gpurun_out/y_tbl_$3.txt; }
run X=1 "" a; run X=1 "" b
