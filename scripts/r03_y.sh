#!/bin/bash
SV_DETERMINISTIC=1 timeout 300 python scripts/r03_step_hash.py 2>&1 | tail -4
SV_DETERMINISTIC=1 SV_FINALIZE_MAIN=1 SV_NO_LATENT_FUSE=1 timeout 300 python scripts/r03_step_hash.py 2>&1 | tail -4
run() { echo -n "$1 $2  "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-rows $2 2>gpurun_out/y_tbl_$3.txt | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; }
run X=1 "" a; run SV_FINALIZE_MAIN=1 "" b; run X=1 "" c; run SV_FINALIZE_MAIN=1 "" d
run X=1 "--batch 64" e; run SV_FINALIZE_MAIN=1 "--batch 64" f; run X=1 "--batch 64" g
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_graph.py tests/test_gpu_viz.py -x -q 2>&1 | grep -E "passed|failed"
