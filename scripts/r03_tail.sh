#!/bin/bash
# early optimizer tail + weight images one step ahead: same bits as the plain tail, then the step with and without them
export SV_DETERMINISTIC=1
for k in X=1 SV_NO_EARLY_TAIL=1 SV_NO_PREP_AHEAD=1 "SV_NO_EARLY_TAIL=1 SV_NO_PREP_AHEAD=1"; do echo "== $k"; env $k timeout 300 python scripts/r03_step_hash.py 2>&1 | tail -4; done
unset SV_DETERMINISTIC
run() { echo -n "$1 $2  "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-rows $2 2>gpurun_out/tail_tbl_$3.txt | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; }
run X=1 "" a; run SV_NO_EARLY_TAIL=1 "" b; run SV_NO_PREP_AHEAD=1 "" c; run "SV_NO_EARLY_TAIL=1 SV_NO_PREP_AHEAD=1" "" d; run X=1 "" e
run X=1 "--batch 64" f; run "SV_NO_EARLY_TAIL=1 SV_NO_PREP_AHEAD=1" "--batch 64" g; run "SV_NO_EARLY_TAIL=1" "--batch 64" g2; run X=1 "--batch 64" h
run X=1 "--batch 256" i; run "SV_NO_EARLY_TAIL=1 SV_NO_PREP_AHEAD=1" "--batch 256" j
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py tests/test_gpu_dist.py -x -q 2>&1 | tail -5
