#!/bin/bash
run() { echo -n "$1 $2  "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-rows $2 2>gpurun_out/x_tbl_$3.txt | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; grep -E "^fwd.d5|^dgrad.d5" gpurun_out/x_tbl_$3.txt; }
run X=1 "" a; run "SV_TC_WRES_MAX=13 SV_TC_WRES_KB=56 SV_TC_VERBOSE=1" "" b; run X=1 "" c
grep "tile_conv plan" gpurun_out/x_tbl_b.txt | sort | uniq -c | head -20
