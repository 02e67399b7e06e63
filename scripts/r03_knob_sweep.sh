#!/bin/bash
# re-check round-2 decisions under the round-3 schedule (two side streams): each knob alone against the default
run() { echo -n "$1  "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-rows 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; }
run X=1; for k in "$@"; do run $k; done; run X=1
