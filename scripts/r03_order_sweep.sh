#!/bin/bash
for o in "$@"; do echo -n "SV_SIDE_ORDER=$o  "; SV_SIDE_ORDER=$o timeout 300 python bench.py --no-cpu-baseline --no-rows 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; done
