#!/bin/bash
# which weight gradients stay on the main stream, with the rolling-window d4 kernel (a persistent one-workgroup-per-CU launch co-runs badly)
for m in "$@"; do
  echo -n "SV_WGRAD_MAIN=$m  "
  SV_WGRAD_MAIN=$m timeout 300 python bench.py --batch ${BATCH:-512} --no-cpu-baseline --no-rows 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['kernel'], j['roofline']['avg_launch_ms'])"
done
