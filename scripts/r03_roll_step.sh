#!/bin/bash
mkdir -p gpurun_out/roll; O=gpurun_out/roll
for b in 64 128 256; do
for v in roll tile; do
  if [ $v = tile ]; then export SV_NO_WGRAD_ROLL=1; else unset SV_NO_WGRAD_ROLL; fi
  timeout 300 python bench.py --batch $b --no-cpu-baseline --no-rows > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
j = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1]); print("$b $v", j["ms_per_step"], j["value"])
PY
  grep -E "wgrad.d4 " $O/bench_$v.err
done; done
