#!/bin/bash
# fp32 weight-gradient tile knobs (wgrad_tile_f32.hip): serial rows of the wgrad launches + the step, per setting
for v in "BASE=1" "SV_WTF32_BM=128" "SV_WTF32_WGS=768" "SV_WTF32_WGS=1024" "SV_WTF32_BM=128 SV_WTF32_WGS=1024" "SV_WTF32_LDS=52000"; do
  echo -n "${v}: "; env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>&1 >/tmp/o.json | grep "^wgrad\.[de][1-5] " | awk '{printf "%s %s  ", $1, $5}'; python -c "import sys,json; d=[json.loads(l) for l in open('/tmp/o.json') if l.startswith('{')][-1]; print(' step', d['ms_per_step'])"
done
