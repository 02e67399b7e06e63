#!/bin/bash
# step time over shard sizes (cliff hunt: a layer falling to a generic kernel shows as a bump in us per image): bash scripts/batch_sweep.sh <dtype> B1 B2 ...
DT=$1; shift
for B in "$@"; do
  python bench.py --batch $B --dtype $DT --steps 30 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>gpurun_out/_sweep.err | python -c "
import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('$DT B=%4d  %.4f ms  %.2f us/image' % ($B, d['ms_per_step'], 1e3*d['ms_per_step']/$B))"
  grep -E "^(fwd|dgrad|wgrad)" gpurun_out/_sweep.err | sort -k6 -n -r | awk '{ if ($NF+0 < 1) print }' > /dev/null
done
