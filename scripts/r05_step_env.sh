#!/bin/bash
# whole-step A/B over environment settings (two interleaved rounds): bash scripts/r05_step_env.sh <batch> <dtype> "BASE=1" "SV_X=1" ...
B=$1; DT=$2; shift 2
for r in 1 2; do for v in "$@"; do
  echo -n "B=$B $DT ${v}: "; env $v python bench.py --batch $B --dtype $DT --steps 40 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(d['ms_per_step'])"
done; done
