#!/bin/bash
# one rank through the RCCL path at a shard size, A/B over environment settings (two rounds): bash scripts/r05_dp_ab.sh <batch> <dtype> "BASE=1" "SV_X=1" ...
B=$1; DT=$2; shift 2
for r in 1 2; do
  echo -n "B=$B $DT plain step: "; python bench.py --batch $B --dtype $DT --steps 200 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(d['ms_per_step'])"
  for v in "$@"; do
    echo -n "B=$B $DT dp one rank ${v}: "; env SV_DIST_FORCE=1 $v python bench.py --batch $B --dtype $DT --steps 200 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(d['ms_per_step'])"
  done
done
