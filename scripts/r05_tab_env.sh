#!/bin/bash
# per-launch table rows under environment settings: bash scripts/r05_tab_env.sh <batch> <dtype> <grep pattern> "BASE=1" "SV_X=1" ...
B=$1; DT=$2; PAT=$3; shift 3
for v in "$@"; do
  echo "== B=$B $DT $v"
  env $v python bench.py --batch $B --dtype $DT --steps 30 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>gpurun_out/_tab.err | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('ms_per_step', d['ms_per_step'])"
  grep -E "$PAT" gpurun_out/_tab.err
done
