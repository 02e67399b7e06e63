#!/bin/bash
# fp32 small-shard A/B: bash scripts/r05_f32_small_ab.sh <batch> "BASE=1" "SV_X=1" ...
B=$1; shift
for r in 1 2; do for v in "$@"; do
  echo -n "B=$B ${v}: "; env $v python bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(d['ms_per_step'], [(r['kernel'], r['ms']) for r in d['roofline']['table'][:8]])"
done; done
