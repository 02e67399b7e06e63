#!/bin/bash
# fp32 step A/B over environment settings: bash scripts/r05_f32_ab.sh "BASE=1" "SV_X=1" ...   (two interleaved rounds; ms per step + the decoder rows of the serial table)
for r in 1 2; do for v in "$@"; do
  echo -n "step ${v}: "; env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print(d['ms_per_step'], d['roofline']['decoder_stack']['frac'], [(r['kernel'], r['ms']) for r in d['roofline']['table'][:12] if r['kernel'].split('.')[1] in ('d4','d5')])"
done; done
