#!/bin/bash
# phase ablation of the bf16 tile-conv launches (fwd.d5 with the fused loss, dgrad.d2, dgrad.e3 ...) on the -DSV_DEBUG_KNOBS build (WRONG results, timing only):
# bash scripts/r06_bf16_ablate.sh <tag>      SV_TC_DBG bits: 1 no staging, 2 no K loop, 4 return before the store / loss phase, 8 stale weights, 16 no per-step barrier
T=${1:-r06_a}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_bf16_ablate.txt
: > $OUT
export SV_LIB_NAME=libsplitvae_dbg.so
for d in 0 1 2 4 3 6 7 8 16; do
  echo "SV_TC_DBG=$d" >> $OUT
  SV_TC_DBG=$d timeout 200 python bench.py --dtype bf16 --table-only 10 2>&1 | grep -E "^(fwd.d5|dgrad.d2|dgrad.e3|dgrad.head|fwd.d1) " | cut -c1-100 >> $OUT
done
cat $OUT
