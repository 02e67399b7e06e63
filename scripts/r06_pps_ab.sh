#!/bin/bash
# K-step size of the 32- / 64-column tile kernels (build-time -DSV_TC_PPS32=16 / -DSV_TC_PPS64=16 variants built beside the shipped library) at bf16:
# bash scripts/r06_pps_ab.sh <tag>
T=${1:-r06_k}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_pps_ab.txt
: > $OUT
run() { local lib=$1 dt=$2 b=$3 k=$4; echo -n "$lib $dt B=$b : " >> $OUT; SV_LIB_NAME=$lib timeout 200 python bench.py --batch $b --dtype $dt --steps $k --warmup 8 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
for rep in 1 2; do
for lib in libsplitvae_hip.so libsplitvae_p32.so libsplitvae_p64.so; do
  run $lib bf16 512 200; run $lib bf16 256 200; run $lib bf16 64 300
done; done
for lib in libsplitvae_hip.so libsplitvae_p32.so libsplitvae_p64.so; do
  echo "serial table, $lib:" >> $OUT
  SV_LIB_NAME=$lib python bench.py --dtype bf16 --table-only 3 2>&1 | grep -E "^(fwd.d5|dgrad.d5|dgrad.d2|dgrad.e3|dgrad.e2|fwd.e1) " | cut -c1-110 >> $OUT
done
cat $OUT
