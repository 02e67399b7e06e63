#!/bin/bash
# d4's fp32 weight gradient with the input tile staged once per class PAIR (SV_WGRAD_POLYC_FUSED=2) against the four class launches (0) and the four-class launch (1):
# parity tests under the mode, the step at four shard sizes, the serial wgrad.d4 scope and its HBM traffic.   bash scripts/r06_polyc_pair.sh <tag>
T=${1:-r06_p}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; OUT=$O/${T}_polyc_pair.txt
: > $OUT
SV_WGRAD_POLYC_FUSED=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_fullsize.py -m gpu -x -q -k "f32 or fp32 or polyphase or oracle" 2>&1 | tail -3 >> $OUT
run() { local b=$1 k=$2; shift 2; echo -n "f32 B=$b $* : " >> $OUT; env "$@" timeout 200 python bench.py --batch $b --dtype f32 --steps $k --warmup 8 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'] if d else 'FAILED')" >> $OUT; }
for rep in 1 2; do
for cfg in "512 60" "256 100" "128 150" "64 200"; do set -- $cfg
for m in 0 2 1; do run $1 $2 SV_WGRAD_POLYC_FUSED=$m; done
done; done
for m in 0 2 1; do
  echo "serial table, SV_WGRAD_POLYC_FUSED=$m:" >> $OUT
  SV_WGRAD_POLYC_FUSED=$m python bench.py --dtype f32 --table-only 3 2>&1 | grep -E "wgrad.d4 |dgrad.d4 |wgrad.d3 " | cut -c1-150 >> $OUT
done
traffic() {
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_tR $O/_tW
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/_tR -o r -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-other-precision > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/_tW -o w -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-other-precision > /dev/null 2>&1 )
  python3 scripts/traffic_summary.py $O/_tR $O/_tW > $1
  rm -rf $O/_tR $O/_tW
}
export SV_WGRAD_POLYC_FUSED=2
traffic $O/${T}_pair_f32_traffic.json
python3 - <<PY >> $OUT
import json
d = json.load(open("$O/${T}_pair_f32_traffic.json"))
print("HBM traffic per launch, SV_WGRAD_POLYC_FUSED=2:")
for n, k in sorted(d["kernels"].items(), key=lambda x: -x[1].get("hbm_bytes_per_launch", 0)):
    if "wgrad_polyc" in n or "polyc_wgrad" in n or "wgrad_tile_f32_kernel<7" in n or "wgrad_tile_f32_kernel<5" in n or "wgrad_tile_f32_kernel<4, 2" in n:
        print("  %-100s %8.1f MB" % (n.replace("(anonymous namespace)::", "")[:100], k["hbm_bytes_per_launch"] / 1e6))
print("  step total %.2f GB" % (d.get("step_bytes", 0) / 1e9) if "step_bytes" in d else "")
PY
cat $OUT
