#!/bin/bash
# round 6: the whole GPU suite + the bench line + the small-shard rows + one-step timelines: bash scripts/r06_check.sh <tag> [notests]
T=${1:-r06_b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
if [ "$2" != "notests" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1; grep -E "passed|failed" $O/${T}_tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/${T}_tests.log | head; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_table.txt; python3 - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print("f32 B=512", d["ms_per_step"], d["value"], {k: d["roofline"].get(k) for k in ("serial_frac_direct", "serial_frac_issued", "decoder_stack_frac_direct", "decoder_stack_frac_issued", "step_frac_direct", "step_frac_issued")})
print("bf16", d["bf16"]["ms_per_step"], d["bf16"]["value"])
for k, v in d["rows"].items():
    print(k, v.get("ms_per_step"), v.get("value"), {q: (v[q].get("ms_per_step"), v[q].get("vs_plain_step")) for q in ("f32", "bf16") if isinstance(v.get(q), dict)})
PY
bash scripts/r05_timeline.sh $T f32 64 > /dev/null
bash scripts/r05_timeline.sh $T f32 512 > /dev/null
tail -1 $O/${T}_f32_b64_timeline.txt; tail -1 $O/${T}_f32_b512_timeline.txt
