#!/bin/bash
# deferred slab reduce: parity + determinism + the two bench sizes, with and without
mkdir -p gpurun_out/defer; O=gpurun_out/defer
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for b in 512 64; do
  timeout 300 python bench.py --batch $b --no-cpu-baseline --no-rows > $O/bench_$b.json 2> $O/bench_$b.err
  SV_NO_DEFER_REDUCE=1 timeout 300 python bench.py --batch $b --no-cpu-baseline --no-rows > $O/bench_${b}_nodefer.json 2> $O/bench_${b}_nodefer.err
  python - <<PY
import json
for t in ("", "_nodefer"):
    try:
        j = json.loads(open("$O/bench_$b%s.json" % t).read().strip().splitlines()[-1]); print("$b", t, j["ms_per_step"], j["value"], j["roofline"].get("serial"))
    except Exception as e: print("$b", t, "ERR", e)
PY
done
