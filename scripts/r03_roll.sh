#!/bin/bash
# rolling-window d4 weight gradient: parity, then the layer alone with and without
mkdir -p gpurun_out/roll; O=gpurun_out/roll
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_upsample or fwd_dgrad_wgrad" > $O/tests.log 2>&1; grep -E "passed|failed|Error|error" $O/tests.log | tail -5
for b in 512 1024 64; do
  SV_BENCH_OPS=wgrad timeout 120 python scripts/bench_layers.py $b d4 2>&1 | grep d4
  SV_NO_WGRAD_ROLL=1 SV_BENCH_OPS=wgrad timeout 120 python scripts/bench_layers.py $b d4 2>&1 | grep d4 | sed 's/^/tile: /'
done
