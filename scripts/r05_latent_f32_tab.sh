#!/bin/bash
for B in 512 256; do for v in "BASE=1" "SV_NT_F32_BM=64"; do
  echo "== B=$B $v"; env $v python bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>&1 >/dev/null | grep -E "head|\.d1 "
done; done > gpurun_out/latent_f32_tab.txt 2>&1
cat gpurun_out/latent_f32_tab.txt
