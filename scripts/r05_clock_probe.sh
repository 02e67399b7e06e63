#!/bin/bash
# sample the shader clock and power while the fp32 step loops (is the 2.4 GHz the fp32 matrix peak assumes held under MFMA load?)
python bench.py --steps 2500 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision > gpurun_out/_clk_bench.txt 2>/dev/null &
BP=$!
sleep 9
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" ; sleep 1; done > gpurun_out/clock_probe.txt 2>&1
wait $BP
tail -c 300 gpurun_out/_clk_bench.txt | head -c 300 >> gpurun_out/clock_probe.txt
cat gpurun_out/clock_probe.txt
