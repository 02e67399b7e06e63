#!/bin/bash
# fp32 latent block on latent_gemm.hip: parity tests, then A/B at B=512 and B=64 (SV_NO_LATENT_GEMM_F32=1 = the tap_gemm / im2col launches)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/latent_f32_tests.txt
pick='import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith("{")][-1]; print(d["ms_per_step"], [(r["kernel"], r["ms"]) for r in d["roofline"]["table"] if r["kernel"].split(".")[-1] in ("head","d1")])'
for r in 1 2; do for v in "BASE=1" "SV_NO_LATENT_GEMM_F32=1"; do
  echo -n "B=512 ${v}: "; env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "$pick"
  echo -n "B=64 ${v}: "; env $v python bench.py --batch 64 --steps 100 --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python -c "$pick"
done; done > gpurun_out/latent_f32_ab.txt 2>&1
cat gpurun_out/latent_f32_tests.txt gpurun_out/latent_f32_ab.txt
