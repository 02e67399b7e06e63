mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
timeout 900 python -m pytest -x -q tests/test_gpu_step.py tests/test_gpu_graph.py tests/test_abi.py -m gpu > $O/t_step.txt 2>&1; tail -3 $O/t_step.txt
for s in 0 1 2 3; do SV_GM_TEST_SEED=$s timeout 600 python -m pytest -q tests/test_gpu_gm.py -k "test_gm_step_fp32_matches_oracle" > $O/t_gm_seed$s.txt 2>&1; echo "gm seed $s: $(tail -1 $O/t_gm_seed$s.txt)"; done
timeout 600 python -m pytest -q tests/test_gpu_gm.py -k "default_summation" > $O/t_gm_default.txt 2>&1; tail -1 $O/t_gm_default.txt
timeout 900 python -m pytest -q tests/test_gpu_dist.py -k "one_rank" > $O/t_dist.txt 2>&1; tail -1 $O/t_dist.txt
timeout 600 python bench.py --steps 40 --warmup 5 --no-rows > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['dtype']); f=d['fp32']; print('fp32', f['value'], f['ms_per_step'], f['roofline']['kernel'], f['roofline']['frac'], f['roofline']['serial']['frac'], f['roofline']['decoder_stack']['frac'])"
