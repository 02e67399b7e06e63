#!/bin/bash
# round-3 final evidence set: bash scripts/r03_final.sh <tag>
T=${1:-r03_k}
bash scripts/r02_profile.sh $T > gpurun_out/${T}_profile.log 2>&1
timeout 300 python bench.py --batch 64 --no-cpu-baseline --no-rows > gpurun_out/${T}_bench_b64.json 2> gpurun_out/${T}_table_b64.txt
timeout 400 bash scripts/r02_timeline.sh ${T}_tl > /dev/null 2>&1; cp gpurun_out/${T}_tl_timeline.txt gpurun_out/${T}_step_timeline.txt
timeout 400 bash scripts/r03_spair_prof.sh $T > /dev/null 2>&1
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputests_$i.log 2>&1; grep -E "passed|failed" gpurun_out/${T}_gputests_$i.log | tail -1; grep -E "^FAILED" gpurun_out/${T}_gputests_$i.log; done
tail -3 gpurun_out/${T}_profile.log
