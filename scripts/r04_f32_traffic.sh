#!/bin/bash
# HBM bytes per launch of the fp32 (reference-precision) step: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (no trace domains
# beside --kernel-trace), summed per kernel by scripts/traffic_summary.py with the gfx950 read correction.   usage: bash scripts/r04_f32_traffic.sh <tag>
T=${1:-r04_f32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trafR32 $O/trafW32
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/trafR32 -o r -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/trafW32 -o w -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
cd $R
python3 scripts/traffic_summary.py $O/trafR32 $O/trafW32 > $O/${T}_f32_traffic.json
python3 -c "
import json; d=json.load(open('$O/${T}_f32_traffic.json')); print('fp32 step: total traffic per step (GB):', d.get('total_bytes_per_step', 0)/1e9)
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:14]: print('%-110s %8.1f MB' % (k[:110], v['hbm_bytes_per_launch'] / 1e6))"
rm -rf $O/trafR32 $O/trafW32
