#!/bin/bash
# one rank through the data-parallel path (SV_DIST_FORCE=1, nccl / sv_comm), every schedule: bash scripts/r06_dp_ab.sh <tag>
T=${1:-r06_dp}; O=$GRAFT_REPO_ROOT/gpurun_out; OUT=$O/${T}_dp_ab.txt
: > $OUT
run() { local dt=$1 k=$2; shift 2; echo -n "$dt B=64 [$*]: " >> $OUT; env "$@" RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 200 python bench.py --batch 64 --dtype $dt --steps $k --warmup 10 --no-cpu-baseline --no-rows --no-other-precision 2>/dev/null | python3 -c "import sys,json; d=[json.loads(l) for l in sys.stdin if l.startswith('{')]; print(d[-1]['ms_per_step'], d[-1].get('dp_mode'), d[-1].get('exposed_allreduce_ms'), d[-1].get('allreduce_ms')) if d else print('FAILED')" >> $OUT; }
for dt in f32 bf16; do
  k=150; [ $dt = bf16 ] && k=300
  run $dt $k A=0
  for m in single events overlap; do run $dt $k SV_DIST_FORCE=1 SV_DP_MODE=$m; done
  run $dt $k SV_DIST_FORCE=1 SV_DP_MODE=events SV_DIST_BACKEND=sv_comm
  run $dt $k SV_DIST_FORCE=1 SV_DP_MODE=single SV_DIST_BACKEND=sv_comm
done
cat $OUT
