#!/bin/bash
# one against two weight-gradient side streams at every shard size, SVHN-32, SPLIT-GMVAE and the one-rank RCCL path
for b in 256 128 64; do for s in 1 2; do echo -n "B=$b side=$s  "; SV_SIDE_STREAMS=$s BATCH=$b bash scripts/r03_main_sweep.sh e1,e2 | cut -d' ' -f3-5; done; done
for s in 1 2; do echo -n "svhn32 B=64 side=$s  "; SV_SIDE_STREAMS=$s timeout 200 python bench.py --size 32 --batch 64 --no-cpu-baseline --no-rows 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'])"; done
for s in 1 2; do echo -n "gm side=$s  "; SV_SIDE_STREAMS=$s timeout 200 python scripts/bench_gm.py 2>/dev/null | tail -2 | tr '\n' ' '; echo; done
for s in 1 2; do for be in nccl sv_comm; do echo -n "one-rank $be side=$s  "; SV_SIDE_STREAMS=$s SV_DIST_FORCE=1 SV_DIST_BACKEND=$be MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --no-cpu-baseline --no-rows 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j.get('rccl_ranks'))"; done; done
