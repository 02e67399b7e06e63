#!/bin/bash
# the latent block's fused slab sums + ring kernel: same bits as the two-launch forms, then the step with and without them
export SV_DETERMINISTIC=1
for k in X=1 SV_NO_LATENT_FUSE=1 SV_NO_NT_RING=1 "SV_NO_LATENT_FUSE=1 SV_NO_NT_RING=1"; do echo "== $k"; env $k timeout 300 python scripts/r03_step_hash.py 2>&1 | tail -5; done
unset SV_DETERMINISTIC
run() { echo -n "$1 $2  "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-rows $2 2>gpurun_out/lat_tbl_$3.txt | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])"; }
run X=1 "" a; run SV_NO_LATENT_FUSE=1 "" b; run SV_NO_NT_RING=1 "" c; run "SV_NO_LATENT_FUSE=1 SV_NO_NT_RING=1" "" d; run X=1 "" e
run X=1 "--batch 64" f; run "SV_NO_LATENT_FUSE=1" "--batch 64" g; run X=1 "--batch 64" h
grep -E "head|d1|reparam" gpurun_out/lat_tbl_a.txt gpurun_out/lat_tbl_d.txt
