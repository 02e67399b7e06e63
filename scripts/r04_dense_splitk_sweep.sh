# K slices of the few-row Dense layers (28 MB kernels at 32 rows): workgroups in flight vs atomics, on the native SPLIT-SPAIR step
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in BASE=1 SV_DENSE_SPLIT_WGS=512 SV_DENSE_SPLIT_WGS=1024 SV_DENSE_SPLIT_WGS=2048 "SV_DENSE_SPLIT_WGS=1024 SV_DENSE_SPLIT_MIN_STEPS=4"; do echo -n "f32 $v: "; env $v python scripts/bench_spair_native.py 32 f32 2>/dev/null | tail -1; done; done
