# tile size / workgroup count of the fp32 LDS-tile weight gradient on the native SPLIT-SPAIR step
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in BASE=1 SV_WTF32_BM=128 SV_WTF32_BM=64 SV_WTF32_WGS=256 SV_WTF32_WGS=1024; do echo -n "f32 $v: "; env $v python scripts/bench_spair_native.py 32 f32 2>/dev/null | tail -1; done; done
