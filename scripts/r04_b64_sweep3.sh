# which form the 8 x 8-grid layers take at small batches: e3's forward on the tile kernel (SV_RC_NO_E3), d2's input gradient on the image-pair row kernel
# (SV_RC_PAIR_DGRAD)   -> gpurun_out/r04_b64_sweep3.txt
cd $GRAFT_REPO_ROOT
for B in 64 128 256; do for r in 1 2; do for v in "BASE=1" "SV_RC_NO_E3=1" "SV_RC_PAIR_DGRAD=1" "SV_RC_NO_E3=1 SV_RC_PAIR_DGRAD=1"; do echo -n "B=$B $v: "; env $v python bench.py --batch $B --steps 300 --warmup 20 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done; done
