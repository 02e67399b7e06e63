# the tile conv's launch-shape knobs on the fp32 step (forward / input gradients)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "BASE=1" "SV_TC_NW8=1" "SV_TC_MF2=1" "SV_TC_NO_XCD=1" "SV_TC_NPH=1" "SV_TC_NO_YR=1" "SV_TC_NO_PLANAR=1"; do echo -n "f32 $v: "; env "$v" python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done
