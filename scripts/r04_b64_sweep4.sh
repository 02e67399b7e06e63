# strips per workgroup of the rolling-window d4 weight gradient at small batches (one slab per workgroup: fewer, longer workgroups write and reduce less)
cd $GRAFT_REPO_ROOT
for B in 64 128; do for r in 1 2; do for v in "BASE=1" "SV_ROLL_MIN_STRIPS=2" "SV_ROLL_MIN_STRIPS=4" "SV_ROLL_MIN_STRIPS=8"; do echo -n "B=$B $v: "; env $v python bench.py --batch $B --steps 300 --warmup 20 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done; done
