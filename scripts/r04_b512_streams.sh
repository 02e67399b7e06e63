# stream assignment of the bf16 step at 512 images per network, re-measured with round 4's kernels
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "BASE=1" "SV_WGRAD_MAIN=e1,e2" "SV_WGRAD_MAIN=e1" "SV_WGRAD_MAIN=e2" "SV_WGRAD_MAIN=e1,e3" "SV_WGRAD_MAIN=e1,d1"; do echo -n "bf16 B=512 $v: "; env "$v" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done
