# generic env-knob A/B of the step.  usage: bash scripts/r02_ab.sh <tag> "<ENV=1 ...>" ["<ENV2=...>" ...]   (first variant = defaults: pass "")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
cd $R
i=0
for v in "$@"; do
  env $v python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-rows ${BARGS} > $O/${T}_bench_$i.json 2> $O/${T}_bench_$i.err
  echo "== [$i] $v: $(grep -o '"ms_per_step": [0-9.]*' $O/${T}_bench_$i.json | head -1)"
  grep "${GREP:-wgrad}" $O/${T}_bench_$i.err
  i=$((i+1))
done
