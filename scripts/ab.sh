# A/B of library knobs on the full bench (GPU box): one line per setting, images/s.
# usage: bash scripts/ab.sh [VAR=value ...]      e.g.  bash scripts/ab.sh SV_NO_SIDE=1 SV_TC_NO_YR=1
# (DESIGN.md section 7 lists the variables; per-layer timings: SV_BENCH_OPS=fwd,dgrad,wgrad python scripts/bench_layers.py 512 d4 d5)
run() { echo -n "$1: "; env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
run BASE=1
for kv in "$@"; do run "$kv"; done
run BASE=2
