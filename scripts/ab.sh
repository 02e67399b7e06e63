run() { echo -n "$1: "; env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
run SV_TC_NO_WRES=1
run A=1
run SV_TC_NO_WRES=1
run A=1
export SV_BENCH_OPS=fwd,dgrad
echo "--- no WRES"; SV_TC_NO_WRES=1 python scripts/bench_layers.py 512 e1 d5 2>&1 | grep -v amdgpu
echo "--- WRES"; python scripts/bench_layers.py 512 e1 d5 2>&1 | grep -v amdgpu
