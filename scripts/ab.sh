run() { echo -n "$1: "; env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
run SV_WT_NO_PAD16=1
run A=1
run SV_WT_NO_PAD16=1
run A=1
run SV_WT_NO_PAD16=1
run A=1
