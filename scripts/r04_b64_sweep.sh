# B = 64 (config 4's shard): thresholds re-measured with the round-4 kernels
run() { echo -n "$1: "; env $1 python bench.py --steps 150 --warmup 10 --batch ${B:-64} --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run BASE=1
run SV_RC_ADJ_MIN=128
run SV_RC_ADJ_MIN=64
run SV_SIDE_STREAMS=2
run BASE=2
B=128 run BASE=1
B=128 run SV_RC_ADJ_MIN=128
B=256 run BASE=1
