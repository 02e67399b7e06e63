export SV_BENCH_OPS=wgrad
L="e1 d3 d4 d5"
for rep in 1 2; do
echo "--- base"; python scripts/bench_layers.py 512 $L
echo "--- hiocc"; SV_WT_HIOCC=6 SV_WT_CW16=127 python scripts/bench_layers.py 512 $L
done
echo "=== full base"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
echo "=== full hi6"; SV_WT_HIOCC=6 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
echo "=== full cw1"; SV_WT_CW16=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
echo "=== full cw2"; SV_WT_CW16=2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
echo "=== full cw7"; SV_WT_CW16=7 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
echo "=== full all"; SV_WT_HIOCC=6 SV_WT_CW16=127 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
echo "=== full base"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-120
