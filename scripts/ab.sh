export SV_BENCH_OPS=fwd,dgrad
L="d4 d3 d2 e2 e3"
for rep in 1 2; do
echo "--- NPH=1"; python scripts/bench_layers.py 512 $L
echo "--- NPH=2"; SV_TC_NPH=2 python scripts/bench_layers.py 512 $L
echo "--- NPH=4"; SV_TC_NPH=4 python scripts/bench_layers.py 512 $L
done
