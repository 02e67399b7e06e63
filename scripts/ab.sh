export SV_BENCH_OPS=fwd,dgrad
L="d3 d4 d5"
for rep in 1 2; do
echo "--- no YR"; SV_TC_NO_YR=1 python scripts/bench_layers.py 512 $L 2>&1 | grep -v amdgpu
echo "--- YR"; python scripts/bench_layers.py 512 $L 2>&1 | grep -v amdgpu
done
