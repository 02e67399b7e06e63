export SV_BENCH_OPS=fwd
for rep in 1 2; do
echo "--- base"; python scripts/bench_layers.py 512 d5 e1 e2 e3 2>&1 | grep -v amdgpu
echo "--- NO_S2PAD"; SV_TC_NO_S2PAD=1 python scripts/bench_layers.py 512 d5 e1 e2 e3 2>&1 | grep -v amdgpu
done
SV_TC_NO_S2PAD=1 SV_TC_VERBOSE=1 python scripts/bench_layers.py 512 d5 2>&1 | grep "plan" | sort | uniq -c
