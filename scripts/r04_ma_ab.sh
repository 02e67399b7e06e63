# A/B of the matrix-pipe resize adjoint (row_conv.hip RowCfg::MA) on the d4 / d3 input gradients: the step's serial table rows, two rounds
for r in 1 2; do
  for v in "BASE=1" "SV_RC_NO_MA3=1" "SV_RC_NO_MA=1"; do
    echo -n "step ${v}: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms']) for r in d['roofline']['table'][:9]])"
  done
done
