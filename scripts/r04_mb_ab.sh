# A/B of the matrix-pipe blend (row_conv.hip RowCfg::MB) on the d4 / d3 forward: the step's serial table rows
for r in 1 2; do
  for v in "BASE=1" "SV_RC_NO_MB3=1" "SV_RC_NO_MB=1"; do
    echo -n "step ${v}: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); t={r['kernel']: r['ms'] for r in d['roofline']['table']}; print(d['value'], d['ms_per_step'], 'fwd.d4', t.get('fwd.d4'), 'fwd.d3', t.get('fwd.d3'), 'dgrad.d3', t.get('dgrad.d3'))"
  done
done
