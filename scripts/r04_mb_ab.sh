# A/B of the matrix-pipe blend (row_conv.hip RowCfg::MB) on the d4 forward: layer alone (1024 images = the step's twin launch) and the step
# variants: default (one wave per SIMD, whole K per wave), SV_RC_MB_WAVES=4 (two waves per SIMD + K-half exchange), SV_RC_NO_MB=1 (VALU blend)
for r in 1 2; do
  for v in "BASE=1" "SV_RC_MB_WAVES=4" "SV_RC_NO_MB=1"; do
    echo -n "== ${v} (round $r) "
    env $v SV_BENCH_OPS=fwd python scripts/bench_layers.py 1024 d4 2>/dev/null | grep -i d4
  done
done
for r in 1 2; do
  for v in "BASE=1" "SV_RC_MB_WAVES=4" "SV_RC_NO_MB=1"; do
    echo -n "step ${v}: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms']) for r in d['roofline']['table'][:6]])"
  done
done
