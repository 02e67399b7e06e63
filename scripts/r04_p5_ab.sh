# A/B of the rolling-window polyphase weight gradient of the head (wgrad_p5.hip) against the tile kernel (SV_NO_WGRAD_P5=1): serial table rows, the step
for r in 1 2; do
  for v in "BASE=1" "SV_NO_WGRAD_P5=1"; do
    echo -n "step ${v}: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); t={r['kernel']: r['ms'] for r in d['roofline']['table']}; print(d['value'], d['ms_per_step'], 'dominant', d['roofline']['kernel'], d['roofline']['serial']['frac'], 'wgrad.d5', t.get('wgrad.d5'), 'wgrad.d4', t.get('wgrad.d4'), 'decoder', d['roofline']['decoder_stack']['frac'])"
  done
done
for v in "BASE=1" "SV_NO_WGRAD_P5=1"; do echo -n "b64 ${v}: "; env $v python bench.py --steps 150 --warmup 10 --batch 64 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
for v in "BASE=1" "SV_NO_WGRAD_P5=1"; do echo -n "b256 ${v}: "; env $v python bench.py --steps 100 --warmup 10 --batch 256 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
