# from how many images per launch does the polyphase weight gradient of the head pay, with its main term on wgrad_p5.hip?  (default SV_POLY_WGRAD_MIN=768)
run() { echo -n "B=$B $1: "; env $1 python bench.py --steps 120 --warmup 10 --batch $B --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for B in 256 128 64; do
  run BASE=1; run SV_POLY_WGRAD_MIN=128; run BASE=2
done
