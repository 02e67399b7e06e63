for r in 1 2; do
  for v in "BASE=1" "SV_LIB_NAME=libsplitvae_bv.so"; do
    echo -n "step ${v}: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms']) for r in d['roofline']['table'][:12]])"
  done
done
for v in "BASE=1" "SV_LIB_NAME=libsplitvae_bv.so"; do
  echo -n "b64 ${v}: "; env $v python bench.py --steps 100 --warmup 10 --batch 64 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
