# A/B of two builds of the library on the whole step: bash scripts/r04_ab_lib.sh <other lib name> [rows]
L=${1:-libsplitvae_old.so}; N=${2:-10}
for r in 1 2 3; do
  for v in "BASE=1" "SV_LIB_NAME=$L"; do
    echo -n "step ${v}: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (r['kernel'], r['ms']) for r in d['roofline']['table'][:$N]])"
  done
done
