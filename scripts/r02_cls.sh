# merged parity classes of dgrad.e2: parity + A/B.  usage: bash scripts/r02_cls.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02_cls}
cd $R
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_fwd_dgrad or merged_parity" 2>&1 | tail -5 > $O/${T}_tests.txt
for v in merged classes; do
  if [ $v = classes ]; then export SV_NO_CLS_MERGE=1; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-rows > $O/${T}_bench_$v.json 2> $O/${T}_bench_$v.err
done
cat $O/${T}_tests.txt
for v in merged classes; do echo "== $v"; grep -o '"ms_per_step": [0-9.]*' $O/${T}_bench_$v.json; grep "dgrad.e[23]\|dgrad.d[2345]" $O/${T}_bench_$v.err; done
