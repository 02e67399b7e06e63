# HBM traffic of the bench's kernels from the L2 memory-side counters (GPU box), as MI355X_MICROARCH.md
# "HBM [CDNA4]" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one),
# kernel-trace only.  Summarised by scripts/traffic_summary.py into gpurun_out/traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/trafR $R/gpurun_out/trafW
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/trafR -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/trafW -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R && python3 scripts/traffic_summary.py gpurun_out/trafR gpurun_out/trafW > gpurun_out/traffic.json && cat gpurun_out/traffic.json | head -60
