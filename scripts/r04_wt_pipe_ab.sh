# the pipelined tile weight gradient (whole-image tiles: d2, e3): parity, per-layer and whole-step A/B against the previous build (libsplitvae_old.so) and
# against the non-pipelined form of the same build (SV_WT_NO_PIPE)      -> gpurun_out/<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04_wt_pipe_ab}
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -q -x -k "wgrad or step or conv" 2>&1 | tail -3
{
for B in 1024 128; do
  for v in "BASE=1" "SV_WT_NO_PIPE=1" "SV_WT_PIPE_NBUF=2" "SV_WT_PIPE_NBUF=3" "SV_LIB_NAME=libsplitvae_old.so"; do echo -n "$v: "; env $v SV_BENCH_OPS=wgrad python scripts/bench_layers.py $B d2 e3 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done; done
bash scripts/r04_ab_lib.sh libsplitvae_old.so 12
for r in 1 2; do for v in "BASE=1" "SV_WT_NO_PIPE=1"; do echo -n "step $v: "; env $v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
for r in 1 2; do for l in hip old; do echo -n "B=64 lib=$l: "; SV_LIB_NAME=libsplitvae_$l.so python bench.py --batch 64 --steps 200 --warmup 20 --no-cpu-baseline --no-rows --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
} 2>&1 | tee $O/${T}.txt
