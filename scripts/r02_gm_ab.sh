# SPLIT-GMVAE step A/B over env knobs.  usage: bash scripts/r02_gm_ab.sh <tag> "<ENV=1 ...>" ...   (pass "" for the defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
cd $R
i=0
for v in "$@"; do
  echo "== [$i] $v"
  env GM_NO_CPU=1 GM_DTYPES=bf16 $v python scripts/bench_gm.py ${GMB:-64 512} 2>&1 | grep "train step" | grep MI355X
  i=$((i+1))
done
