# device tests + the default bench line (GPU box).  usage: bash scripts/r02_check.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r02b}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${T}_gputests.txt 2>&1; tail -15 $O/${T}_gputests.txt
timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_table.txt; cat $O/${T}_bench.json; tail -5 $O/${T}_table.txt
