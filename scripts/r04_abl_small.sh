# phase ablation of the row-ring kernel on the small-grid layers (debug-knob build shipped with the snapshot)   -> gpurun_out/r04_abl_small.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
{ echo "#### dgrad e2 e3 d2 d3"; bash scripts/r03_abl_rowconv.sh dgrad e2 e3 d2 d3; echo "#### fwd e2 e3 d2 d3 e1"; bash scripts/r03_abl_rowconv.sh fwd e2 e3 d2 d3 e1; } 2>&1 | tee $O/r04_abl_small.txt
