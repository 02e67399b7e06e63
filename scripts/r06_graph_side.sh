#!/bin/bash
T=${1:-r06_g}; O=$GRAFT_REPO_ROOT/gpurun_out
for dt in bf16 f32; do for b in 64 128; do for m in eager graph graph_side eager graph_side; do timeout 120 python scripts/r06_graph_side.py $m $dt $b 300 2>&1 | tail -1; done; done; done > $O/${T}_graph_side.txt 2>&1
cat $O/${T}_graph_side.txt
