#!/bin/bash
# ablation of the row-ring kernel's phases on one layer's input gradient / forward (debug-knob build, wrong results): SV_RC_DBG bits
# 1 skip staging, 2 skip the MFMA loop, 4 skip the stores / the adjoint epilogue, 8 skip the K-half exchange
# usage: bash scripts/r03_abl_rowconv.sh <op: dgrad|fwd> <layers...>      (build first: SV_LIB_NAME=libsplitvae_dbg.so SV_OBJ_TAG=_dbg SV_EXTRA_FLAGS=-DSV_DEBUG_KNOBS python split_vae_amd/build.py)
OP=${1:-dgrad}; shift
export SV_LIB_NAME=libsplitvae_dbg.so SV_BENCH_OPS=$OP
for d in 0 1 2 4 3 5 6 7; do echo -n "dbg=$d "; SV_RC_DBG=$d timeout 120 python scripts/bench_layers.py 1024 ${@:-d5} 2>&1 | grep -v amdgpu | tr '\n' ' '; echo; done
