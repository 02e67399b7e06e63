#!/bin/bash
# PMC passes over the d4 weight gradient alone (rolling-window kernel)
mkdir -p gpurun_out/roll; O=$GRAFT_REPO_ROOT/gpurun_out/roll
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_upsample" > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SV_BENCH_OPS=wgrad
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmcA -o a -- python3 $R/scripts/bench_layers.py 1024 d4 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmcB -o b -- python3 $R/scripts/bench_layers.py 1024 d4 > /dev/null 2>&1
cd $R; python scripts/pmc_summary.py gpurun_out/roll/pmcA wgrad_roll; python scripts/pmc_summary.py gpurun_out/roll/pmcB wgrad_roll
