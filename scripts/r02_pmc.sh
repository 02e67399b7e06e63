# PMC passes over the per-layer microbench (GPU box).  usage: bash scripts/r02_pmc.sh <tag> <ops> <layers...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift
export SV_BENCH_OPS=$1; shift
rm -rf $R/gpurun_out/pmcA $R/gpurun_out/pmcB
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcA -o a -- python3 $R/scripts/bench_layers.py 1024 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcB -o b -- python3 $R/scripts/bench_layers.py 1024 "$@" > /dev/null 2>&1
cd $R; (python scripts/pmc_summary.py gpurun_out/pmcA row_conv tile_conv; python scripts/pmc_summary.py gpurun_out/pmcB row_conv tile_conv) | tee gpurun_out/${T}_pmc.txt
