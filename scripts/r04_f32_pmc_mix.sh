#!/bin/bash
# per-kernel instruction mix of the fp32 step (rocprofv3 --pmc over bench.py --dtype f32): MFMA-busy against launch cycles for the fp32 matrix kernels
# DESIGN 4j) -> gpurun_out/<tag>_f32_pmc_mix.txt
T=${1:-r03}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcmixf
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcmixf -o m -- python3 $R/bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-rows --no-fp32 > /dev/null 2>&1
cd $R
python3 - <<'PY' > gpurun_out/${T}_f32_pmc_mix.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcmixf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    rows.append((m.get("GRBM_GUI_ACTIVE", 0) * n, k, n, m))
rows.sort(reverse=True)
print("%-64s %5s %10s %12s %12s %12s %12s" % ("kernel", "n", "gui_active", "valu_active", "mfma_busy", "lds_active", "insts_valu"))
for _, k, n, m in rows[:40]:
    print("%-64s %5d %10.0f %12.0f %12.0f %12.0f %12.0f" % (k[:64], n, m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_ACTIVE_INST_VALU", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("SQ_ACTIVE_INST_LDS", 0), m.get("SQ_INSTS_VALU", 0)))
PY
head -30 gpurun_out/${T}_f32_pmc_mix.txt
