#!/bin/bash
# per-kernel counter mix of the serial per-launch table (every launch alone on the chip): bash scripts/r06_pmc_mix.sh <tag> [f32|bf16]
T=${1:-r06_x}; DT=${2:-f32}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_pmc && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/_pmc -o m -- python3 $R/bench.py --dtype $DT --table-only 2 > /dev/null 2>&1 )
python3 - <<PY > $O/${T}_${DT}_pmc_mix.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch, alone on the chip (bench.py --dtype $DT --table-only 2).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8) / 1024: the busy cycles are summed over the chip's")
print("# 1024 SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 / wall time) -- round 5's file divided by the")
print("# un-normalised GUI count and read 8x too low.  = the fraction of the launch's cycles a SIMD's matrix pipe was busy: the figure roofline.*_issued is an estimate of")
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    g = m.get("GRBM_GUI_ACTIVE", 0)
    if g <= 0: continue
    rows.append((g, k, m))
for g, k, m in sorted(rows, reverse=True)[:40]:
    gx = g / 8.0
    print("%-90s cycles %9.0f  mfma_util %.3f  wait_lds/wave %.3f  valu/wave %.3f  wait_any/wave %.3f" % (k[:90], gx, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / gx / 1024, m.get("SQ_WAIT_INST_LDS", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
rm -rf $O/_pmc
