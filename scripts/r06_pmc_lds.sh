#!/bin/bash
# per-kernel LDS counters of the serial per-launch table: bash scripts/r06_pmc_lds.sh <tag> [f32|bf16]   (SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones)
T=${1:-r06_x}; DT=${2:-bf16}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/_pmc && rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/_pmc -o m -- python3 $R/bench.py --dtype $DT --table-only 2 > $O/${T}_pmc_lds.log 2>&1 )
python3 - <<PY > $O/${T}_${DT}_pmc_lds.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch, alone on the chip (bench.py --dtype $DT --table-only 2).  lds_busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8) / 256 CUs; conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    g = m.get("GRBM_GUI_ACTIVE", 0)
    if g <= 0: continue
    rows.append((g, k, m))
for g, k, m in sorted(rows, reverse=True)[:45]:
    gx = g / 8.0
    la = m.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-90s cycles %9.0f  lds_busy %.3f  conflict %.3f  unaligned %.3f  insts lds %.2e valu %.2e mfma %.2e" % (k[:90], gx, la / gx / 256, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(la, 1), m.get("SQ_LDS_UNALIGNED_STALL", 0) / max(la, 1), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_MFMA", 0)))
PY
rm -rf $O/_pmc
