"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads (64 B tallied per 128-B request), so the
read figure is doubled; WRITE_SIZE is exact for 16-B-per-lane stores and fp32 atomics."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


rd, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(rd) | set(wr)):
    r = rd.get(k, [0.0]); w = wr.get(k, [0.0])
    out[k[:120]] = {"launches_sampled": len(r), "read_bytes_per_launch": round(2 * 1024 * sum(r) / len(r)),
                    "write_bytes_per_launch": round(1024 * sum(w) / len(w)),
                    "hbm_bytes_per_launch": round(2 * 1024 * sum(r) / len(r) + 1024 * sum(w) / len(w))}
# bytes per training step: every launch of every kernel that runs at least once per step, divided by the number of steps run (= launches
# of the Adam kernel).  Kernels launched fewer times than there are steps are the process's SET-UP (torch zero-filling the 1.6 GB
# workspace and the parameter / moment buffers once): they are listed, not charged to the step (rounds 1-3 charged them: +0.21 GB per
# step at 12 profiled steps, `setup_bytes_per_step_if_charged`).
steps = max([len(v) for k, v in wr.items() if "adam_kernel" in k] + [1])
setup = sorted(k for k in set(rd) | set(wr) if max(len(rd.get(k, [])), len(wr.get(k, []))) < steps)
total = sum(2 * 1024 * sum(v) for k, v in rd.items() if k not in setup) + sum(1024 * sum(v) for k, v in wr.items() if k not in setup)
setup_total = sum(2 * 1024 * sum(v) for k, v in rd.items() if k in setup) + sum(1024 * sum(v) for k, v in wr.items() if k in setup)
print(json.dumps({"unit": "bytes per launch", "steps_profiled": steps, "total_bytes_per_step": round(total / steps),
                  "setup_kernels_not_charged": [k[:80] for k in setup], "setup_bytes_per_step_if_charged": round(setup_total / steps),
                  "correction": "FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE KiB x1",
                  "kernels": out}, indent=1))
