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
# bytes per training step: every launch of every kernel, divided by the number of steps run (= launches of the Adam kernel)
steps = max([len(v) for k, v in wr.items() if "adam_kernel" in k] + [1])
total = sum(2 * 1024 * sum(v) for v in rd.values()) + sum(1024 * sum(v) for v in wr.values())
print(json.dumps({"unit": "bytes per launch", "steps_profiled": steps, "total_bytes_per_step": round(total / steps), "correction": "FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE KiB x1",
                  "kernels": out}, indent=1))
