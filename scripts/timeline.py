"""One training step's kernels in launch order from a rocprofv3 kernel trace: start / end / duration (us), queue, grid."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from one random_perm_kernel (the augmentation's first launch) to the next (the early optimizer tail launches Adam several times)
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("random_perm_kernel")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)[:64]
    q = qs.setdefault(r["Queue_Id"], len(qs))
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    print("%8.1f %8.1f %7.1f q%d %-64s wgs=%d" % (s / 1e3, e / 1e3, (e - s) / 1e3, q, n, wgs))

if "--gaps" in sys.argv:
    # gap accounting: the step's span, kernel-busy time per queue (union of intervals), and the idle time of the chip (no kernel on any queue)
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in rows[a:b])
    busy, cur_s, cur_e = 0, None, None
    for s_, e_ in iv:
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        busy += cur_e - cur_s
    ksum = sum(e_ - s_ for s_, e_ in iv)
    print("# step span %.1f us; %d launches; sum of kernel durations %.1f us; chip busy (union) %.1f us; idle (no kernel running) %.1f us = %.1f %% of the step; "
          "mean idle per launch boundary %.2f us" % (span, len(iv), ksum / 1e3, busy / 1e3, span - busy / 1e3, 100 * (span - busy / 1e3) / span, (span - busy / 1e3) / max(len(iv), 1)))
