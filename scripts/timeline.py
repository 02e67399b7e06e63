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
