import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    bad = ~np.isfinite(x)
    d = np.abs(np.where(bad, 0, x) - y)
    worst = np.unravel_index(d.argmax(), d.shape)
    print("%-8s nonfinite %8d  rel-err %.3e  max|d| %.3e at %s (of max|ref| %.3e)" % (k, bad.sum(), np.linalg.norm(np.where(bad, 0, x) - y) / (np.linalg.norm(y) + 1e-30), d.max(), worst, np.abs(y).max()))
    if bad.sum():
        idx = np.argwhere(bad)
        print("   first non-finite at", idx[0], "last", idx[-1], "images", np.unique(idx[:, 0])[:10], "rows", np.unique(idx[:, 1])[:12], "cols", np.unique(idx[:, 2])[:12], "ch", np.unique(idx[:, 3])[:12])
