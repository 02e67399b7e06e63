import sys, numpy as np
a = np.load(sys.argv[1]); 
for f in sys.argv[2:]:
    b = np.load(f); d = np.abs(a - b)
    print(f, "max abs diff", d.max(), "n diff", int((d > 0).sum()), "rel", float(np.linalg.norm(a - b) / np.linalg.norm(a)))
