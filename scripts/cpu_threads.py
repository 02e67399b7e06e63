import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import np_ref, torch_ref
H, B = 64, 128
params = np_ref.glorot_init(H, H)
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.integers(0, 256, (B, H, H, 6)) / 255 * 2 - 1).astype(np.float32))
eps = torch.randn(2, B, 128)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    tr = torch_ref.RefTrainer(params, 120.0, dtype=torch.float32)
    tr.train_step(img, eps[0], eps[1])
    t = time.time(); n = 0
    while time.time() - t < 4: tr.train_step(img, eps[0], eps[1]); n += 1
    print("threads %d: %.1f img/s" % (nt, B * n / (time.time() - t)), flush=True)
