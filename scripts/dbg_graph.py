import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SV_GRAPH", "1")
import numpy as np, torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
H, B = 32, 16
def setup():
    m = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="bf16", device=torch.device("cuda"), seed=11); m.beta = 40.0
    return m, Adam(learning_rate=1e-3)
x = data.synthetic_images(B, H, H, seed=0, device="cuda")
images = Augmentator("scramble", size=4, seed=1).augment(x)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    e, oe = setup(); g, og = setup()
    side = torch.cuda.Stream()
    for t in range(5):
        pe = trainer.train_step(e, images, oe)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pg = trainer.train_step(g, images, og)
        side.synchronize(); torch.cuda.synchronize()
        d = (g.flat - e.flat).abs()
        bad = []
        for name, off, shape in pe.param_table:
            n = int(np.prod(shape))
            c = int((d[off:off + n] > 1e-5).sum())
            if c: bad.append((name, c, n))
        gd = (g.grad_flat - e.grad_flat).abs()
        gbad = []
        for name, off, shape in pe.param_table:
            n = int(np.prod(shape))
            c = int((gd[off:off + n] > 1e-4 * float(e.grad_flat[off:off+n].abs().max())).sum())
            if c: gbad.append((name, c, n))
        if bad or gbad:
            print("rep", rep, "t", t, "weights", bad, "grads", gbad, flush=True)
            break
print("done")
