import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
model = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=dev, seed=3); model.beta = 120.0
opt = Adam(learning_rate=1e-4); aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(B, 64, 64, seed=100, device=dev)
out = []
for step in range(14):
    plan = trainer.train_step(model, aug.augment(x), opt)
    torch.cuda.synchronize()
    out.append(float(plan.buffer("losses", torch.float32, (8,))[5]))
    if step in (0, 13):
        g = model.grad_flat
        bad = [(n, int(torch.isnan(g[o:o + 1]).sum())) for n, o, s in model.param_table if not torch.isfinite(g[o:o + int(torch.tensor(s).prod())]).all()]
        print("step", step, "non-finite grads:", bad[:6])
print(os.environ.get("TAG", ""), [round(v, 1) for v in out])
