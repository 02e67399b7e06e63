import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
B = int(sys.argv[1])
dev = torch.device("cuda")
model = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="bf16", device=dev, seed=3); model.beta = 120.0
opt = Adam(learning_rate=1e-4); aug = Augmentator("scramble", size=8, seed=1)
x = data.synthetic_images(B, 64, 64, seed=100, device=dev)
for step in range(3):
    plan = trainer.train_step(model, aug.augment(x), opt)
    torch.cuda.synchronize()
    f32 = ("pre_", "gz_", "z_mean_", "z_sig_", "z_", "eps_", "kl_", "out6_", "nll_")
    bf = ("in8_", "a1_", "a2_", "a3_", "ghead_", "ga3_", "ga2_", "ga1_", "h1_", "h2_", "h3_", "h4_", "g5_", "gu4_", "g4_", "gu3_", "g3_", "gu2_", "g2_", "g1_")
    rep = []
    for k in f32 + bf:
        for sfx in ("x", "xh"):
            try:
                t = plan.buffer(k + sfx, torch.float32 if k in f32 else torch.bfloat16, (-1,))
            except Exception as e:
                continue
            n = int((~torch.isfinite(t.float())).sum())
            if n:
                rep.append((k + sfx, n, int(t.numel())))
    g = model.grad_flat
    gb = [n for n, o, s in model.param_table if not torch.isfinite(g[o:o + int(np.prod(s))]).all()]
    print("step", step, "loss", float(plan.buffer("losses", torch.float32, (8,))[5]), "non-finite buffers:", rep, "grads:", gb[:8], flush=True)
