"""sha256 of five training steps (losses, gradients, weights) at B = 64 and B = 512, bf16 CelebA-64: run under different knobs
(SV_NO_LATENT_FUSE, SV_NO_NT_RING, ...) the hashes must agree -- the fused slab sums and the ring kernel keep the summation order."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam


def run(dtype, H, B, patch):
    x = data.synthetic_images(B, H, H, seed=0, device="cuda")
    img = Augmentator("scramble", size=patch, seed=1).augment(x)
    m = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dtype, device=torch.device("cuda"), seed=3)
    m.beta = 120.0
    opt = Adam(learning_rate=1e-4)
    h = hashlib.sha256()
    for i in range(5):
        plan = trainer.train_step(m, img, opt)
        torch.cuda.synchronize()
        for t in (plan.buffer("losses", torch.float32, (8,)), m.grad_flat, m.flat):
            h.update(t.detach().cpu().numpy().tobytes())
        if i == 2:                       # the weights change behind the library's back: images prepared ahead must not be used
            m.flat.mul_(1.001)
        if i == 3:                       # an evaluation call between two training steps
            h.update(repr(sorted(trainer.test_step(m, img).items())).encode())
    return h.hexdigest()


if __name__ == "__main__":
    for name, a in (("bf16 64x64 B=64", ("bf16", 64, 64, 8)), ("bf16 64x64 B=512", ("bf16", 64, 512, 8)), ("bf16 32x32 B=70", ("bf16", 32, 70, 4)),
                    ("f32 32x32 B=16", ("f32", 32, 16, 1))):
        print(name, run(*a)[:24], flush=True)
