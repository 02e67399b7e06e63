"""Does anything read plan workspace it never wrote?  The C ABI takes a caller-owned workspace without asking for zeros (include/splitvae.h: sv_lgvae_plan_bind); the Python
mirror happens to allocate zeros.  Here the workspace is 0xFF bytes (NaN patterns at every precision) before the bind; five training steps must hash exactly as with zeros."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from split_vae_amd import data, trainer, ops
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam

POISON = [False]
_zeros = torch.zeros
def zeros_or_poison(*a, **k):
    if POISON[0] and k.get("dtype") == torch.uint8:
        return torch.full(*a, 255, **k) if not isinstance(a[0], tuple) else torch.full(a[0], 255, **k)
    return _zeros(*a, **k)

def run(dtype, H, B, patch, poison):
    POISON[0] = poison
    torch.zeros = zeros_or_poison
    try:
        x = data.synthetic_images(B, H, H, seed=0, device="cuda")
        img = Augmentator("scramble", size=patch, seed=1).augment(x)
        m = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dtype, device=torch.device("cuda"), seed=3)
        m.beta = 120.0
        opt = Adam(learning_rate=1e-4)
        h = hashlib.sha256()
        for i in range(3):
            plan = trainer.train_step(m, img, opt)
            torch.cuda.synchronize()
            for t in (plan.buffer("losses", torch.float32, (8,)), m.grad_flat, m.flat):
                h.update(t.detach().cpu().numpy().tobytes())
            if i == 0:
                nan = bool(torch.isnan(m.grad_flat).any()) or bool(torch.isnan(plan.buffer("losses", torch.float32, (8,))).any())
        return h.hexdigest()[:20], nan
    finally:
        torch.zeros = _zeros
        POISON[0] = False

def run_gm(dtype, poison):
    from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
    POISON[0] = poison
    torch.zeros = zeros_or_poison
    try:
        x = data.synthetic_images(64, 32, 32, seed=0, device="cuda")
        img = Augmentator("scramble", size=4, seed=1).augment(x)
        m = LGGMVae(128, 128, [-1, 32, 32, 3], 30, 0.4, dtype=dtype, device="cuda", seed=3)
        m.beta, m.alpha = 40.0, 40.0
        opt = Adam(learning_rate=1e-4)
        h = hashlib.sha256()
        nan = False
        for _ in range(3):
            train_step_lg_gm_vae(m, img, opt)
            torch.cuda.synchronize()
            for g in m.gradients:
                h.update(g.detach().cpu().numpy().tobytes()); nan = nan or bool(torch.isnan(g).any())
            for v in m.trainable_variables:
                h.update(v.detach().cpu().numpy().tobytes())
        return h.hexdigest()[:20], nan
    finally:
        torch.zeros = _zeros
        POISON[0] = False


if __name__ == "__main__":
    bad = 0
    from split_vae_amd import _lib
    _lib.load().sv_set_deterministic(1)                  # (SPLIT-GMVAE's dense layers use fp32 atomics on the default path: fixed order, so the hashes can be compared)
    for name, dt in (("lggmvae f32", "f32"), ("lggmvae bf16", "bf16")):
        z, _ = run_gm(dt, False)
        p, nan = run_gm(dt, True)
        print("%-20s zeros %s  poisoned %s  %s%s" % (name, z, p, "SAME" if z == p else "DIFFERENT", "  (NaN)" if nan else ""), flush=True)
        bad += z != p
    for name, a in (("f32 64x64 B=64", ("f32", 64, 64, 8)), ("f32 64x64 B=512", ("f32", 64, 512, 8)), ("f32 32x32 B=64", ("f32", 32, 64, 4)), ("f32 64x64 B=5", ("f32", 64, 5, 8)),
                    ("bf16 64x64 B=64", ("bf16", 64, 64, 8)), ("bf16 64x64 B=512", ("bf16", 64, 512, 8)), ("bf16 32x32 B=70", ("bf16", 32, 70, 4))):
        z, _ = run(*a, poison=False)
        p, nan = run(*a, poison=True)
        print("%-20s zeros %s  poisoned %s  %s%s" % (name, z, p, "SAME" if z == p else "DIFFERENT", "  (NaN in the first step)" if nan else ""), flush=True)
        bad += z != p
    sys.exit(1 if bad else 0)
