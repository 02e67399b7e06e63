"""SPLIT-GMVAE (config 3: SVHN-32, y_size 30, tau 0.4, beta 40, alpha 40, patch 4) train step and the evaluation /
inference surface on one MI355X, with the torch-CPU restatement (oracle/gm_ref.py, fp32) timed beside it.
Usage: python scripts/bench_gm.py [batch ...]      (prints one JSON line per measurement)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae, test_step_lg_gm_vae as eval_step_gm
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam

H, PATCH, K, TAU = 32, 4, 30, 0.4


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def gpu(B, dtype):
    x = data.synthetic_images(B, H, H, seed=0, device="cuda")
    aug = Augmentator("scramble", size=PATCH, seed=1)
    m = LGGMVae(128, 128, [-1, H, H, 3], K, TAU, dtype=dtype, device="cuda", seed=3)
    m.beta, m.alpha = 40.0, 40.0
    opt = Adam(learning_rate=1e-4)
    dt = timed(lambda: train_step_lg_gm_vae(m, aug.augment(x), opt), 60)
    print(json.dumps({"what": "LGGMVae train step (scramble+fwd+losses+bwd+Adam)", "device": "MI355X", "dtype": dtype, "batch": B,
                      "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}), flush=True)
    img = aug.augment(x)
    dt = timed(lambda: eval_step_gm(m, img), 60)
    print(json.dumps({"what": "LGGMVae test step (fwd+losses, training=False)", "device": "MI355X", "dtype": dtype, "batch": B,
                      "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}), flush=True)
    v = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dtype, device=torch.device("cuda"), seed=3)
    v.beta = 40.0
    dt = timed(lambda: trainer.test_step(v, img), 60)
    print(json.dumps({"what": "LGVae test step (fwd+losses)", "device": "MI355X", "dtype": dtype, "batch": B,
                      "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}), flush=True)
    dt = timed(lambda: v(img), 60)
    print(json.dumps({"what": "LGVae call (inference forward)", "device": "MI355X", "dtype": dtype, "batch": B,
                      "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3)}), flush=True)


def cpu(B, steps=6):
    from oracle import gm_ref, np_ref
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    rng = np.random.Generator(np.random.PCG64(0))
    x = (rng.integers(0, 256, size=(B, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
    perm = np.stack([rng.permutation((H // PATCH) ** 2) for _ in range(B)]).astype(np.int32)
    images = np_ref.scramble_batch(x, perm, PATCH).astype(np.float32)
    F_ = (H // 8) ** 2 * 128
    nz = (rng.standard_normal((B, 128)).astype(np.float32), rng.standard_normal((B, 128)).astype(np.float32),
          rng.uniform(0.02, 0.98, (B, K)).astype(np.float32), (rng.uniform(size=(B, 1024)) > 0.2).astype(np.float32),
          (rng.uniform(size=(B, F_)) > 0.2).astype(np.float32))
    ref = gm_ref.GMRefTrainer(gm_ref.gm_glorot_init(H, H, seed=3, y_size=K), 40.0, 40.0, y_size=K, tau=TAU, dtype=torch.float32)
    ref.train_step(images, *nz)
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.train_step(images, *nz)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"what": "LGGMVae train step, torch-CPU fp32 restatement (oracle/gm_ref.py)", "device": "host CPU", "cores": cores,
                      "batch": B, "images_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    batches = [int(a) for a in sys.argv[1:]] or [64, 512]
    for B in batches:
        for dt in os.environ.get("GM_DTYPES", "bf16,f32").split(","):
            gpu(B, dt)
    if not os.environ.get("GM_NO_CPU"):
        cpu(64)
