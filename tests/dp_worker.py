"""Worker of tests/test_gpu_dist.py: one data-parallel rank of the SPLIT-VAE step on the (shared) GPU.
usage: python tests/dp_worker.py <out.npz> <global_batch> <steps> [config]   (RANK / WORLD_SIZE / MASTER_* in the env)
config: "svhn32_f32" (default: SVHN-32 fp32, the oracle's precision) | "celeba64_bf16" (config 4's shape: CelebA-64, bf16 contractions,
beta 120, patch 8 -- at 64 images per rank the shard takes its own small-launch tile rules and stream placement)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from split_vae_amd import data, dist as svdist, trainer   # noqa: E402
from split_vae_amd.augmentation import Augmentator          # noqa: E402
from split_vae_amd.model import LGVae                       # noqa: E402
from split_vae_amd.optimizer import Adam                    # noqa: E402


def main():
    out, GB, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    cfg = sys.argv[4] if len(sys.argv) > 4 else "svhn32_f32"
    rank, local_rank, world = svdist.init_from_env()
    torch.cuda.set_device(local_rank % torch.cuda.device_count())      # nccl: one device per rank; gloo: the ranks share device 0
    H, patch, dtype, beta = (64, 8, "bf16", 120.0) if cfg == "celeba64_bf16" else (32, 4, "f32", 40.0)
    lo, hi = svdist.shard_bounds(GB, rank, world)
    model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dtype, device="cuda", seed=3)
    model.beta = beta
    opt = Adam(learning_rate=1e-3)
    aug = Augmentator("scramble", size=patch, seed=1)
    x = data.synthetic_images(hi - lo, H, H, seed=0, device="cuda", sample_offset=lo)
    reducer = svdist.make_reducer(model.param_table, model.n_params) if (world > 1 or os.environ.get("SV_DIST_FORCE")) else None
    if os.environ.get("SV_PYTEST_SIDE_DELAY_US"):       # the plan's own test hook (include/splitvae.h: sv_lgvae_plan_debug); the library reads no such variable
        model.plan(hi - lo).debug("side_delay_us", int(os.environ["SV_PYTEST_SIDE_DELAY_US"]))
    losses = []
    for _ in range(steps):
        img = aug.augment(x, sample_offset=lo)
        plan = trainer.train_step(model, img, opt, reducer=reducer, sample_offset=lo)
        torch.cuda.synchronize()
        losses.append(plan.buffer("losses", torch.float32, (8,)).cpu().numpy().copy())
    if rank == 0:
        np.savez(out, params=model.flat.cpu().numpy(), grads=model.grad_flat.cpu().numpy(), losses=np.stack(losses))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
