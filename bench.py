#!/usr/bin/env python3
"""bench.py -- images/sec of the full SPLIT-VAE training step on MI355X.

One "step" = patch scramble (incl. the per-image random permutation) + forward + ELBO + backward
+ Keras-Adam (+ RCCL gradient all-reduce for N>1) on one synthetic CelebA-64 batch that is already
resident in HBM, i.e. train_step_lg_vae (vae/trainer.py:120-144) behind the augmentation of
vae/main.py:57-61.  Workload (BASELINE.json metric: "SPLIT-VAE CelebA-64 bs512"): H=W=64,
beta=120, patch_size=8, latents 128+128, lr 1e-4, bf16 MFMA contractions with fp32 accumulate /
master weights / ELBO / Adam.  Per-GPU batch is fixed at 512 (weak scaling): `value` is the
whole-job aggregate N*512*K / t.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel, hipEvent-timed inside
the timed region on the launch stream) and `cpu_baseline` (the oracle restatement timed on the
host cores; N=1 only; reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks
# plan scope -> the stem of its HIP kernel symbol (as rocprofv3 prints it; the trailing template arguments select
# variants of the same kernel) for the launches that can be the dominant one
SCOPE_KERNEL = {"fwd.d5": "_Z16tile_conv_kernelIDF16bLi16ELi4ELi4E",
                "wgrad.d5": "void wgrad_tile_kernel<11, ",
                "wgrad.d4": "void wgrad_tile_kernel<9, 1, 2, 8, ",
                "dgrad.d4": "_Z16tile_conv_kernelIDF16bLi64ELi4ELi4ELi6E",
                "fwd.d4": "_Z16tile_conv_kernelIDF16bLi32ELi4ELi4ELi6E"}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_k_traffic.json")   # scripts/traffic.sh: FETCH_SIZE / WRITE_SIZE passes


def measured_traffic(scope):
    """(HBM bytes per launch, kernel symbol) of the scope's kernel from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs, gfx950 x2 read correction); (None, stem) when not measured."""
    stem = SCOPE_KERNEL.get(scope)
    try:
        kernels = json.load(open(TRAFFIC_FILE))["kernels"]
        hits = [(n, k) for n, k in kernels.items() if stem and n.startswith(stem)]
        if len(hits) == 1:
            return hits[0][1]["hbm_bytes_per_launch"], hits[0][0]
    except (OSError, ValueError, KeyError):
        pass
    return None, stem


TRAIN_FLOP_PER_IMAGE = {64: 2.249196e9, 32: 0.562299e9}


def cpu_baseline(H, patch, beta, seconds_budget=20.0):
    """The oracle (torch-CPU fp32 restatement, kind "port") on this box's host cores: same step
    definition on a bounded sample of the workload (a 128-image batch instead of 512)."""
    import numpy as np
    from oracle import np_ref, torch_ref
    # measured on the GPU box (256 hardware threads): the oneDNN/ATen step peaks at 16 threads
    # (8: 320, 16: 499, 32: 460, 64: 208, 128: 96, 256: 1.5 images/s), so 16 is what is used and reported
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    Bc = 128
    rng = np.random.Generator(np.random.PCG64(0))
    x = (rng.integers(0, 256, size=(Bc, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
    G2 = (H // patch) ** 2
    tr = torch_ref.RefTrainer(np_ref.glorot_init(H, H, seed=3), beta, dtype=torch.float32)
    eps = torch.randn(2, Bc, 128)

    def one():
        perm = np.stack([rng.permutation(G2) for _ in range(Bc)])
        img = torch_ref.scramble_batch(torch.from_numpy(x), perm, patch)
        tr.train_step(img, eps[0], eps[1])

    one()
    t0 = time.time()
    n = 0
    while True:
        one()
        n += 1
        dt = time.time() - t0
        if dt > seconds_budget or n >= 40:
            break
    return {"value": round(Bc * n / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d steps of a %d-image CelebA-64 batch (scramble+fwd+ELBO+bwd+Adam), torch-CPU fp32 "
                      "restatement of the TF2 reference (TF2 not installable)" % (n, Bc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (weak scaling: the default, 512 per GPU)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling instead: total batch split evenly over the GPUs (e.g. 512 -> 64 per GPU at N=8, SURVEY config C4)")
    ap.add_argument("--size", type=int, default=64, choices=[32, 64])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel hipEvent table to stderr")
    args = ap.parse_args()

    from split_vae_amd import dist as svdist
    rank, local_rank, world = svdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP device); none visible")
    dev_index = local_rank % torch.cuda.device_count()     # == local_rank on a real N-GPU node; lets a gloo dry run share one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from split_vae_amd import data, trainer
    from split_vae_amd.augmentation import Augmentator
    from split_vae_amd.model import LGVae
    from split_vae_amd.optimizer import Adam
    import torch.distributed as tdist

    H = args.size
    B = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch %d is not divisible by %d GPUs" % (args.global_batch, world))
        B = args.global_batch // world
    beta, patch = (120.0, 8) if H == 64 else (40.0, 1)
    model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=args.dtype, device=dev, seed=3)
    model.beta = beta
    opt = Adam(learning_rate=1e-4)
    aug = Augmentator("scramble", size=patch, seed=1)
    off = rank * B
    x = data.synthetic_images(B, H, H, seed=0, device=dev, sample_offset=off)   # resident in HBM
    reducer = svdist.GradReducer(model.param_table, model.n_params) if world > 1 else None

    def step():
        images = aug.augment(x, sample_offset=off)
        return trainer.train_step(model, images, opt, reducer=reducer, sample_offset=off)

    plan = None
    for _ in range(max(args.warmup, 1)):
        plan = step()
    torch.cuda.synchronize()

    # find the dominant kernel family with a short fully-instrumented pass (outside the timed region)
    plan.profile_filter(None)
    plan.profile_enable(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    table = plan.profile_read()
    plan.profile_enable(False)
    table.sort(key=lambda r: -r["total_ms"])
    if args.profile_all and rank == 0:
        tot = sum(r["total_ms"] for r in table)
        for r in table:
            avg = r["total_ms"] / max(r["launches"], 1)
            tf = r["flops"] / (avg * 1e-3) / 1e12 if r["flops"] else 0.0
            gb = r["bytes"] / (avg * 1e-3) / 1e9 if r["bytes"] else 0.0
            print("%-18s n=%3d avg %8.3f ms  %5.1f%%  %8.1f TFLOP/s %8.1f GB/s" %
                  (r["name"], r["launches"], avg, 100 * r["total_ms"] / tot, tf, gb), file=sys.stderr)
    dom = next(r for r in table if r["flops"] > 0)

    plan.profile_filter(dom["name"])
    plan.profile_enable(True)
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        dt = float(tmax.item())
    prof = [r for r in plan.profile_read() if r["name"] == dom["name"]]
    plan.profile_enable(False)

    if rank != 0:
        return
    value = world * B * args.steps / dt
    out = {
        "metric": "images/sec training step, SPLIT-VAE CelebA-64 bs512" if H == 64 else "images/sec training step, SPLIT-VAE SVHN-32",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "strong" if args.global_batch else "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "SPLIT-VAE %s %dx%d beta=%g patch_size=%d latents=128+128 lr=1e-4, full train step "
                               "(scramble+fwd+ELBO+bwd+Adam%s), per-GPU batch %d" %
                               ("CelebA-64" if H == 64 else "SVHN-32", H, H, beta, patch,
                                "+RCCL grad all-reduce" if world > 1 else "", B),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": "dp%d" % world},
        "step_tflops": round(value * TRAIN_FLOP_PER_IMAGE[H] / 1e12, 2),
    }
    if prof and prof[0]["launches"]:
        avg_ms = prof[0]["total_ms"] / prof[0]["launches"]
        ach = prof[0]["flops"] / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        traffic, symbol = measured_traffic(prof[0]["name"])
        out["roofline"] = {"bound": "mfma", "kernel": prof[0]["name"], "achieved": round(ach, 2), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                           "traffic_source": "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, bytes per launch)" % os.path.basename(TRAFFIC_FILE),
                           "hip_kernel": symbol,
                           "avg_launch_ms": round(avg_ms, 4), "launches": prof[0]["launches"],
                           "flops_per_launch": prof[0]["flops"]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(H, patch, beta)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
