#!/usr/bin/env python3
"""bench.py -- images/sec of the full SPLIT-VAE training step on MI355X.

One "step" = patch scramble (incl. the per-image random permutation) + forward + ELBO + backward
+ Keras-Adam (+ RCCL gradient all-reduce for N>1) on one synthetic CelebA-64 batch that is already
resident in HBM, i.e. train_step_lg_vae (vae/trainer.py:120-144) behind the augmentation of
vae/main.py:57-61.  Workload (BASELINE.json metric: "SPLIT-VAE CelebA-64 bs512"): H=W=64,
beta=120, patch_size=8, latents 128+128, lr 1e-4.  The headline (`value`, `ms_per_step`, `dtype`, `roofline`) is the step at the
REFERENCE'S precision: fp32 operands and accumulation (exact-fp32 MFMA, graded against the 157.3 TFLOP/s fp32 matrix peak); the bf16-operand
step (BASELINE config 2's throughput mode) rides along as the named block `bf16` with its own roofline (`--dtype bf16` swaps the two).
The metric's batch is GLOBAL: `--gpus N` splits 512 images evenly over the N ranks (strong scaling: 64 per GPU at N = 8, SURVEY 8d
config C4) and `value` is the whole-job aggregate 512*K / t; the weak-scaling figure (512 per GPU) rides along as `weak` in the N > 1
line, and `--batch B` makes it the headline instead.

    python bench.py                                  # N=1, 200 timed steps
    python bench.py --gpus N --steps K --warmup W    # N>1: spawns N ranks itself (one per GPU, RCCL), or runs as one
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # rank of a launcher that set WORLD_SIZE

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      dominant kernel (hipEvent-timed inside the timed region on its launch stream), the decoder conv stack
                aggregate (north_star's >= 70 % target) and the HBM-bound ELBO kernel
  cpu_baseline  the oracle restatement timed on the host cores (N=1 only; reported baseline, not the target)
  bf16          the same step with bf16 operands (fp32 accumulate / master weights / ELBO / Adam): value, ms_per_step and (N=1) a roofline
                block (dominant kernel live + serial, decoder conv stack) against the 2.5 PFLOP/s bf16 peak
  rows          the other configurations of BASELINE.md section 4 measured in the same process, outside the timed region:
                CelebA-64 bs256 bf16 (configs[1]), SVHN-32 B=64, the 64- and 128-image shards of config 4 at both precisions, one rank
                through the data-parallel path (dp_path_b64), SPLIT-GMVAE, SPLIT-SPAIR (Multi-Bird-Hard and -Easy flag sets)
  rccl_ranks / allreduce_ms / weak   (N>1) the collective actually used, its per-bucket time, the 512-per-GPU (weak scaling) row
The per-launch table goes to stderr.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import split_vae_amd  # noqa: E402
split_vae_amd.configure_hw_queues()                 # before any HIP call (a 4th hardware queue slows the DP step: split_vae_amd/__init__.py)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks
PEAK_HBM_GBS = 8000.0                           # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
TRAIN_FLOP_PER_IMAGE = {64: 2.249196e9, 32: 0.562299e9}
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")            # profiles/<tag>_*traffic.json: the committed PMC passes `traffic` cites (newest round first)


def _traffic_file(dtype="bf16"):
    """The newest committed PMC summary of the step at this precision (`<tag>_traffic.json`: bf16; `<tag>_f32_traffic.json`: the fp32 step)."""
    import glob
    for tag in PROFILE_TAGS:
        c = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", tag + "_*traffic.json")) if f.endswith("_f32_traffic.json") == (dtype == "f32"))
        if c:
            return c[-1]
    return os.path.join(ROOT, "profiles", "r01_k_traffic.json" if dtype != "f32" else "none_f32_traffic.json")


def measured_traffic(kernel_stems, dtype="bf16"):
    """(HBM bytes per launch, kernel symbol(s), file) from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, gfx950 x2 read
    correction).  kernel_stems: strings -- the first stem that matches exactly one kernel -- or (stem, launches) pairs -- a scope made of several kernels: the
    sum over those found.  The value is CACHED evidence of that profile run, not measured by this process: None when no symbol matches."""
    path = _traffic_file(dtype)
    try:
        kernels = json.load(open(path))["kernels"]
    except (OSError, ValueError, KeyError):
        return None, None, path
    if kernel_stems and isinstance(kernel_stems[0], tuple):
        total, names = 0, []
        for stem, count in kernel_stems:
            hits = [(n, k) for n, k in kernels.items() if stem in n]
            if len(hits) != 1:
                return None, None, path                      # a kernel of the scope is missing from the profile: no figure rather than a partial one
            total += count * hits[0][1]["hbm_bytes_per_launch"]
            names.append("%d x %s" % (count, hits[0][0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]))
        return total, " + ".join(names), path
    for stem in kernel_stems:
        hits = [(n, k) for n, k in kernels.items() if stem and stem in n]
        if len(hits) == 1:
            return hits[0][1]["hbm_bytes_per_launch"], hits[0][0], path
    return None, None, path


def cpu_baseline(seconds_budget=25.0):
    """The oracle (torch-CPU fp32 restatement, kind "port") on this box's host cores, SURVEY 8d: config C1 (SVHN-32
    B=64) for >= 20 steps and the headline workload (CelebA-64 B=512) for >= 3 steps, same step definition
    (scramble + forward + ELBO + backward + Keras-Adam).  `value` is the CelebA-64 B=512 rate, the unit of `metric`."""
    import numpy as np
    import torch
    from oracle import np_ref, torch_ref
    # measured on the GPU box (256 hardware threads): the oneDNN/ATen step peaks at 16 threads
    # (8: 320, 16: 499, 32: 460, 64: 208, 128: 96, 256: 1.5 images/s), so 16 is what is used and reported
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)

    def run(H, Bc, patch, beta, min_steps, budget):
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
        t0, n = time.time(), 0
        while True:
            one()
            n += 1
            dt = time.time() - t0
            if n >= min_steps and (dt > budget or n >= 4 * min_steps):
                break
        return Bc * n / dt, n

    svhn, n1 = run(32, 64, 1, 40.0, 20, 0.3 * seconds_budget)
    celeba, n2 = run(64, 512, 8, 120.0, 3, 0.7 * seconds_budget)
    return {"value": round(celeba, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "svhn32_b64": round(svhn, 2),
            "sample": "%d steps of a 512-image CelebA-64 batch (value) and %d steps of a 64-image SVHN-32 batch (svhn32_b64); "
                      "scramble+fwd+ELBO+bwd+Adam, torch-CPU fp32 restatement of the TF2 reference (TF2 not installable), "
                      "%d of the box's %d hardware threads (the restatement's measured optimum)" % (n2, n1, cores, os.cpu_count() or 1)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process
    touches the GPU, wait for them, exit non-zero if any failed.  Rank 0 prints the JSON line."""
    import torch
    have = torch.cuda.device_count()             # counting devices does not initialise HIP on this image
    if have < n:
        sys.stderr.write("bench.py: --gpus %d but this box has %d GPU(s): refusing to report a %d-GPU number from fewer "
                         "devices (one rank per GPU over RCCL is the contract)\n" % (n, have, n))
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for r, p in enumerate(procs):
        c = p.wait()
        if c != 0:
            sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, c))
            rc = rc or (c if c > 0 else 1)
    return rc


class Workload:
    """One configuration of the step, resident on this rank's GPU."""

    def __init__(self, H, B, dtype, dev, rank, world, reducer_cls=None):
        from split_vae_amd import data
        from split_vae_amd.augmentation import Augmentator
        from split_vae_amd.model import LGVae
        from split_vae_amd.optimizer import Adam
        self.H, self.B, self.dtype = H, B, dtype
        self.beta, self.patch = (120.0, 8) if H == 64 else (40.0, 1)
        self.model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype=dtype, device=dev, seed=3)
        self.model.beta = self.beta
        self.opt = Adam(learning_rate=1e-4)
        self.aug = Augmentator("scramble", size=self.patch, seed=1)
        self.off = rank * B
        self.x = data.synthetic_images(B, H, H, seed=0, device=dev, sample_offset=self.off)   # resident in HBM
        # SV_DIST_FORCE=1: the data-parallel path (phase split + bucketed RCCL all-reduce) with a world of one rank
        self.reducer = reducer_cls(self.model.param_table, self.model.n_params) if (reducer_cls and (world > 1 or os.environ.get("SV_DIST_FORCE"))) else None

    def step(self):
        from split_vae_amd import trainer
        # plan=: the augmentation kernel also writes the step's padded low-precision inputs (one pass over the batch less)
        images = self.aug.augment(self.x, sample_offset=self.off, plan=None if os.environ.get("SV_BENCH_NO_STAGED") else self.model.plan(self.x.shape[0]))
        # keep_recon=False: like the reference's step (returns nothing), the reconstructions die inside the fused loss
        return trainer.train_step(self.model, images, self.opt, reducer=self.reducer, sample_offset=self.off,
                                  keep_recon=os.environ.get("SV_BENCH_KEEP_RECON") is not None)

    def timed(self, steps, warmup, world, dev):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks -> seconds."""
        import torch
        import torch.distributed as tdist
        for _ in range(max(warmup, 1)):
            self.step()
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()                                 # (as timeit does: no cyclic-GC pass of the interpreter inside a 20-step window; nothing of the step is skipped)
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
        dt = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        self.last_rank_dt = dt                       # this rank's own time (the reported one is the MAX over ranks)
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt


TABLE_PASSES = 3


def kernel_table(w, passes=TABLE_PASSES, prime=True):
    """Per-launch hipEvent table of the whole step (serial launches: the weight-gradient side stream is off in this
    mode), outside the timed region.  prime=False: not one overlapped step before it (--table-only under rocprofv3: every launch the
    profiler sees ran alone on the chip)."""
    import torch
    plan = w.step() if prime else w.model.plan(w.B)
    torch.cuda.synchronize()
    plan.profile_filter(None)
    if not prime:                       # a discarded serial pass: a kernel's first launch carries its code-object load and LDS-cap call (0.3-1 ms on the host)
        plan.profile_enable(True)
        w.step()
        torch.cuda.synchronize()
    plan.profile_enable(True)           # (clears what was recorded)
    for _ in range(passes):
        w.step()
    torch.cuda.synchronize()
    table = plan.profile_read()
    plan.profile_enable(False)
    table.sort(key=lambda r: -r["total_ms"])
    return plan, table


def grade(r, dtype):
    """One row of the per-launch table against the roofline that binds it: the larger of flops / MFMA peak and algorithmic bytes /
    HBM peak is the launch's floor; `frac` = floor / measured.  Rows with neither (slab reduces, memsets of scratch) are overhead."""
    avg = r["total_ms"] / max(r["launches"], 1) * 1e-3
    t_mfma = r["flops"] / (PEAK_TFLOPS[dtype] * 1e12) if r["flops"] else 0.0
    t_hbm = r["bytes"] / (PEAK_HBM_GBS * 1e9) if r["bytes"] else 0.0
    bound = "mfma" if t_mfma >= t_hbm and t_mfma > 0 else ("hbm" if t_hbm > 0 else "none")
    iss = r.get("issued", r["flops"])
    return {"avg_ms": avg * 1e3, "tflops": r["flops"] / avg / 1e12 if r["flops"] else 0.0, "gbs": r["bytes"] / avg / 1e9 if r["bytes"] else 0.0,
            "bound": bound, "frac": max(t_mfma, t_hbm) / avg if avg else 0.0,
            # MFMA-bound rows: the same launch at the FLOPs its algorithm really multiplies (polyphase forms: < the direct count) = matrix-pipe utilisation
            "tflops_issued": iss / avg / 1e12 if iss else 0.0,
            "frac_issued": (iss / (PEAK_TFLOPS[dtype] * 1e12) / avg if (avg and bound == "mfma") else (max(t_mfma, t_hbm) / avg if avg else 0.0))}


DECODER_STACK = ("fwd.d", "dgrad.d", "wgrad.d", "wgrad.all.reduce", "upsample_bwd", "upsample_fwd")   # (the one slab reduce of ALL layers is booked here whole: conservative)   # d2..d5 convs + everything that exists only for them


def decoder_stack(table, dtype, passes):
    """FLOPs and time of both decoders' conv stacks (d2..d5: forward, input and weight gradients incl. their slab reduces,
    the bilinear-resize adjoints) from the serial per-launch table -- the sub-target north_star quotes against the MFMA peak."""
    fl = ms = iss = 0.0
    for r in table:
        n = r["name"]
        if not n.startswith(DECODER_STACK) or any(p.startswith("d1") for p in n.split(".")[1:]):
            continue
        ms += r["total_ms"] / passes
        fl += r["flops"] * r["launches"] / passes
        iss += r.get("issued", r["flops"]) * r["launches"] / passes
    ach = fl / (ms * 1e-3) / 1e12 if ms else 0.0
    achi = iss / (ms * 1e-3) / 1e12 if ms else 0.0
    return {"flops_per_step": fl, "issued_flops_per_step": iss, "ms_per_step": round(ms, 4), "achieved": round(ach, 1), "achieved_issued": round(achi, 1), "unit": "TFLOP/s",
            "peak": PEAK_TFLOPS[dtype], "frac": round(ach / PEAK_TFLOPS[dtype], 4), "frac_issued": round(achi / PEAK_TFLOPS[dtype], 4),
            "scope": "d2-d5 of both decoders: fwd + dgrad + wgrad (+ slab reduce) + resize adjoint, serial launches"}


def step_flops(table, passes):
    """(direct, issued) FLOPs of one step summed over the serial per-launch table's scopes."""
    fl = sum(r["flops"] * r["launches"] for r in table) / passes
    iss = sum(r.get("issued", r["flops"]) * r["launches"] for r in table) / passes
    return fl, iss


# plan scope -> substrings of its HIP kernel symbol as rocprofv3 prints it (profiles/*_traffic.json keys)
SCOPE_KERNEL = {"fwd.d5": ["tile_conv_kernelIDF16bLi32ELi4ELi4ELi0E"],   # polyphase head
                # (row-ring kernel: RowCfg<KH, KW, CIN, N, WIDTH, MF, NBW, KS, XG, RG, UPS, WAVES, ADJ, ...>; the tile-kernel symbols are the small-batch forms)
                "fwd.d4": ["RowCfg<6, 6, 64, 32, 32, 4, 1, 2, 1, 1, true", "tile_conv_kernelIDF16bLi32ELi4ELi4ELi6E"],
                "fwd.d3": ["tile_conv_kernelIDF16bLi64ELi4ELi4ELi4E", "RowCfg<4, 4, 128, 64, 16, 4, 1, 2, 1, 1, true"],
                "fwd.d2": ["RowCfg<4, 4, 128, 64, 16, 4, 1, 2, 1, 1, false"],
                "fwd.e1": ["RowCfg<3, 3, 32, 32, 32, "], "fwd.e2": ["RowCfg<3, 3, 128, 64, 16, ", "tile_conv_kernelIDF16bLi64ELi4ELi4ELi0E"],
                "fwd.e3": ["RowCfg<2, 2, 256, 128, 16, "],
                "wgrad.d5": ["wgrad_tile_kernel<7, 1, 2, 8, ", "wgrad_tile_kernel<11, "], "wgrad.d4": ["wgrad_roll_kernel", "wgrad_tile_kernel<9, 1, 2, 8, "],
                "wgrad.d3": ["wgrad_tile_kernel<4, 2, 4, 8, 1, 2>"], "wgrad.e2": ["wgrad_e2_kernel"], "wgrad.e1": ["wgrad_e1_kernel"],
                "dgrad.d4": ["RowCfg<6, 6, 32, 64, 32, "], "dgrad.d5": ["RowCfg<6, 6, 8, 32, 64, "], "dgrad.d3": ["RowCfg<4, 4, 64, 128, "],
                "dgrad.e2": ["RowCfg<3, 3, 64, 128, 16, 4, 2, "]}


# the fp32 step: a plan scope may be SEVERAL kernels (the polyphase forms: class launches + border kernels); (symbol stem, launches per scope) -- the scope's
# traffic is the sum.  wgrad_tile_f32_kernel<TPW, COF, LDY, CW, SX, G4>; <7, 2, 32, 16, 1, .> serves d4's class (0, 0) AND the head's merged 25-tap form (one symbol:
# the per-launch figure in profiles/*_f32_traffic.json is the mean of the two uses)
SCOPE_KERNEL_F32 = {
    "wgrad.d4": [("wgrad_tile_f32_kernel<7, 2, 32, 16, 1, ", 1), ("wgrad_tile_f32_kernel<5, 2, 32, 16, 1, ", 2), ("wgrad_tile_f32_kernel<4, 2, 32, 16, 1, ", 1),
                 ("polyc_wgrad_frame_kernel<6, 4, 2", 1), ("polyc_frame_sum_kernel", 1), ("polyc_wgrad_project_kernel", 1)],
    "wgrad.d5": [("wgrad_tile_f32_kernel<7, 2, 32, 16, 1, ", 1), ("polyc_wgrad_frame_kernel<6, 2, 1", 1), ("polyc_frame_sum_kernel", 1), ("polyc_wgrad_project_kernel", 1)],
    "wgrad.d3": [("wgrad_tile_f32_kernel<4, 4, 64, 16, 1, ", 1)], "wgrad.e2": [("wgrad_tile_f32_kernel<9, 4, 64, 16, 2, ", 1)],
    "wgrad.e1": [("wgrad_tile_f32_kernel<3, 2, 32, 16, 1, ", 1)],
    "dgrad.d4": [("polyd_edge_kernel<float, 9, 2", 1), ("polyd_corner_kernel<float, 2", 1), ("tile_conv_kernel<float, 64, 2, ", 1)],
    "fwd.d4": [("polyc_fix_kernel<float, 6, 4, 2", 1), ("tile_conv_kernel<float, 32, 4, ", 1)],
}


def wgrad_main_layers(images_per_launch, dtype, world=1):
    """The layers whose weight gradient the plan keeps on the main stream (csrc/lgvae_plan.hip: run_wgrad_layers; the rest go to the side stream)."""
    env = os.environ.get("SV_WGRAD_MAIN")
    if env is not None:
        return env.split(",")
    if images_per_launch < 768:
        return ["e1"] if images_per_launch >= 512 else ["e1", "e2"]
    if dtype == "bf16" and world == 1 and int(os.environ.get("SV_SIDE_STREAMS", "2")) >= 2:
        return ["e1"]                            # whole steps: two side streams take everything else
    if dtype == "f32":
        return ["e1", "e2", "e3"]                # fp32: the encoders' on the main stream, the decoders' on the side stream
    return ["e1", "e2", "d4"] if dtype == "bf16" and os.environ.get("SV_NO_WGRAD_ROLL") is None else ["e1", "e2", "d5"]


SPAIR_FLAGS = {
    # README.md:107 (SPLIT-VAE on Multi-Bird-Hard, dataset cub_ckb_rot_6) -- BASELINE configs[4]
    "hard": dict(model="lg_spair", latent_size=64, bg_latent_size=64, local_latent_size=64, patch_size=8, z_bg_beta=1.0, z_what_beta=0.5,
                 split_z_l=True, concat_z_what=True, dense_local=True, dense_bg=True),
    # README.md:93 (SPLIT-VAE on Multi-Bird-Easy, dataset cub_solid_fixed)
    "easy": dict(model="lg_spair", latent_size=64, bg_latent_size=4, local_latent_size=4, patch_size=8, z_bg_beta=10.0,
                 split_z_l=True, concat_z_what=True, dense_local=True, dense_bg=True),
}
SPAIR_WORKLOAD = {
    "hard": "SPLIT-SPAIR (lg_spair) Multi-Bird-Hard flag set, README.md:107: --z_bg_beta 1 --patch_size 8 --latent_size 64 --bg_latent_size 64 "
            "--local_latent_size 64 -split_z_l --z_what_beta 0.5 -concat_z_what -dense_local -dense_bg; batch 32 (spair/main.py default); canvases "
            "48x48 (the reference hard-codes 48x48 and a 4x4 cell grid, spair/spair.py:99,411: BASELINE's '128x128' label is stale), synthetic",
    "easy": "SPLIT-SPAIR (lg_spair) Multi-Bird-Easy flag set, README.md:93: --z_bg_beta 10 --patch_size 8 --latent_size 64 --bg_latent_size 4 "
            "--local_latent_size 4 -split_z_l -concat_z_what -dense_local -dense_bg; batch 32; canvases 48x48 (hard-coded in the reference), synthetic",
}


def timed_blocks(step, steps, blocks=3):
    """`steps` steps of a host-launch-bound row timed as `blocks` equal blocks (synchronize on both sides of each): returns (median block's seconds per step,
    every block's ms per step, seconds per step over all blocks).  The extra rows report the MEDIAN block: a one-off host stall (allocator trim after the
    previous row's model was freed, a page fault storm) inside a 60-step window once read 3.51 ms for a 2.29-ms step (profiles/r06_m_bench.json); the
    all-blocks mean stays in the row as `ms_per_step_mean`.  The headline metric is NOT timed this way (exactly K steps, one window: main())."""
    import torch
    per = max(steps // blocks, 1)
    ts = []
    k = 0
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            step(k)
            k += 1
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / per)
    return sorted(ts)[len(ts) // 2], [round(1e3 * t, 4) for t in ts], sum(ts) / len(ts)


def spair_row(dev, which="hard", B=32, steps=60, warmup=5):
    """SPLIT-SPAIR (config 5; `which` picks README.md:107 Multi-Bird-Hard -- what BASELINE names -- or README.md:93 Multi-Bird-Easy) train step:
    forward + losses + backward + clipnorm Adam as one native launch sequence (sv_tape_run); fp32 like the reference and with bf16 convolutions."""
    import torch
    from split_vae_amd import spair, spair_main, spair_trainer
    from split_vae_amd.augmentation import Augmentator
    out = {"unit": "images/s", "batch": B, "steps": steps, "workload": SPAIR_WORKLOAD[which],
           "launch": "one native launch sequence per step (sv_tape_run: forward, losses, adjoint, Adam; eager launches) on two HIP streams: the x-hat / background "
                     "networks on a lane beside the object pipeline, the lane-0 layers' weight gradients on that lane too (csrc/tape.hip); f32 = the reference's "
                     "precision, bf16 = bf16 operands in the spatial convolutions only"}
    for dt_ in ("f32", "bf16"):
        cfg = spair_main.default_config(dtype=dt_, **SPAIR_FLAGS[which])
        model = spair.get_model(cfg, device=dev, seed=0)
        x, _ = spair_main.synthetic_canvases(B, seed=1, device=dev)
        images = Augmentator("scramble", size=cfg.patch_size, seed=2).augment(x)
        opt = spair_trainer.ClipnormAdam(cfg.learning_rate, clipnorm=1.0)
        step_fn = lambda im, i: spair_trainer.train_step(model, im, opt, i, cfg)          # noqa: E731 (native: spair_native.NativeStep)
        for i in range(warmup):
            step_fn(images, i)
        t, blocks_ms, t_mean = timed_blocks(lambda i: step_fn(images, warmup + i), steps)
        out[dt_] = {"value": round(B / t, 1), "ms_per_step": round(1e3 * t, 4), "blocks_ms": blocks_ms, "ms_per_step_mean": round(1e3 * t_mean, 4),
                    "timing": "ms_per_step = the median of the equal blocks in blocks_ms (bench.py: timed_blocks); ms_per_step_mean = all of them"}
        if dt_ == "f32":
            ns = model.native(B, cfg)
            # algorithmic HBM bytes of the step: the variables once per pass (forward, input gradients, weight gradients written) + Adam's
            # 28 B per variable; achieved = that / the step time (the step is HBM- and launch-latency-bound: 31.9 M variables, 32 images)
            nv = model.count_params()
            by = nv * (4 * 3 + 28)
            out[dt_]["roofline"] = {"bound": "hbm", "algorithmic_bytes": by, "achieved": round(by / t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(by / t / 1e9 / PEAK_HBM_GBS, 4), "tape_nodes": ns.n_nodes}
        del step_fn, model
    return out


def gm_row(dev, B=64, steps=100, warmup=10, dtype="bf16"):
    """SPLIT-GMVAE (config 3, README.md:62: --model lggmvae --beta 40 --alpha 40 --y_size 30 --patch_size 4 on SVHN-32, batch 64 =
    vae/main.py:23's default) train step: scramble + forward + losses + backward + Keras-Adam; dtype f32 = the reference's precision, bf16 = bf16 contractions."""
    import torch
    from split_vae_amd import data
    from split_vae_amd.augmentation import Augmentator
    from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
    from split_vae_amd.optimizer import Adam
    x = data.synthetic_images(B, 32, 32, seed=0, device=dev)
    aug = Augmentator("scramble", size=4, seed=1)
    m = LGGMVae(128, 128, [-1, 32, 32, 3], 30, 0.4, dtype=dtype, device=dev, seed=3)
    m.beta, m.alpha = 40.0, 40.0
    opt = Adam(learning_rate=1e-4)
    plan = m.plan(B)                       # plan=: the augmentation kernel also writes the step's padded inputs (as the SPLIT-VAE rows do)
    for _ in range(warmup):
        train_step_lg_gm_vae(m, aug.augment(x, plan=plan), opt)
    t, blocks_ms, t_mean = timed_blocks(lambda i: train_step_lg_gm_vae(m, aug.augment(x, plan=plan), opt), steps, blocks=4)
    # 135.0 M forward MACs per image for the whole LGGMVae at SVHN-32 [derived]; train FLOP = 6 MACs_fwd - 4 MACs of the two first convs
    # (gmvae encoder 16 x 16 x 128 x 108 = 3.54 M, local encoder 0.88 M: no input gradient).  A 64-image step is launch-bound: the fraction says so.
    fl = 6 * 135.0e6 - 4 * (3.54e6 + 0.88e6)
    return {"value": round(B / t, 1), "unit": "images/s", "ms_per_step": round(1e3 * t, 4), "blocks_ms": blocks_ms, "ms_per_step_mean": round(1e3 * t_mean, 4),
            "timing": "ms_per_step = the median of the equal blocks in blocks_ms (bench.py: timed_blocks); ms_per_step_mean = all of them",
            "steps": steps, "batch": B, "dtype": dtype,
            "workload": "BASELINE configs[2]: SPLIT-GMVAE SVHN-32 y_size=30 beta=40 alpha=40 patch_size=4 tau=0.4" + (" at the reference's precision" if dtype == "f32" else ""),
            "roofline": {"bound": "mfma", "flops_per_image": fl, "achieved": round(B / t * fl / 1e12, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                         "frac": round(B / t * fl / 1e12 / PEAK_TFLOPS[dtype], 4),
                         "note": "64 images per step: ~100 launches of a few microseconds each; the step is launch- and latency-bound, not matrix-pipe-bound"}}


def roofline_block(table, dom, worst, prof, dtype, B, world, step_ms=None, passes=TABLE_PASSES):
    """The `roofline` object of one precision.  Top level = the dominant scope of the serial per-launch table ALONE ON THE CHIP (hipEvents on its launch
    stream, measured by this process outside the timed region); flat scalars give serial scope / decoder stack / whole step against the MFMA peak at BOTH
    FLOP counts -- `*_direct` (SURVEY 8d's count, 2 B OH OW Cout KH KW Cin: the polyphase forms multiply fewer, so a row may exceed 1) and `*_issued` (what
    the chosen algorithm really multiplies: matrix-pipe utilisation).  `live` = the same scope inside the timed region, where up to three queues share the
    chip (a co-scheduling figure, not a kernel figure)."""
    gs = grade(dom, dtype)
    peak = PEAK_TFLOPS[dtype]
    traffic, symbol, tfile = measured_traffic((SCOPE_KERNEL if dtype == "bf16" else SCOPE_KERNEL_F32).get(dom["name"], []), dtype)
    ds = decoder_stack(table, dtype, passes)
    fl_step, iss_step = step_flops(table, passes)
    rl = {"bound": "mfma", "kernel": dom["name"], "achieved": round(gs["tflops"], 2), "peak": peak, "unit": "TFLOP/s", "frac": round(gs["frac"], 4),
          "traffic": traffic,
          "traffic_source": "cached: profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                            "bench, bytes per launch set; null when a kernel symbol of this build is not in it)" % os.path.basename(tfile),
          "algorithmic_bytes": dom["bytes"],
          "hip_kernel": symbol,
          "avg_launch_ms": round(gs["avg_ms"], 4), "launches": dom["launches"], "flops_per_launch": dom["flops"], "issued_flops_per_launch": dom.get("issued", dom["flops"]),
          "measured": "serial: the scope alone on the chip, hipEvents on its launch stream, this process (per-launch table passes outside the timed region)",
          "serial_frac_direct": round(gs["frac"], 4), "serial_frac_issued": round(gs["frac_issued"], 4),
          "decoder_stack_frac_direct": ds["frac"], "decoder_stack_frac_issued": ds["frac_issued"],
          "count_note": "direct = SURVEY 8d's FLOP count of the layer as the reference computes it; issued = the FLOPs the chosen algorithm really multiplies on the "
                        "matrix pipe (polyphase forms of the upsample->conv layers: 81 or 100 of 144 tap products per low-res pixel + border terms; include/splitvae.h: "
                        "sv_lgvae_profile_read_issued) -- only the issued figure is a matrix-pipe utilisation",
          "decoder_stack": ds}
    if step_ms:
        rl["step_frac_direct"] = round(fl_step / (step_ms * 1e-3) / 1e12 / peak, 4)
        rl["step_frac_issued"] = round(iss_step / (step_ms * 1e-3) / 1e12 / peak, 4)
        rl["step_flops_direct"], rl["step_flops_issued"] = fl_step, iss_step
    if prof and prof[0]["launches"]:
        avg_ms = prof[0]["total_ms"] / prof[0]["launches"]
        ach = prof[0]["flops"] / (avg_ms * 1e-3) / 1e12
        rl["live"] = {"avg_launch_ms": round(avg_ms, 4), "launches": prof[0]["launches"], "achieved": round(ach, 2), "frac": round(ach / peak, 4),
                      "stream": "a weight-gradient side stream (co-runs with the input-gradient chain and, in whole steps at this size, a second side stream)"
                                if dom["name"].startswith("wgrad.") and dom["name"].split(".")[1] not in wgrad_main_layers(2 * B, dtype, world) else
                                "main (the weight-gradient side stream runs other layers' launches beside it)",
                      "note": "hipEvents around the scope on its stream INSIDE the timed region, where up to three launches share the chip: the wall time counts what "
                              "runs beside it -- a co-scheduling figure, not the kernel's"}
    if worst is not None:
        gw = grade(worst, dtype)
        rl["worst_large"] = {"kernel": worst["name"], "bound": gw["bound"], "avg_launch_ms": round(gw["avg_ms"], 4),
                             "achieved": round(gw["tflops"] if gw["bound"] == "mfma" else gw["gbs"], 2),
                             "unit": "TFLOP/s" if gw["bound"] == "mfma" else "GB/s", "frac": round(gw["frac"], 4), "frac_issued": round(gw["frac_issued"], 4),
                             "note": "the launch >= 0.1 ms with the lowest fraction of its roofline (serial table, direct count)"}
    rl["table"] = [{"kernel": r["name"], "ms": round(grade(r, dtype)["avg_ms"], 4), "bound": grade(r, dtype)["bound"],
                    "frac": round(grade(r, dtype)["frac"], 4), "frac_issued": round(grade(r, dtype)["frac_issued"], 4)} for r in table[:12]]
    # the HBM-bound entry: the ELBO kernel when the step runs it, else (training steps evaluate the loss in the decoder
    # head's epilogue) the Adam update -- 28 algorithmic bytes per parameter
    elbo = next((r for r in table if r["name"].startswith("dlogistic")), None) or \
        next((r for r in table if r["name"] == "adam_step"), None)
    if elbo and elbo["launches"]:
        ems = elbo["total_ms"] / elbo["launches"]
        etr, esym, _ = measured_traffic(["dlogistic_kernel"] if elbo["name"].startswith("dlogistic") else ["adam_kernel"], dtype)
        rl["hbm"] = {"bound": "hbm", "kernel": elbo["name"], "algorithmic_bytes": elbo["bytes"],
                     "avg_launch_ms": round(ems, 4), "achieved": round(elbo["bytes"] / (ems * 1e-3) / 1e9, 1),
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(elbo["bytes"] / (ems * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                     "traffic": etr, "traffic_gbs": round(etr / (ems * 1e-3) / 1e9, 1) if etr else None}
    return rl


def pick_rows(table, dtype):
    """(dominant launch = the top row of the serial table with FLOPs, the worst launch >= 0.1 ms against its roofline)."""
    dom = next(r for r in table if r["flops"] > 0)
    large = [r for r in table if r["total_ms"] / max(r["launches"], 1) >= 0.1 and (r["flops"] or r["bytes"])]
    worst = min(large, key=lambda r: grade(r, dtype)["frac"]) if large else None
    return dom, worst


def print_table(table, dtype, H, B):
    tot = sum(r["total_ms"] for r in table)
    sys.stderr.write("per-launch table (%s %dx%d B=%d, serial launches, hipEvents):\n" % (dtype, H, H, B))
    for r in table:
        g = grade(r, dtype)
        tail = " (direct FLOP count; %5.1f%% at the FLOPs the launch really multiplies)" % (100 * g["frac_issued"]) if g["bound"] == "mfma" else ""
        sys.stderr.write("%-18s n=%3d avg %8.3f ms  %5.1f%%  %8.1f TFLOP/s %8.1f GB/s  %5.1f%% of the %s roofline%s\n" %
                         (r["name"], r["launches"], g["avg_ms"], 100 * r["total_ms"] / tot, g["tflops"], g["gbs"], 100 * g["frac"], g["bound"], tail))


DTYPE_NOTE = {"f32": "f32 operands, f32 accumulate (exact-fp32 MFMA v_mfma_f32_16x16x4_f32): the reference's precision (vae/model.py:12)",
              "bf16": "bf16 operands, fp32 accumulate / master weights / ELBO / Adam (BASELINE config 2's throughput mode; narrower than the reference's fp32)"}


def precision_block(dev, H, B, dtype, steps=40, warmup=5):
    """The same step at the OTHER precision as a first-class measurement (the headline is the reference's fp32; this is the bf16 throughput
    mode of BASELINE config 2 -- or the fp32 step when --dtype bf16 makes bf16 the headline): the same timed protocol (W untimed + K timed
    steps between synchronizes), its own serial table and roofline block graded against that precision's dense MFMA peak."""
    import torch
    w = Workload(H, B, dtype, dev, 0, 1)
    for _ in range(warmup):
        w.step()
    torch.cuda.synchronize()
    plan, table = kernel_table(w)
    print_table(table, dtype, H, B)
    dom, worst = pick_rows(table, dtype)
    plan.profile_filter(dom["name"])
    plan.profile_enable(True)
    dt = w.timed(steps, warmup, 1, dev)
    prof = [r for r in plan.profile_read() if r["name"] == dom["name"]]
    plan.profile_enable(False)
    value = B * steps / dt
    out = {"value": round(value, 1), "unit": "images/s", "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps, "warmup": warmup,
           "dtype": DTYPE_NOTE[dtype], "per_gpu_batch": B,
           "step_tflops": round(value * TRAIN_FLOP_PER_IMAGE[H] / 1e12, 2),
           "step_frac_of_peak": round(value * TRAIN_FLOP_PER_IMAGE[H] / 1e12 / PEAK_TFLOPS[dtype], 4),
           "roofline": roofline_block(table, dom, worst, prof, dtype, B, 1, step_ms=1e3 * dt / steps)}
    del w
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak scaling) instead of the default global batch")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total batch split evenly over the GPUs (default 512 = BASELINE.json's metric: 64 per GPU at N=8, SURVEY config C4)")
    ap.add_argument("--size", type=int, default=64, choices=[32, 64])
    ap.add_argument("--dtype", default="f32", choices=["bf16", "f32"],
                    help="the headline's precision: f32 = the reference's (default); the other precision rides along as a named block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rows", action="store_true", help="skip the extra configurations (SVHN-32, small shards, SPLIT-GMVAE, SPLIT-SPAIR)")
    ap.add_argument("--no-other-precision", "--no-fp32", dest="no_other", action="store_true",
                    help="skip the block of the other precision (`bf16` beside the fp32 headline)")
    ap.add_argument("--table-only", type=int, default=0, metavar="PASSES",
                    help="run PASSES passes of the serial per-launch table and stop (for `rocprofv3 --kernel-trace --stats`: kernel durations alone on the chip, "
                         "what `roofline.serial` and the fractions of the table quote)")
    ap.add_argument("--profile-all", action="store_true", help="(kept for compatibility: the per-launch table always goes to stderr)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # nothing above touched the GPU

    import torch
    from split_vae_amd import dist as svdist
    rank, local_rank, world = svdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to report one as the other" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP device); none visible")
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    dev_index = local_rank % torch.cuda.device_count()     # == local_rank on a real N-GPU node; lets a gloo dry run share one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    H = args.size
    if args.batch and args.global_batch:
        raise SystemExit("--batch (per GPU, weak scaling) and --global-batch (strong scaling) exclude each other")
    if args.batch:
        B, strong = args.batch, False
    else:
        gb = args.global_batch or 512                  # BASELINE.json's metric: bs512 is the GLOBAL batch at every N
        if gb % world:
            raise SystemExit("global batch %d is not divisible by %d GPUs" % (gb, world))
        B, strong = gb // world, True
    w = Workload(H, B, args.dtype, dev, rank, world, svdist.make_reducer)
    for _ in range(0 if args.table_only else max(args.warmup, 1)):
        w.step()
    torch.cuda.synchronize()

    # per-launch table + the dominant kernel family (outside the timed region)
    plan, table = kernel_table(w, passes=args.table_only or TABLE_PASSES, prime=not args.table_only)
    if rank == 0:
        print_table(table, args.dtype, H, B)
    if args.table_only:
        if rank == 0:
            print(json.dumps({"table_only": args.table_only, "rows": [{"kernel": r["name"], "launches": r["launches"], "ms": round(r["total_ms"] / max(r["launches"], 1), 4)} for r in table[:12]]}))
        return
    # the dominant kernel = the top row of the serial table, whatever it is.  In the timed region the streams of the backward pass overlap:
    # `achieved` is what the hipEvents around the launch on ITS stream give there (co-running launches of the other streams included; `stream`
    # names it), `serial` what the launch takes alone on the chip.
    dom, worst = pick_rows(table, args.dtype)

    plan.profile_filter(dom["name"])
    plan.profile_enable(True)
    # the W warm-up steps already ran (and the serial table passes after them); a short timed region (K = 20 is 37 ms) starts from whatever
    # clock / power state the table's profiling syncs left behind (+-3 % run to run), so the chip is brought back to the steady state of
    # back-to-back steps first: untimed steps, never fewer than 30 in total before the clock starts
    dt = w.timed(args.steps, max(0, 30 - args.warmup), world, dev)
    prof = [r for r in plan.profile_read() if r["name"] == dom["name"]]
    plan.profile_enable(False)

    extra = {}
    if w.reducer is not None:
        # per-bucket all-reduce time, alone on the chip (what the backward has to hide)
        ar = {}
        for name in w.reducer.buckets:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(6):
                if it == 1:
                    torch.cuda.synchronize(); e0.record()
                w.reducer.launch(w.model.grad_flat, name)
                w.reducer.wait()
            e1.record(); torch.cuda.synchronize()
            ar[name] = round(e0.elapsed_time(e1) / 5, 4)
        # self-diagnosis of a multi-GPU run: every rank's own ms/step of the timed region, and the time the compute stream really waits for the
        # collectives (hipEvent pair around reducer.wait(), 20 extra untimed steps): ~0 when the all-reduce hides behind the encoders' backward
        if world > 1:
            mine = torch.tensor([w.last_rank_dt / args.steps * 1e3], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
            extra["per_rank_ms_per_step"] = [round(float(t_.item()), 4) for t_ in allr]
        w.reducer.wait_events = []
        for _ in range(20):
            w.step()
        torch.cuda.synchronize()
        ev = w.reducer.wait_events
        w.reducer.wait_events = None
        if ev:
            extra["exposed_allreduce_ms"] = round(sum(a_.elapsed_time(b_) for a_, b_ in ev) / len(ev), 4)
        extra["dp_mode"] = getattr(w.reducer, "mode", None) if getattr(w.reducer, "mode", None) != "auto" else ("single" if world == 1 else "events")
        native = os.environ.get("SV_DIST_BACKEND") == "sv_comm"
        extra["rccl_ranks"] = world if (backend == "nccl" or native) else 0
        extra["dist_backend"] = "sv_comm (RCCL through the C ABI)" if native else backend
        extra["allreduce_ms"] = ar
        extra["allreduce_bytes"] = {k: int(sum(e - b for b, e in v) * 4) for k, v in w.reducer.buckets.items()}
        if world > 1 and strong and H == 64 and not args.no_rows:
            # the weak-scaling row: 512 images per GPU (global 512 N), what rounds 1-3 reported as the headline
            ws = Workload(64, 512, args.dtype, dev, rank, world, svdist.make_reducer)
            ks = max(args.steps, 60)
            dts = ws.timed(ks, 10, world, dev)
            extra["weak"] = {"global_batch": 512 * world, "per_gpu_batch": 512, "value": round(world * 512 * ks / dts, 1),
                             "ms_per_step": round(1e3 * dts / ks, 4), "steps": ks, "scaling": "weak"}
            del ws
    rows = {}
    other = None
    other_dt = "bf16" if args.dtype == "f32" else "f32"
    if world == 1 and not args.no_other:
        # N = 1: the full block (own serial table + roofline), 200 timed steps for the sub-2-ms bf16 step, 40 for fp32
        other = precision_block(dev, H, B, other_dt, steps=200 if other_dt == "bf16" else 40, warmup=10 if other_dt == "bf16" else 5)
    elif world > 1 and not args.no_other:
        # N > 1: the other precision's whole-job rate through the same data-parallel path (no table: the serial table is an N = 1 measurement)
        wo = Workload(H, B, other_dt, dev, rank, world, svdist.make_reducer)
        ko = max(args.steps, 60)
        dto = wo.timed(ko, 10, world, dev)
        other = {"value": round(world * B * ko / dto, 1), "unit": "images/s", "ms_per_step": round(1e3 * dto / ko, 4), "steps": ko, "warmup": 10,
                 "dtype": DTYPE_NOTE[other_dt], "per_gpu_batch": B, "global_batch": world * B}
        del wo
    if world == 1 and not args.no_rows and H == 64 and B == 512:
        def row(Hr, Br, dt_, steps, warm=10, reducer_cls=None, workload=None):
            wr = Workload(Hr, Br, dt_, dev, 0, 1, reducer_cls)
            t = wr.timed(steps, warm, 1, dev)
            r = {"value": round(Br * steps / t, 1), "unit": "images/s", "ms_per_step": round(1e3 * t / steps, 4), "steps": steps, "dtype": dt_,
                 "step_tflops": round(Br * steps / t * TRAIN_FLOP_PER_IMAGE[Hr] / 1e12, 2)}
            r["frac_of_peak"] = round(r["step_tflops"] / PEAK_TFLOPS[dt_], 4)
            if workload:
                r["workload"] = workload
            return r
        rows["celeba64_b256"] = row(64, 256, "bf16", 200, workload="BASELINE configs[1]: SPLIT-VAE CelebA-64 beta=120 patch_size=8 bs256 bf16 on one MI355X")
        rows["celeba64_b256_f32"] = row(64, 256, "f32", 60)
        rows["svhn32_b64"] = row(32, 64, "bf16", 200)          # config C1's shape on the GPU
        rows["svhn32_b64_f32"] = row(32, 64, "f32", 200, workload="BASELINE configs[0]'s shape (SVHN-32 beta=40 patch_size=1 bs64) at the reference's precision")
        for key, dt_ in (("lggmvae_svhn32_b64", "bf16"), ("lggmvae_svhn32_b64_f32", "f32")):      # config 3 (SPLIT-GMVAE), both precisions
            try:
                rows[key] = gm_row(dev, dtype=dt_)
            except Exception as e:
                rows[key] = {"error": repr(e)[:200]}
        # config 4's per-GPU shards (global 512 over 8 / 4 GPUs), both precisions: what bounds strong scaling before any link time
        rows["celeba64_b64"] = row(64, 64, "bf16", 200)
        rows["celeba64_b128"] = row(64, 128, "bf16", 200)
        rows["celeba64_b64_f32"] = row(64, 64, "f32", 100)
        rows["celeba64_b128_f32"] = row(64, 128, "f32", 100)
        rows["long_run"] = {"steps": 400, "dtype": args.dtype, "ms_per_step": round(1e3 * w.timed(400, 0, 1, dev) / 400, 4)}
        try:
            rows["lg_spair_b32"] = spair_row(dev, "hard")          # config 5 (SPLIT-SPAIR, Multi-Bird-Hard: README.md:107), SURVEY 8f F4
            rows["lg_spair_easy_b32"] = spair_row(dev, "easy")     # README.md:93 (Multi-Bird-Easy)
        except Exception as e:                                    # never at the headline's expense
            rows["lg_spair_b32"] = {"error": repr(e)[:200]}
        # config 4's shard through the DATA-PARALLEL path with a world of one rank (process group, phase split, RCCL all-reduce of every bucket
        # over one rank): what the DP host path costs on top of rows.celeba64_b64*.  In a FRESH process each, as a rank of a real job is (this
        # process has run five model families by now: its streams and allocator state are not a rank's).
        rows["dp_path_b64"] = {"workload": "config 4's 64-image shard, one rank through the RCCL path (SV_DIST_FORCE=1), a fresh process per precision"}
        for k in ("f32", "bf16"):
            try:
                env = dict(os.environ, SV_DIST_FORCE="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--batch", "64", "--dtype", k, "--steps", "200" if k == "bf16" else "100",
                                    "--warmup", "10", "--no-rows", "--no-other-precision", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
                c = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                base = rows["celeba64_b64" + ("_f32" if k == "f32" else "")]["ms_per_step"]
                rows["dp_path_b64"][k] = {"value": c["value"], "unit": "images/s", "ms_per_step": c["ms_per_step"], "steps": c["steps"],
                                          "backend": c.get("dist_backend"), "vs_plain_step": round(c["ms_per_step"] / base, 4)}
            except Exception as e:
                rows["dp_path_b64"][k] = {"error": repr(e)[:200]}

        # a `--gpus 2`-shaped dry run on the ONE device: two rank processes over gloo sharing the GPU, global batch 512 = 256 images per rank, fp32.  Not a
        # scaling number (the ranks take turns on one chip): it exercises the whole N > 1 host path -- process group, sharding, bucket events, all-reduce
        # hand-over, 1/world in Adam -- so that a regression there shows in the driver's one-GPU line.  Reference for the ratio: two plain 256-image steps.
        try:
            port = _free_port()
            procs = []
            for r_ in range(2):
                env = dict(os.environ, RANK=str(r_), LOCAL_RANK="0", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                           SV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gpus", "2", "--dtype", "f32", "--steps", "40", "--warmup", "5",
                                               "--no-rows", "--no-other-precision", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
            try:
                outs = [p_.communicate(timeout=240)[0] for p_ in procs]
            finally:
                for p_ in procs:                         # (a rank that is still alive here is stuck: never leave it behind on the GPU)
                    if p_.poll() is None:
                        p_.kill()
            c = [json.loads(ln) for ln in outs[0].splitlines() if ln.startswith("{")][-1]
            base = 2 * rows["celeba64_b256_f32"]["ms_per_step"]
            rows["dp2_shared_gpu_b256_f32"] = {"value": c["value"], "unit": "images/s", "ms_per_step": c["ms_per_step"], "steps": c["steps"], "n_ranks": 2,
                                               "backend": c.get("dist_backend"), "dp_mode": c.get("dp_mode"), "per_rank_ms_per_step": c.get("per_rank_ms_per_step"),
                                               "exposed_allreduce_ms": c.get("exposed_allreduce_ms"),
                                               "vs_two_plain_b256_steps": round(c["ms_per_step"] / base, 4),
                                               "workload": "two ranks x 256 images (global 512) over gloo SHARING the one GPU: the N > 1 host path, not a scaling number"}
        except Exception as e:
            rows["dp2_shared_gpu_b256_f32"] = {"error": repr(e)[:200]}

    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return
    value = world * B * args.steps / dt
    out = {
        "metric": "images/sec training step, SPLIT-VAE CelebA-64 bs512" if H == 64 else "images/sec training step, SPLIT-VAE SVHN-32",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": DTYPE_NOTE[args.dtype] + ("" if args.dtype == "f32" else "; the fp32 step is the `fp32` block of this line"),
        "dtype_short": args.dtype, "data": "synthetic",
        "config": {"workload": "SPLIT-VAE %s %dx%d beta=%g patch_size=%d latents=128+128 lr=1e-4, full train step "
                               "(scramble+fwd+ELBO+bwd+Adam%s), per-GPU batch %d" %
                               ("CelebA-64" if H == 64 else "SVHN-32", H, H, w.beta, w.patch,
                                "+RCCL grad all-reduce" if world > 1 else "", B),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": "dp%d" % world,
                   # the step's outputs are the losses and the updated variables (vae/trainer.py:121-144 returns nothing): the
                   # reconstruction tensors are consumed by the loss inside the head conv and not written to HBM
                   "reconstructions": "stored" if os.environ.get("SV_BENCH_KEEP_RECON") is not None else "not stored (dead after the fused loss)"},
        # untimed steps run before the clock starts: the W requested, the serial per-launch table passes, and the steady-state top-up
        "untimed_steps_before_timing": max(args.warmup, 1) + TABLE_PASSES + 1 + max(max(0, 30 - args.warmup), 1),
        "step_tflops": round(value * TRAIN_FLOP_PER_IMAGE[H] / 1e12, 2),
        "step_frac_of_peak": round(value * TRAIN_FLOP_PER_IMAGE[H] / 1e12 / PEAK_TFLOPS[args.dtype] / world, 4),     # direct FLOP count; issued: roofline.step_frac_issued
    }
    out.update(extra)
    rl = roofline_block(table, dom, worst, prof, args.dtype, B, world, step_ms=1e3 * dt / args.steps)
    if rl:
        out["roofline"] = rl
    if other is not None:
        out["bf16" if other_dt == "bf16" else "fp32"] = other
    if rows:
        out["rows"] = rows
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
