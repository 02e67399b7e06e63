"""CPU, world_size = 2 over gloo: the data-parallel bookkeeping of split_vae_amd.dist (equal
contiguous shards, per-sample RNG keyed by global index, bucketed asynchronous all-reduce of the
flat gradient buffer in backward-completion order, 1/world applied in Adam) reproduces the
single-process full-batch gradient and update.  Per-shard gradients come from the oracle (the
HIP step cannot run without a GPU); everything between "gradient produced" and "weights
updated" is the product's distributed code."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from oracle import np_ref, torch_ref

H, GB, BETA, PATCH = 32, 4, 40.0, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_batch():
    from split_vae_amd import data
    x = data.synthetic_images(GB, H, H, seed=0, device="cpu").numpy()
    perm = np.stack([np.random.Generator(np.random.PCG64([7, i])).permutation((H // PATCH) ** 2) for i in range(GB)])
    eps = np.stack([np.random.Generator(np.random.PCG64([9, i])).standard_normal((2, 128)) for i in range(GB)], 1)
    return x, perm, eps          # eps[2, GB, 128]


def _flat(tensors, table, n):
    f = torch.zeros(n, dtype=torch.float64)
    for (name, off, shape), t in zip(table, tensors):
        f[off:off + t.numel()] = t.detach().double().flatten()
    return f


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import ctypes as C
    from split_vae_amd import _lib, data, dist, ops
    r, lr_, w = dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.set_num_threads(2)
    desc = _lib.LGVaeDesc(GB // world, H, H, 128, 128, 1, BETA)
    table = ops.param_table(desc)
    n = _lib.load().sv_lgvae_param_count(C.byref(desc))
    lo, hi = dist.shard_bounds(GB, rank, world)
    # shard inputs drawn independently by global sample index == slices of the global batch
    x = data.synthetic_images(hi - lo, H, H, seed=0, device="cpu", sample_offset=lo).numpy()
    gx, gperm, geps = _global_batch()
    assert np.array_equal(x, gx[lo:hi])
    images = torch_ref.scramble_batch(x, gperm[lo:hi], PATCH).double()
    params = np_ref.glorot_init(H, H, seed=3)
    ref = torch_ref.RefTrainer(params, BETA, dtype=torch.float64)
    _, _, grads = ref.grads(images, geps[0, lo:hi], geps[1, lo:hi])
    flat = _flat(grads, table, n)
    red = dist.GradReducer(table, n)
    # same order as the trainer: decoders -> encoder heads -> encoder convs, all asynchronous
    for bucket in ("decoders", "enc_heads", "enc_convs"):
        red.launch(flat, bucket)
    red.wait()
    assert red.grad_scale == 1.0 / world
    if rank == 0:
        torch.save(flat * red.grad_scale, os.path.join(out_dir, "avg.pt"))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch(lib_built, tmp_path):
    import ctypes as C
    from split_vae_amd import _lib, ops
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    avg = torch.load(os.path.join(str(tmp_path), "avg.pt"))
    desc = _lib.LGVaeDesc(GB, H, H, 128, 128, 1, BETA)
    table = ops.param_table(desc)
    n = _lib.load().sv_lgvae_param_count(C.byref(desc))
    gx, gperm, geps = _global_batch()
    images = torch_ref.scramble_batch(gx, gperm, PATCH).double()
    ref = torch_ref.RefTrainer(np_ref.glorot_init(H, H, seed=3), BETA, dtype=torch.float64)
    _, _, grads = ref.grads(images, geps[0], geps[1])
    full = _flat(grads, table, n)
    assert float((avg - full).abs().max()) <= 1e-9 * float(full.abs().max())
