"""Data parallelism for the SPLIT-VAE step: one process per GPU, gradients averaged with
all-reduce over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm; "gloo" on CPU tests).

The reference has no distributed code.  The step shards over the batch axis: every
per-image quantity is independent and the loss is a batch mean (vae/trainer.py:13,:127-128), so
the global-batch gradient is the mean of equal-sized shard gradients.  The flat fp32 gradient
buffer is reduced in buckets that follow backward completion order, each launched on RCCL's own
stream as soon as the phase that fills it has been enqueued, so the transfers overlap the rest of
the backward pass.  The 1/world factor is applied inside the Adam kernel (grad_scale).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run."""
    from . import configure_hw_queues
    configure_hw_queues()                            # main / weight-gradient / communication queue (split_vae_amd/__init__.py)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SV_DIST_FORCE=1: take the distributed path with ONE rank too (process group, phase split, bucketed all-reduce over a world
    # of 1) -- how the RCCL path is exercised on a single-GPU box (tests/test_gpu_dist.py)
    if (world > 1 or os.environ.get("SV_DIST_FORCE")) and not dist.is_initialized():
        if backend is None:      # SV_DIST_BACKEND=gloo: several ranks sharing one GPU (tests); RCCL wants one device per rank
            backend = os.environ.get("SV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend in ("nccl", "sv_comm"):
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        if backend == "sv_comm":  # gradients over the library's own RCCL communicator (sv_comm_*); gloo only as control plane
            backend = "gloo"
            os.environ["SV_DIST_BACKEND"] = "sv_comm"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(global_batch, rank, world):
    """Contiguous equal slices of the global batch; sample_offset keys the per-sample RNG so a
    1-GPU run and an N-GPU run draw identical eps / permutations for the same global sample."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d (equal shards are what makes "
                         "mean-of-shard-gradients == global-batch gradient)" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def param_buckets(param_table, n_params):
    """Element ranges of the flat buffer in backward-completion order:
    [decoder_x + decoder_x_hat] -> [both encoder heads (e4_mean, e4_sd)] -> [encoder convs]."""
    def span(pred):
        rs = []
        for i, (name, off, shape) in enumerate(param_table):
            if not pred(name):
                continue
            end = param_table[i + 1][1] if i + 1 < len(param_table) else n_params
            if rs and rs[-1][1] == off:
                rs[-1] = (rs[-1][0], end)
            else:
                rs.append((off, end))
        return rs
    dec = span(lambda n: n.startswith("decoder"))
    heads = span(lambda n: n.startswith("encoder") and "/e4_" in n)
    convs = span(lambda n: n.startswith("encoder") and "/e4_" not in n)
    covered = sum(e - b for rs in (dec, heads, convs) for b, e in rs)
    assert covered == n_params, (covered, n_params)
    # "encoders" = heads + convs as ONE bucket (small shards: trainer.train_step launches two buckets instead of three -- every all-reduce call
    # costs ~30 us of host time, which a 0.6 ms step cannot hide: profiles/r04_dp_one_rank.txt)
    return {"decoders": dec, "enc_heads": heads, "enc_convs": convs, "encoders": span(lambda n: n.startswith("encoder"))}


def make_reducer(param_table, n_params):
    """GradReducer over torch.distributed (default: backend nccl = RCCL) or, with SV_DIST_BACKEND=sv_comm, NativeGradReducer
    over the C ABI's own RCCL communicator."""
    if os.environ.get("SV_DIST_BACKEND") == "sv_comm" and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("SV_DIST_FORCE")):
        return NativeGradReducer(param_table, n_params)
    return GradReducer(param_table, n_params)


class NativeGradReducer:
    """The same bucketed asynchronous all-reduce through sv_comm_* (include/splitvae.h): one RCCL communicator owned by
    libsplitvae_hip.so, one dedicated HIP stream; a bucket = one RCCL group over its contiguous parameter runs, forked off
    the compute stream by an event when the phase that filled it has been enqueued, joined before Adam.  torch.distributed
    (any backend) only carries the 128-byte rendezvous id."""

    def __init__(self, param_table, n_params, group=None):
        import ctypes as C
        from . import _lib
        self.C, self.lib = C, _lib.load()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.buckets = param_buckets(param_table, n_params)
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            raw = C.create_string_buffer(128)
            _lib.check(self.lib.sv_comm_unique_id(raw), "sv_comm_unique_id")
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        if dist.get_backend(group) == "nccl":
            idbuf = idbuf.cuda()
        dist.broadcast(idbuf, src=0, group=group)
        self.handle = C.c_void_p()
        _lib.check(self.lib.sv_comm_init(C.c_char_p(bytes(idbuf.cpu().numpy().tobytes())), self.rank, self.world,
                                         C.byref(self.handle)), "sv_comm_init")
        # RCCL runs ON this stream: a stream of its own (on the library's low-priority shared stream 1 the one-rank step went 1.74 -> 2.96 ms: profiles/r06_dp_ab.txt);
        # the torch-process-group reducer below only HANDS OVER on a stream (the collective runs on torch's), and takes the library's
        self.stream = torch.cuda.Stream()
        self._ranges = {}
        for k, spans in self.buckets.items():
            n = len(spans)
            self._ranges[k] = ((C.c_int64 * n)(*[b for b, _ in spans]), (C.c_int64 * n)(*[e for _, e in spans]), n)
        self._dirty = False
        self.force = True
        self.mode = os.environ.get("SV_DP_MODE", "auto")                # 'auto' (default: 'events' with peers, 'single' on one rank) | 'events' (one backward call, buckets picked up by their events) | 'overlap' (phase split) | 'single'

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def launch(self, flat, bucket, after=None):
        """after=(plan, k): the communication stream waits for the plan's bucket-k events (sv_lgvae_bucket_wait) instead of for everything
        enqueued on the compute stream so far -- the whole backward is one call and the bucket starts as soon as ITS kernels are done."""
        from . import _lib
        if after is not None:
            after[0].bucket_wait(after[1], self.stream)
        else:
            self.stream.wait_stream(torch.cuda.current_stream())       # the phase that filled the bucket
        b, e, n = self._ranges[bucket]
        _lib.check(self.lib.sv_comm_allreduce_ranges(self.handle, self.C.c_void_p(flat.data_ptr()), b, e, n,
                                                     self.C.c_void_p(self.stream.cuda_stream)), "sv_comm_allreduce_ranges")
        self._dirty = True

    def launch_all(self, flat):
        """mode 'single': the whole flat buffer as one all-reduce."""
        from . import _lib
        self.stream.wait_stream(torch.cuda.current_stream())
        _lib.check(self.lib.sv_comm_allreduce(self.handle, self.C.c_void_p(flat.data_ptr()), flat.numel(),
                                              self.C.c_void_p(self.stream.cuda_stream)), "sv_comm_allreduce")
        self._dirty = True

    def wait(self):
        ev = getattr(self, "wait_events", None)      # bench.py's diagnosis: (before, after) event pairs on the compute stream = the time it really waits
        if ev is not None:
            a = torch.cuda.Event(enable_timing=True); a.record()
        if self._dirty:
            torch.cuda.current_stream().wait_stream(self.stream)        # no host sync
            self._dirty = False
        if ev is not None:
            b = torch.cuda.Event(enable_timing=True); b.record(); ev.append((a, b))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                torch.cuda.synchronize()
                self.lib.sv_comm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def library_side_stream(index):
    """The library's shared side stream `index` of the current device (include/splitvae.h: sv_side_stream) as a torch stream.  The data-parallel step uses one
    weight-gradient side stream (index 0); index 1 is free for the communication hand-over."""
    import ctypes as C
    from . import _lib
    h = C.c_void_p()
    _lib.check(_lib.load().sv_side_stream(int(index), C.byref(h)), "sv_side_stream")
    return torch.cuda.ExternalStream(h.value, device=torch.cuda.current_device())


class GradReducer:
    """Bucketed asynchronous all-reduce(sum) of a flat gradient buffer."""

    def __init__(self, param_table, n_params, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(os.environ.get("SV_DIST_FORCE")) and dist.is_initialized()
        self.group = group
        self.buckets = param_buckets(param_table, n_params)
        self._pending = []
        self.mode = os.environ.get("SV_DP_MODE", "auto")                # 'auto' (default: 'events' with peers, 'single' on one rank) | 'events' (one backward call, buckets picked up by their events) | 'overlap' (phase split) | 'single'

    def launch_all(self, flat):
        """mode 'single': the whole flat buffer as one all-reduce."""
        if self.world == 1 and not self.force:
            return
        self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def launch(self, flat, bucket, after=None):
        """Enqueue the all-reduce of one bucket (call right after the phase that produced it).  after=(plan, k): the collective is ordered
        behind the plan's bucket-k events only (sv_lgvae_bucket_wait on an auxiliary stream that is `current` while the collective is enqueued:
        torch's process groups order a collective after the current stream), not behind the rest of the backward the compute stream already holds."""
        if self.world == 1 and not self.force:
            return
        if after is None:
            for b, e in self.buckets[bucket]:
                self._pending.append(dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if getattr(self, "_aux", None) is None:
            self._aux = library_side_stream(1)     # (not a stream of our own: HIP maps streams onto hardware queues in creation order, csrc/streams.hip)
        after[0].bucket_wait(after[1], self._aux)
        with torch.cuda.stream(self._aux):
            for b, e in self.buckets[bucket]:
                self._pending.append(dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Make the compute stream wait for every outstanding bucket (no host sync on nccl)."""
        ev = getattr(self, "wait_events", None)      # bench.py's diagnosis: (before, after) event pairs on the compute stream = the time it really waits
        if ev is not None:
            a = torch.cuda.Event(enable_timing=True); a.record()
        for w in self._pending:
            w.wait()
        self._pending = []
        if ev is not None:
            b = torch.cuda.Event(enable_timing=True); b.record(); ev.append((a, b))
