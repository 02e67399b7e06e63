"""Data-parallel step on the device: two ranks, each with half of the global batch, sharing
the one GPU of the test box through the gloo backend (RCCL needs a device per rank; the step, the phase
split, the bucketed asynchronous all-reduce of the flat gradient buffer and the 1/world factor inside the
Adam kernel are the same code with either backend).  The result must equal the single-process step on the
whole batch: the per-sample RNG is keyed by the global sample index and the loss is a batch mean."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, out, gb=32, steps=3, backend="gloo", force=False, config="svhn32_f32", deterministic=False, buckets=None, mode=None, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r) if backend == "nccl" else "0", WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SV_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if force:
            env["SV_DIST_FORCE"] = "1"
        if deterministic:
            env["SV_DETERMINISTIC"] = "1"         # fixed-order reductions in the worker (include/splitvae.h: sv_set_deterministic)
        if buckets == 3:
            env["SV_DP_TWO_BUCKETS_MAX"] = "0"    # the three-bucket schedule of shards above 256 images (trainer.train_step) at this small shard
        elif buckets == 2:
            env["SV_DP_TWO_BUCKETS_MAX"] = "1024"
        if mode:
            env["SV_DP_MODE"] = mode              # 'auto' (default: events with peers, single on one rank) | 'events' (one backward call + bucket events) | 'overlap' (the phase split of rounds 1-4) | 'single'
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out, str(gb), str(steps), config],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    return np.load(out)


@pytest.mark.parametrize("mode", ["events", "overlap"])
@pytest.mark.parametrize("buckets", [2, 3])
def test_two_ranks_equal_one(lib_built, tmp_path, buckets, mode):
    one = _run(1, str(tmp_path / "one.npz"))
    two = _run(2, str(tmp_path / "two.npz"), buckets=buckets, mode=mode)
    # rank 0's loss scalars are its SHARD means; gradients and weights are global
    g1, g2 = one["grads"], two["grads"] / 2.0                 # all-reduce(sum); 1/world lives in the Adam kernel
    assert np.linalg.norm(g1 - g2) <= 1e-2 * np.linalg.norm(g1)          # (bounds: see test_one_rank_through_the_rccl_path_equals_the_plain_step)
    p1, p2 = one["params"], two["params"]
    # Adam turns rounding-level differences of near-zero gradients into full-lr moves: compare to the 3-step movement
    assert np.linalg.norm(p1 - p2) <= 1e-1 * np.linalg.norm(p1 - _init_params())
    assert np.all(np.isfinite(two["losses"]))


def test_two_ranks_equal_one_at_config_4s_shard(lib_built, tmp_path):
    """Config 4's per-GPU shape (BASELINE.json configs[3]: CelebA-64, bf16, global batch 512 = 64 images per GPU at N = 8): two ranks of
    64 images against one process with 128.  At 64 images the plan takes its small-launch tile rules, the banded input gradients
    and the shard-size-dependent stream placement, none of which the SVHN-32 fp32 case above reaches.  bf16: the two halves round
    their activations exactly as the whole batch does (per-image arithmetic), only the batch reductions reorder."""
    one = _run(1, str(tmp_path / "one.npz"), gb=128, steps=1, config="celeba64_bf16")
    two = _run(2, str(tmp_path / "two.npz"), gb=128, steps=1, config="celeba64_bf16")
    g1, g2 = one["grads"], two["grads"] / 2.0
    assert np.linalg.norm(g1 - g2) <= 1e-2 * np.linalg.norm(g1)
    assert np.all(np.isfinite(two["losses"]))


def _bucket_probe(extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bucket_probe.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("BUCKET_SNAPSHOT_ERR")][-1]
    return [float(v) for v in line.split()[1:]]


def test_buckets_wait_for_the_side_stream(lib_built):
    """The event path's dependency, made observable on ONE GPU (a world-1 all-reduce is the identity on a buffer that is final by the time anyone
    reads it, so the one-rank RCCL tests cannot see a missing dependency): the weight-gradient side stream is held back for 5 ms at its first use of
    the step (SV_PYTEST_SIDE_DELAY_US), and a fresh stream ordered ONLY by sv_lgvae_bucket_wait snapshots each bucket's gradient range.  The snapshot
    must equal the final gradients bit for bit -- the stream waited for the side stream's part of the bucket -- and with the side streams' events
    dropped on purpose (SV_PYTEST_BUCKET_SKIP_SIDE: negative control) the decoders' snapshot must NOT (measured: taken 1.2 ms into a 6.2 ms step, all of
    the side stream's weight gradients missing): the probe can see the failure it guards against.  (The library's default of three hardware queues; a
    probe stream that happens to share a queue with the held-back side stream inherits its order, which is why only the first bucket is asserted on.)"""
    knobs = {"SV_PYTEST_SIDE_DELAY_US": "5000"}
    good = _bucket_probe(knobs)
    assert good == [0.0, 0.0, 0.0], good
    bad = _bucket_probe(dict(knobs, SV_PYTEST_BUCKET_SKIP_SIDE="1"))
    assert bad[0] > 0.1, bad


def test_two_ranks_equal_one_with_a_late_side_stream(lib_built, tmp_path):
    """... and end to end: two ranks over gloo with the side stream held back, against the single-process step."""
    one = _run(1, str(tmp_path / "one.npz"), steps=1)
    two = _run(2, str(tmp_path / "two.npz"), steps=1, mode="events", extra_env={"SV_PYTEST_SIDE_DELAY_US": "5000"})
    g1, g2 = one["grads"], two["grads"] / 2.0
    assert np.linalg.norm(g1 - g2) <= 1e-2 * np.linalg.norm(g1)


@pytest.mark.parametrize("backend", ["nccl", "sv_comm"])
def test_two_ranks_over_rccl_equal_one(lib_built, tmp_path, backend):
    """The same equivalence with the production backends: one rank per GPU, `nccl` (= RCCL over xGMI through
    torch.distributed) and `sv_comm` (RCCL through the C ABI's own communicator).  Needs two devices."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL wants one device per rank; this box has %d" % torch.cuda.device_count())
    one = _run(1, str(tmp_path / "one.npz"))
    two = _run(2, str(tmp_path / "two.npz"), backend=backend)
    g1, g2 = one["grads"], two["grads"] / 2.0
    # the original bounds: never observed to flake on a multi-GPU box (none was ever available), so not loosened on a guess
    assert np.linalg.norm(g1 - g2) <= 2e-3 * np.linalg.norm(g1)
    assert np.linalg.norm(one["params"] - two["params"]) <= 5e-2 * np.linalg.norm(one["params"] - _init_params())


@pytest.mark.parametrize("buckets", [2, 3])
@pytest.mark.parametrize("backend", ["nccl", "sv_comm"])
def test_one_rank_through_the_rccl_path_equals_the_plain_step(lib_built, tmp_path, backend, buckets):
    """The production data-parallel path on the ONE device of this box (SV_DIST_FORCE=1): process group on the `nccl` backend
    (= RCCL) resp. the library's own communicator, the phase split of train_step -- both schedules: three gradient buckets (shards above
    256 images) and two (decoders, then the encoders as one contiguous range) -- all-reduced asynchronously over a world of one rank,
    1/world inside Adam -- against the plain single-call step.  Primary form: fixed-order reductions in both runs (SV_DETERMINISTIC=1 in
    the workers), the original tight bounds."""
    one = _run(1, str(tmp_path / "one.npz"), deterministic=True)
    dp = _run(1, str(tmp_path / "dp.npz"), backend=backend, force=True, deterministic=True, buckets=buckets)
    g1, g2 = one["grads"], dp["grads"]
    assert np.linalg.norm(g1 - g2) <= 1e-3 * np.linalg.norm(g1)
    assert np.linalg.norm(one["params"] - dp["params"]) <= 1e-2 * np.linalg.norm(one["params"] - _init_params())
    assert np.allclose(one["losses"], dp["losses"], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("backend", ["nccl", "sv_comm"])
def test_one_rank_through_the_rccl_path_default_summation_order(lib_built, tmp_path, backend):
    """The same on the default (atomics-ordered) path, looser."""
    one = _run(1, str(tmp_path / "one.npz"))
    dp = _run(1, str(tmp_path / "dp.npz"), backend=backend, force=True)
    g1, g2 = one["grads"], dp["grads"]
    # two runs of the SAME step differ by the order of their fp32 atomics; after an update that occasionally (about one run in
    # twenty) puts a ReLU unit's pre-activation within that noise of zero, and its gate -- with that unit's share of the
    # gradients -- falls differently in the two runs: a few thousandths of the gradient's norm (tests/test_gpu_gm.py has the
    # analysis).  The bounds leave room for that, not for a missing bucket.
    assert np.linalg.norm(g1 - g2) <= 1e-2 * np.linalg.norm(g1)
    assert np.linalg.norm(one["params"] - dp["params"]) <= 1e-1 * np.linalg.norm(one["params"] - _init_params())
    assert np.allclose(one["losses"], dp["losses"], rtol=1e-3, atol=1e-3)


def test_sv_comm_single_rank_on_the_device(lib_built):
    """sv_comm_* end to end on the one GPU of this box: RCCL is found at run time, a world-1 communicator is created on the
    current device, all-reduce(sum) over one rank leaves the buffer as it was, asynchronously on the caller's stream."""
    import ctypes as C
    import torch
    from split_vae_amd import _lib
    lib = _lib.load()
    raw = C.create_string_buffer(128)
    assert lib.sv_comm_unique_id(raw) == 0 and any(raw.raw)
    h = C.c_void_p()
    assert lib.sv_comm_init(raw, 0, 1, C.byref(h)) == 0 and h.value
    assert lib.sv_comm_init(raw, 1, 1, C.byref(C.c_void_p())) == _lib.STATUS_BADARG        # rank outside [0, world)
    buf = torch.randn(1 << 20, device="cuda")
    want = buf.clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    assert lib.sv_comm_allreduce(h, C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(st.cuda_stream)) == 0
    b, e = (C.c_int64 * 2)(0, 4096), (C.c_int64 * 2)(1024, 8192)
    assert lib.sv_comm_allreduce_ranges(h, C.c_void_p(buf.data_ptr()), b, e, 2, C.c_void_p(st.cuda_stream)) == 0
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    assert lib.sv_comm_allreduce(h, None, 4, None) == _lib.STATUS_BADARG
    assert lib.sv_comm_destroy(h) == 0


def test_bench_refuses_more_ranks_than_gpus(lib_built):
    """`python bench.py --gpus N` spawns N ranks itself; with fewer devices than N it must fail loudly, never report
    a 1-GPU number as an N-GPU one."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
    # a rank count that disagrees with the launcher's WORLD_SIZE is refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def _init_params():
    import torch
    from split_vae_amd.model import LGVae
    return LGVae(128, 128, image_shape=[-1, 32, 32, 3], dtype="f32", device="cuda", seed=3).flat.cpu().numpy()
