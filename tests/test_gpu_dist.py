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


def _run(world, out, gb=32, steps=3):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out, str(gb), str(steps)],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    return np.load(out)


def test_two_ranks_equal_one(lib_built, tmp_path):
    one = _run(1, str(tmp_path / "one.npz"))
    two = _run(2, str(tmp_path / "two.npz"))
    # rank 0's loss scalars are its SHARD means; gradients and weights are global
    g1, g2 = one["grads"], two["grads"] / 2.0                 # all-reduce(sum); 1/world lives in the Adam kernel
    assert np.linalg.norm(g1 - g2) <= 2e-3 * np.linalg.norm(g1)
    p1, p2 = one["params"], two["params"]
    # Adam turns rounding-level differences of near-zero gradients into full-lr moves: compare to the 3-step movement
    assert np.linalg.norm(p1 - p2) <= 5e-2 * np.linalg.norm(p1 - _init_params())
    assert np.all(np.isfinite(two["losses"]))


def _init_params():
    import torch
    from split_vae_amd.model import LGVae
    return LGVae(128, 128, image_shape=[-1, 32, 32, 3], dtype="f32", device="cuda", seed=3).flat.cpu().numpy()
