"""GPU: the pieces the native SPLIT-SPAIR step adds (dense_f32.hip, tape.hip) against torch fp64 on the same operands.  The assembled
step is under the oracle in tests/test_gpu_spair_model.py (train_step takes the native launch sequence by default)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(lib_built):
    assert torch.cuda.is_available()
    from split_vae_amd import ops as o
    return o


@pytest.mark.parametrize("M,K,N", [(32, 6912, 1024), (32, 1024, 6912), (512, 177, 64), (512, 64, 1), (512, 68, 128), (32, 500, 4), (512, 4096, 128),
                                   (7, 13, 5), (65, 33, 129)])
def test_dense_f32_matches_fp64(ops, M, K, N):
    """tf.keras.layers.Dense forward / input gradient / weight + bias gradient for SPLIT-SPAIR's shapes (spair/spair.py:135-154, :185-202,
    :424-467) and ragged ones: exact-fp32 MFMA vs fp64, 2e-6 relative (split-K sums reorder fp32 adds)."""
    g = torch.Generator().manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm().clamp_min(1e-30))
    y = ops.dense_f32_fwd(xd, wd, bd)
    assert rel(y, x.double() @ w.double() + b.double()) < 2e-6
    yr = ops.dense_f32_fwd(xd, wd, bd, act="relu")
    assert rel(yr, torch.relu(x.double() @ w.double() + b.double())) < 2e-6
    dx = ops.dense_f32_dgrad(dyd, wd)
    assert rel(dx, dy.double() @ w.double().T) < 2e-6
    acc = torch.ones(M, K, device="cuda")
    ops.dense_f32_dgrad(dyd, wd, out=acc)
    assert rel(acc, 1.0 + dy.double() @ w.double().T) < 2e-6
    dw, db = ops.dense_f32_wgrad(xd, dyd)
    assert rel(dw, x.double().T @ dy.double()) < 2e-6
    assert rel(db, dy.double().sum(0)) < 2e-6
    # row pitches wider than the logical width (tape tensors are padded to 4 floats)
    xp = torch.zeros(M, K + 3, device="cuda"); xp[:, :K] = xd
    assert rel(ops.dense_f32_fwd(xp[:, :K], wd, bd), x.double() @ w.double() + b.double()) < 2e-6
