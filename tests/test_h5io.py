"""Keras HDF5 weight container (split_vae_amd/h5io.py; vae/trainer.py:421 model.save_weights('models/<run>.h5')):
CPU round trip through libhdf5 and, when the HDF5 command-line tools are present, an independent read-back with h5ls /
h5dump; on the device, save -> load -> identical forward for LGVae and LGGMVae in both containers."""
import os
import shutil
import subprocess
import types

import numpy as np
import pytest

from split_vae_amd import h5io

needs_hdf5 = pytest.mark.skipif(not h5io.available(), reason="libhdf5 not present in this image")


def _layers(seed=0):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    return [("encoder", [("lg_vae/encoder/conv2d/kernel:0", f(6, 6, 3, 32)), ("lg_vae/encoder/conv2d/bias:0", f(32)),
                         ("lg_vae/encoder/dense/kernel:0", f(40, 8))]),
            ("decoder_1", [("lg_vae/decoder_1/dense_5/kernel:0", f(8, 16)), ("lg_vae/decoder_1/dense_5/bias:0", f(16))])]


@needs_hdf5
def test_keras_h5_round_trip(tmp_path):
    layers = _layers()
    path = h5io.save_keras_weights(str(tmp_path / "w.h5"), layers)
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"              # the HDF5 signature
    back = h5io.load_keras_weights(path)
    assert [n for n, _ in back] == [n for n, _ in layers]
    for (_, ws), (_, wb) in zip(layers, back):
        assert [n for n, _ in wb] == [n for n, _ in ws]
        for (_, a), (_, b) in zip(ws, wb):
            assert b.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b)
    with pytest.raises(IOError):
        h5io.load_keras_weights(str(tmp_path / "missing.h5"))


@needs_hdf5
@pytest.mark.skipif(not (shutil.which("h5dump") or os.path.exists("/opt/conda/bin/h5dump")), reason="HDF5 tools not present")
def test_keras_h5_layout_read_by_hdf5_tools(tmp_path):
    """An independent reader (the HDF5 project's own h5dump) sees the Keras layout: root attributes layer_names / backend /
    keras_version, per-layer weight_names, nested float32 datasets with the stored values."""
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    layers = _layers(1)
    path = h5io.save_keras_weights(str(tmp_path / "w.h5"), layers)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/conda/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    hdr = subprocess.run([h5dump, "-A", path], capture_output=True, text=True, env=env, check=True).stdout
    for token in ('ATTRIBUTE "layer_names"', 'ATTRIBUTE "backend"', 'ATTRIBUTE "keras_version"', '"tensorflow"', '"2.2.4-tf"',
                  'GROUP "encoder"', 'GROUP "decoder_1"', 'ATTRIBUTE "weight_names"', '"lg_vae/encoder/conv2d/kernel:0"',
                  'DATASET "kernel:0"', "H5T_IEEE_F32LE", "( 6, 6, 3, 32 )"):
        assert token in hdr, token
    one = subprocess.run([h5dump, "-d", "/decoder_1/lg_vae/decoder_1/dense_5/bias:0", "-m", "%.9g", path], capture_output=True,
                         text=True, env=env, check=True).stdout
    data = one[one.index("DATA {"):]
    vals = [float(t) for t in data.replace("\n", " ").replace(",", " ").split() if t.replace(".", "").replace("-", "").replace("e", "").replace("+", "").isdigit() and ("." in t or "e" in t)]
    want = layers[1][1][1][1]
    assert len(vals) == want.size and np.allclose(np.array(vals, np.float32), want, rtol=0, atol=1e-7)


def test_keras_weight_names_follow_the_uid_convention(lib_built):
    """Names under which the variables go into the file (cosmetic; loading is by order): sub-models encoder, encoder_1,
    decoder, decoder_1; per-class uid counters in creation order; <model>/<layer>/<inner>/<kernel|bias>:0."""
    import ctypes as C
    from split_vae_amd import _lib, ops
    from split_vae_amd.model import LGVae
    desc = _lib.LGVaeDesc(1, 32, 32, 128, 128, _lib.SV_F32, 1.0)
    table = ops.param_table(desc)
    stub = types.SimpleNamespace(keras_names=lambda: [n + ":0" for n, _, _ in table], KERAS_MODEL_NAME="lg_vae",
                                 _keras_kind=lambda n: LGVae._keras_kind(None, n))
    layers = LGVae.keras_h5_layers(stub)
    assert [n for n, _ in layers] == ["encoder", "encoder_1", "decoder", "decoder_1"]
    assert [len(w) for _, w in layers] == [10, 10, 10, 10]
    assert layers[0][1][:2] == ["lg_vae/encoder/conv2d/kernel:0", "lg_vae/encoder/conv2d/bias:0"]
    assert layers[0][1][6] == "lg_vae/encoder/dense/kernel:0" and layers[0][1][8] == "lg_vae/encoder/dense_1/kernel:0"
    assert layers[1][1][0] == "lg_vae/encoder_1/conv2d_3/kernel:0" and layers[1][1][8] == "lg_vae/encoder_1/dense_3/kernel:0"
    assert layers[2][1][0] == "lg_vae/decoder/dense_4/kernel:0" and layers[2][1][2] == "lg_vae/decoder/conv2d_6/kernel:0"
    assert layers[3][1][0] == "lg_vae/decoder_1/dense_5/kernel:0" and layers[3][1][-1] == "lg_vae/decoder_1/conv2d_13/bias:0"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lgvae", "lggmvae"])
@pytest.mark.parametrize("ext", [".h5", ".npz"])
def test_save_load_identical_forward(lib_built, tmp_path, kind, ext):
    """A12: save_weights -> a fresh model -> load_weights -> bit-identical variables and forward outputs (same eps)."""
    import torch
    from split_vae_amd import data
    from split_vae_amd.augmentation import Augmentator
    from split_vae_amd.gm import LGGMVae
    from split_vae_amd.model import LGVae
    if ext == ".h5" and not h5io.available():
        pytest.skip("libhdf5 not present")
    H, B = 32, 4
    mk = (lambda seed: LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="f32", device="cuda", seed=seed)) if kind == "lgvae" else \
         (lambda seed: LGGMVae(128, 128, [-1, H, H, 3], 30, 0.4, dtype="f32", device="cuda", seed=seed))
    a, b = mk(11), mk(12)
    for v in a.trainable_variables:                                  # biases away from zero so every array is distinctive
        if v.dim() == 1:
            v.add_(torch.randn_like(v) * 0.05)
    assert any(not torch.equal(x, y) for x, y in zip(a.trainable_variables, b.trainable_variables))
    path = a.save_weights(str(tmp_path / ("w" + ext)))
    assert path.endswith(ext) and os.path.exists(path)
    b.load_weights(path)
    assert len(a.trainable_variables) == (40 if kind == "lgvae" else 54)
    for n, x, y in zip(a.keras_names(), a.trainable_variables, b.trainable_variables):
        assert torch.equal(x, y), n
    img = Augmentator("scramble", size=4, seed=1).augment(data.synthetic_images(B, H, H, seed=3, device="cuda"))
    eps = (torch.randn(B, 128, device="cuda"), torch.randn(B, 128, device="cuda"))
    kw = {} if kind == "lgvae" else {"noise": (torch.rand(B, 30, device="cuda") * 0.9 + 0.05, None, None)}
    oa, ob = a(img, eps=eps, **kw), b(img, eps=eps, **kw)
    for x, y in zip(oa, ob):            # identical weights; the latents pass through split-K fp32 atomics (last-bit run-to-run noise)
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-5)
    if ext == ".h5":                                                 # a file of another architecture is refused, not mis-assigned
        other = LGVae(64, 64, image_shape=[-1, H, H, 3], dtype="f32", device="cuda") if kind == "lgvae" else \
            LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="f32", device="cuda")
        with pytest.raises(ValueError):
            other.load_weights(path)
