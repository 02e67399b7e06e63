"""CPU: host-side mirror of the reference's Python surface (flags, dotdict, optimizer schedule,
augmentor selection, data-parallel bookkeeping) -- everything that needs no device."""
import numpy as np
import pytest

from oracle import np_ref


def test_cli_flags_match_reference():
    """vae/main.py:15-31: same names, types and defaults."""
    from split_vae_amd.main import build_parser
    a = build_parser().parse_args([])
    want = dict(viz=False, global_latent_dims=128, local_latent_dims=128, learning_rate=1e-4, beta=40, dataset='svhn',
                training_steps=1000000, batch_size=64, patch_size=1, augmentation='scramble', no_label=False,
                model='lgvae', y_size=30, tau=0.4, alpha=40, allow_growth=False)
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = build_parser().parse_args("--beta 120 --patch_size 8 --dataset celeba64 -no_label".split())   # README.md:37
    assert (b.beta, b.patch_size, b.dataset, b.no_label) == (120.0, 8, "celeba64", True)
    assert isinstance(b.beta, float) and isinstance(b.patch_size, int)


def test_dotdict_missing_key_is_none():
    from split_vae_amd.utils import dotdict
    d = dotdict({"a": 1})
    assert d.a == 1 and d.bg_model is None        # vae/utils.py:3-7 (dict.get semantics)
    d.b = 2
    assert d["b"] == 2
    del d.a
    assert "a" not in d


def test_augmentator_selection():
    from split_vae_amd.augmentation import Augmentator
    a = Augmentator("scramble", size=8)
    assert a.augment == a.scramble and a.size == 8
    assert Augmentator("no_op").augment("x") == "x"
    for t in ("mix_scramble", "blur", "high_low_pass"):
        with pytest.raises(NotImplementedError):
            Augmentator(t)
    with pytest.raises(ValueError):
        Augmentator("nope")


def test_exponential_decay_schedule():
    from split_vae_amd.optimizer import Adam, ExponentialDecay
    s = ExponentialDecay(1e-4, decay_steps=1000000, decay_rate=0.4, staircase=True)      # vae/main.py:67
    assert s(0) == 1e-4 and s(999999) == 1e-4 and abs(s(1000000) - 4e-5) < 1e-18 and abs(s(2500000) - 1.6e-5) < 1e-18
    o = Adam(learning_rate=s)
    assert o.lr() == 1e-4 and (o.beta_1, o.beta_2, o.epsilon) == (0.9, 0.999, 1e-7)      # Keras defaults


def test_shard_bounds_and_buckets(lib_built):
    import ctypes as C
    from split_vae_amd import _lib, dist, ops
    assert dist.shard_bounds(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        dist.shard_bounds(100, 0, 8)
    desc = _lib.LGVaeDesc(8, 64, 64, 128, 128, 1, 1.0)
    table = ops.param_table(desc)
    n = _lib.load().sv_lgvae_param_count(C.byref(desc))
    b = dist.param_buckets(table, n)
    # decoders are one contiguous tail; heads and convs interleave per encoder
    assert len(b["decoders"]) == 1 and b["decoders"][0][1] == n
    assert len(b["enc_heads"]) == 2 and len(b["enc_convs"]) == 2
    cover = np.zeros(n, int)
    for rs in b.values():
        for lo, hi in rs:
            cover[lo:hi] += 1
    assert np.all(cover == 1)
    sizes = {k: sum(hi - lo for lo, hi in v) for k, v in b.items()}
    assert sizes["enc_heads"] > 4_000_000 and sizes["decoders"] > 4_000_000 and sizes["enc_convs"] < 500_000


def test_synthetic_data_domain_and_shard_invariance():
    from split_vae_amd import data
    full = data.synthetic_images(8, 32, 32, seed=0, device="cpu")
    part = data.synthetic_images(3, 32, 32, seed=0, device="cpu", sample_offset=4)
    assert np.array_equal(part.numpy(), full[4:7].numpy())
    v = np.unique(np.round((full.numpy() + 1) * 127.5).astype(int))
    assert v.min() >= 0 and v.max() <= 255 and full.dtype.is_floating_point      # vae/data.py:52 domain
    assert np.allclose((np.round((full.numpy() + 1) * 127.5) / 127.5 - 1), full.numpy(), atol=1e-6)


def test_cli_selects_lggmvae_schedule():
    """vae/main.py:66-69: --model lggmvae uses ExponentialDecay(lr, 1e6, 0.4, staircase=True) and the y_size/tau/alpha flags."""
    from split_vae_amd.main import build_parser
    from split_vae_amd.optimizer import ExponentialDecay
    a = build_parser().parse_args(["--model", "lggmvae", "--patch_size", "4", "--beta", "40", "--alpha", "40"])
    assert (a.model, a.y_size, a.tau, a.alpha) == ("lggmvae", 30, 0.4, 40)
    sch = ExponentialDecay(1e-4, decay_steps=1000000, decay_rate=0.4, staircase=True)
    assert sch(0) == 1e-4 and sch(999999) == 1e-4 and abs(sch(1000000) - 4e-5) < 1e-12


def test_tfrecord_codec_and_shuffle_buffer(tmp_path):
    """the CelebA files of vae/data.py:93-131 without TensorFlow."""
    from split_vae_amd import tfrecord as tfr
    assert tfr.crc32c(b"123456789") == 0xE3069283                       # CRC-32C check value
    assert tfr.crc32c(b"") == 0
    blob = bytes(range(256)) * 5 + b"tail"                              # native (sv_crc32c, slicing-by-8) == table loop,
    for lo, hi in [(0, len(blob)), (1, 777), (3, 12), (5, 5), (7, 8)]:  # at every alignment / length class
        assert tfr.crc32c(blob[lo:hi]) == tfr.crc32c_py(blob[lo:hi])
    # hand-assembled TensorProto: dtype DT_FLOAT, shape [2,3], tensor_content of 6 floats
    vals = np.arange(6, dtype="<f4") * 0.5 - 1
    proto = b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03" + b"\x22\x18" + vals.tobytes()
    got = tfr.parse_tensor(proto)
    assert got.shape == (2, 3) and np.array_equal(got.reshape(-1), vals)
    assert tfr.serialize_tensor(vals.reshape(2, 3)) == proto
    # file round trip with both CRCs verified; reading through get_dataset's streaming path
    rng = np.random.default_rng(0)
    imgs = (rng.integers(0, 256, (11, 64, 64, 3)) / 255.0 * 2 - 1).astype(np.float32)
    path = tmp_path / "train_64x64.tfrec"
    tfr.write_celeba_tfrec(str(path), imgs)
    back = np.stack(list(tfr.read_celeba_tfrec(str(path), 64, verify_data_crc=True)))
    assert np.array_equal(back, imgs)
    assert np.array_equal(tfr.read_celeba_tfrec_array(str(path), 64), imgs)   # vectorised whole-file reader
    raw = bytearray(path.read_bytes()); raw[40] ^= 1                    # flip a payload bit: the data CRC must catch it
    bad = tmp_path / "bad.tfrec"; bad.write_bytes(bytes(raw))
    with pytest.raises(IOError):
        list(tfr.read_celeba_tfrec(str(bad), 64, verify_data_crc=True))
    # shuffle buffer: a permutation of the input; buffer 1 = identity order; elements leave no earlier than their turn
    out = list(tfr.shuffle_buffer(range(100), 10, seed=3))
    assert sorted(out) == list(range(100)) and out != list(range(100))
    assert list(tfr.shuffle_buffer(range(20), 1, seed=0)) == list(range(20))
    assert all(v <= i + 10 for i, v in enumerate(out))                  # item v cannot be emitted before position v - buffer


def test_svhn_mat_reader_and_array_dataset(tmp_path):
    """vae/data.py:44-53: loadmat(...)['X'] is [32,32,3,N] uint8 -> [N,32,32,3] in [-1,1]; labels 10 -> 0."""
    import scipy.io
    from split_vae_amd import data
    rng = np.random.default_rng(0)
    X = rng.integers(0, 256, (32, 32, 3, 7), dtype=np.uint8)
    y = np.array([[1], [10], [3], [10], [9], [2], [5]], dtype=np.uint8)
    for name in ("train_32x32.mat", "test_32x32.mat"):
        scipy.io.savemat(str(tmp_path / name), {"X": X, "y": y})
    x, lab = data.load_svhn_mat(str(tmp_path / "train_32x32.mat"))
    assert x.shape == (7, 32, 32, 3) and x.dtype == np.float32
    assert np.array_equal(x[2], (X[..., 2].astype(np.float32) / 255.0 * 2 - 1))
    assert lab.tolist() == [1, 0, 3, 0, 9, 2, 5]
    assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    tr, te, shape = data.get_dataset("svhn", batch_size=3, synthetic=False, data_dir=str(tmp_path), device="cpu")
    assert shape == [-1, 32, 32, 3]
    batches = list(te)                      # finite test iterator: 7 images -> two full batches of 3
    assert len(batches) == 2 and tuple(batches[0].shape) == (3, 32, 32, 3)
    it = iter(tr)                           # training iterator repeats
    assert all(tuple(next(it).shape) == (3, 32, 32, 3) for _ in range(5))


def test_png_writer_round_trip(tmp_path):
    from split_vae_amd import visualizer as viz
    rng = np.random.default_rng(0)
    canvas = rng.uniform(-0.2, 1.2, (13, 21, 3))                        # out-of-range values are clipped
    p = viz.save_png(str(tmp_path / "sub" / "c.png"), canvas)
    back = viz.load_png(p)
    assert back.shape == (13, 21, 3) and back.dtype == np.uint8
    assert np.array_equal(back, np.clip(np.rint(canvas * 255), 0, 255).astype(np.uint8))
    grid = viz._tile(np.arange(4 * 2 * 3 * 3, dtype=np.float32).reshape(4, 2, 3, 3), 2, 2)
    assert grid.shape == (4, 6, 3) and np.array_equal(grid[:2, 3:], np.arange(18, 36, dtype=np.float32).reshape(2, 3, 3))
