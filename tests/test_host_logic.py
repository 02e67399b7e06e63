"""CPU: host-side mirror of the reference's Python surface (flags, dotdict, optimizer schedule,
augmentor selection, data-parallel bookkeeping) -- everything that needs no device."""
import os

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
    # the reference is fp32 end to end (vae/model.py:12, no mixed-precision policy): a drop-in command line must run at that
    # precision; bf16 operands are opt-in
    assert a.dtype == "f32" and b.dtype == "f32"
    assert build_parser().parse_args(["--dtype", "bf16"]).dtype == "bf16"


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
    # "encoders" = heads + convs as one bucket (the two-bucket schedule of small shards): one contiguous head of the buffer
    assert b["encoders"] == [(0, b["decoders"][0][0])]
    for names in (("decoders", "enc_heads", "enc_convs"), ("decoders", "encoders")):     # either schedule covers every parameter exactly once
        cover = np.zeros(n, int)
        for k in names:
            for lo, hi in b[k]:
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


def _write_svhn(root, n_train=7, n_extra=5, n_test=7, seed=0):
    """Tiny data/SVHN/{train,extra,test}_32x32.mat files in the layout scipy.io.loadmat gives the reference."""
    import scipy.io
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "SVHN"), exist_ok=True)
    out = {}
    for name, n in (("train", n_train), ("extra", n_extra), ("test", n_test)):
        X = rng.integers(0, 256, (32, 32, 3, n), dtype=np.uint8)
        X[:16, :16, 0, 0] = np.arange(256, dtype=np.uint8).reshape(16, 16)     # image 0 holds every pixel level
        y = rng.integers(1, 11, (n, 1)).astype(np.uint8)                        # SVHN labels are 1..10 (10 = digit 0)
        scipy.io.savemat(os.path.join(root, "SVHN", name + "_32x32.mat"), {"X": X, "y": y})
        out[name] = (X, y)
    return out


def test_svhn_normalisation_is_float64_then_cast():
    """vae/data.py:52: (x / 255.0 * 2 - 1).astype(np.float32) computes in float64; every one of the 256 levels must equal
    that, bit for bit -- the same arithmetic in fp32 is 1 ulp off on 128 of them."""
    from split_vae_amd import data
    levels = np.arange(256, dtype=np.uint8)
    want = np.array([np.float32(float(k) / 255.0 * 2 - 1) for k in range(256)], dtype=np.float32)   # Python floats are float64
    got = data.normalise_u8(levels)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    fp32_way = levels.astype(np.float32) / np.float32(255.0) * np.float32(2) - np.float32(1)
    assert int((fp32_way.view(np.uint32) != want.view(np.uint32)).sum()) == 128      # what the round-1 loader got wrong
    # the synthetic generator draws from the same domain, the same way
    syn = data.synthetic_images(2, 8, 8, seed=0, device="cpu").numpy()
    assert np.isin(syn.view(np.uint32), want.view(np.uint32)).all()


def test_svhn_reader_paths_labels_extra_and_remainder(tmp_path):
    """vae/data.py:23-75: data/SVHN/*.mat, `svhn` = train + extra, `svhn_no_extra` = train only, labels one_hot(y - 1)
    (digit 0 at the LAST index); vae/main.py:56-61: training batches never partial, the test set keeps its remainder."""
    from split_vae_amd import data
    files = _write_svhn(str(tmp_path))
    X, y = files["train"]
    x, lab = data.load_svhn_mat(str(tmp_path / "SVHN" / "train_32x32.mat"))
    assert x.shape == (7, 32, 32, 3) and x.dtype == np.float32
    assert np.array_equal(x, (np.transpose(X, (3, 0, 1, 2)) / 255.0 * 2 - 1).astype(np.float32))
    assert lab.tolist() == y.reshape(-1).tolist()                   # raw labels, 1..10
    oh = data.one_hot_svhn(np.array([[1], [10], [3]]))
    assert oh.shape == (3, 10) and oh[0].argmax() == 0 and oh[1].argmax() == 9 and oh[2].argmax() == 2 and oh.sum() == 3
    # no labels: bare image batches
    tr, te, shape = data.get_dataset("svhn", batch_size=3, data_dir=str(tmp_path), device="cpu")
    assert shape == [-1, 32, 32, 3] and not tr.labelled and not te.labelled
    assert tr.x.shape[0] == 12 and np.array_equal(tr.x[7:], data.normalise_u8(np.transpose(files["extra"][0], (3, 0, 1, 2))))
    tb = list(te)                                                   # 7 test images -> 3 + 3 + 1 (Dataset.batch keeps the remainder)
    assert [tuple(b.shape) for b in tb] == [(3, 32, 32, 3), (3, 32, 32, 3), (1, 32, 32, 3)]
    seen = np.concatenate([b.numpy() for b in tb])                  # a permutation of the test set
    want = data.normalise_u8(np.transpose(files["test"][0], (3, 0, 1, 2)))
    assert sorted(map(bytes, seen)) == sorted(map(bytes, want))
    it = iter(tr)                                                   # .repeat().batch(3) over 12 images: always full batches
    got = [next(it) for _ in range(9)]
    assert all(tuple(b.shape) == (3, 32, 32, 3) for b in got)
    first_epoch = np.concatenate([b.numpy() for b in got[:4]])
    assert sorted(map(bytes, first_epoch)) == sorted(map(bytes, tr.x))          # every image once per epoch
    tr2, _, _ = data.get_dataset("svhn_no_extra", batch_size=3, data_dir=str(tmp_path), device="cpu")
    assert tr2.x.shape[0] == 7
    # 7 images, batch 3, repeat: the third batch spans the epoch boundary
    b3 = [next(iter_) for iter_ in [iter(tr2)] for _ in range(3)]
    assert all(tuple(b.shape) == (3, 32, 32, 3) for b in b3)
    # labels: (images, one-hot) tuples in both sets, rows aligned
    trl, tel, _ = data.get_dataset("svhn", batch_size=4, data_dir=str(tmp_path), device="cpu", get_label=True)
    assert trl.labelled and tel.labelled
    xb, yb = next(iter(trl))
    assert tuple(xb.shape) == (4, 32, 32, 3) and tuple(yb.shape) == (4, 10)
    allx = np.concatenate([trl.x]); ally = trl.y
    for r in range(4):
        k = [i for i in range(allx.shape[0]) if np.array_equal(allx[i], xb[r].numpy())]
        assert any(np.array_equal(ally[i], yb[r].numpy()) for i in k)
    os.remove(str(tmp_path / "SVHN" / "extra_32x32.mat"))
    with pytest.raises(FileNotFoundError):
        data.get_dataset("svhn", batch_size=3, data_dir=str(tmp_path), device="cpu")
    data.get_dataset("svhn_no_extra", batch_size=3, data_dir=str(tmp_path), device="cpu")
    with pytest.raises(NotImplementedError):
        data.get_dataset("mnist", batch_size=3, data_dir=str(tmp_path), device="cpu")          # vae/data.py:21


def test_celeba_stream_keeps_remainder_and_spans_epochs(tmp_path):
    from split_vae_amd import data, tfrecord as tfr
    rng = np.random.default_rng(1)
    imgs = (rng.integers(0, 256, (7, 64, 64, 3)) / 255.0 * 2 - 1).astype(np.float32)
    os.makedirs(str(tmp_path / "celeba"))
    for name in ("train", "test"):
        tfr.write_celeba_tfrec(str(tmp_path / "celeba" / (name + "_64x64.tfrec")), imgs)
    tr, te, shape = data.get_dataset("celeba64", batch_size=3, data_dir=str(tmp_path), device="cpu", get_label=True)
    assert shape == [-1, 64, 64, 3] and not tr.labelled           # get_celeba_tfrec ignores get_label (vae/data.py:18-19)
    assert [b.shape[0] for b in te] == [3, 3, 1]
    it = iter(tr)
    assert all(next(it).shape[0] == 3 for _ in range(6))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["-no_label"]], ids=["labels", "no_label"])
def test_main_runs_on_svhn_files(tmp_path, monkeypatch, capsys, flags, lib_built):
    """The reference's first README command (main.py --beta 40 --patch_size 1, SVHN with labels) and its -no_label form,
    driven through main() on tiny .mat files: labelled batches are (images, labels) tuples and the step sees [B,32,32,6]."""
    from split_vae_amd import main as svmain
    _write_svhn(str(tmp_path / "data"), n_train=30, n_extra=9, n_test=27)
    monkeypatch.chdir(tmp_path)
    path = svmain.main(["--beta", "40", "--patch_size", "1", "--batch_size", "12", "--training_steps", "3", "--log_every", "2",
                        "--dtype", "f32"] + flags)
    out = capsys.readouterr().out
    assert "Training step 0" in out and "Training step 2" in out and "Training done!" in out
    assert os.path.exists(path)
    assert ("classifier-based test metrics are not available" in out) == (not flags)


@pytest.mark.gpu
def test_main_runs_on_celeba_tfrec(tmp_path, monkeypatch, capsys, lib_built):
    """README CelebA command (--dataset celeba64 --beta 120 --patch_size 8 -no_label) on a tiny TFRecord pair, and the
    same without -no_label: the CelebA files serve no labels, the run continues unlabelled."""
    from split_vae_amd import main as svmain, tfrecord as tfr
    rng = np.random.default_rng(2)
    imgs = (rng.integers(0, 256, (25, 64, 64, 3)) / 255.0 * 2 - 1).astype(np.float32)
    os.makedirs(str(tmp_path / "data" / "celeba"))
    for name in ("train", "test"):
        tfr.write_celeba_tfrec(str(tmp_path / "data" / "celeba" / (name + "_64x64.tfrec")), imgs)
    monkeypatch.chdir(tmp_path)
    for flags in (["-no_label"], []):
        svmain.main(["--dataset", "celeba64", "--beta", "120", "--patch_size", "8", "--batch_size", "20", "--training_steps", "2",
                     "--log_every", "2"] + flags)
        out = capsys.readouterr().out
        assert "Training done!" in out
        assert ("serves no labels" in out) == (not flags)
    assert any(f.startswith("style_transfer_celeba") for d, _, fs in os.walk(str(tmp_path / "output")) for f in fs)


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


def test_spair_cli_flags_match_reference():
    """spair/main.py:19-50: same names, types and defaults; README.md:93's SPLIT-SPAIR command parses."""
    from split_vae_amd.spair_main import build_parser, default_config
    a = build_parser().parse_args([])
    want = dict(learning_rate=1e-4, beta=0.5, dataset='cub_solid_fixed', channel=3, training_steps=100000, batch_size=32, runs=1, tau=0.8,
                object_size=32, latent_size=128, no_label=False, anneal_until=1.0, z_pres_anneal_step=10000.0, prior_z_zoom=0.0,
                prior_z_zoom_start=10.0, reconstruction_weight=1.0, bg_latent_size=4, local_latent_size=64, z_bg_beta=10.0, z_l_beta=0.1,
                z_what_beta=0.1, allow_growth=False, model='spair', patch_size=4, augmentation='scramble', split_z_l=False, dense_bg=False,
                dense_local=False, concat_bg=False, concat_z_what=False, concat_backbone=False)
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = build_parser().parse_args("--dataset cub_solid_fixed --z_bg_beta 10 --patch_size 8 --latent_size 64 --bg_latent_size 4 "
                                  "--local_latent_size 4 --model lg_spair -split_z_l -concat_z_what -dense_local -dense_bg "
                                  "--training_steps 200000".split())
    assert (b.model, b.split_z_l, b.concat_z_what, b.dense_local, b.dense_bg, b.local_latent_size) == ("lg_spair", True, True, True, True, 4)
    assert a.clipnorm_semantics == "tf2.0"            # not a reference flag: which TF version's Adam(clipnorm=...) semantics
    c = default_config(model="lg_spair")
    assert c.concat_z_bg is None and c.bg_model is None and c.image_size == [48, 48, 3]     # keys no flag defines read as None
