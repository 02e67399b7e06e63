"""Input side of the SPLIT-VAE path.  The metric runs on synthetic batches in the reference's data
domain (vae/data.py:52: x/255*2-1, fp32 NHWC); the on-disk formats of vae/data.py are
read without TensorFlow: SVHN .mat here, the CelebA TFRecord-of-serialize_tensor files in tfrecord.py,
both behind the pipeline of vae/main.py:56-61,

    train: dataset.shuffle(20000).repeat().map(augment).batch(B)     (batches never partial, they span epochs)
    test : dataset.shuffle(20000).map(augment).batch(B)              (the last batch keeps the remainder)

with (image, label) tuples when `get_label` (vae/main.py:56-58, vae/data.py:54-62) and bare images otherwise.
The augmentation itself (the patch scramble) runs batched on the device, after batching."""
import os

import numpy as np
import torch

SHAPES = {"svhn": [-1, 32, 32, 3], "svhn_no_extra": [-1, 32, 32, 3], "celeba64": [-1, 64, 64, 3],
          "celeba128": [-1, 128, 128, 3]}
SHUFFLE_BUFFER = 20000                      # vae/main.py:57-61


def synthetic_images(n, H, W, seed=0, device="cuda", sample_offset=0):
    """n images uniform over the 256 quantised levels {-1 + 2k/255}; sample i depends only on
    (seed, sample_offset + i) so shards of a global batch are slices of the single-process batch."""
    out = torch.empty((n, H, W, 3), dtype=torch.float32)
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64([seed, sample_offset + i]))
        out[i] = torch.from_numpy((rng.integers(0, 256, size=(H, W, 3)) / 255.0 * 2 - 1).astype(np.float32))
    return out.to(device)


class SyntheticDataset:
    """Infinite (train) or finite (test) iterator of [B,H,W,3] device batches."""

    def __init__(self, H, W, batch_size, n_batches=None, seed=0, device="cuda", pool=4):
        self.pool = [synthetic_images(batch_size, H, W, seed + 1000 * k, device) for k in range(pool)]
        self.n_batches = n_batches
        self.labelled = False

    def __iter__(self):
        i = 0
        while self.n_batches is None or i < self.n_batches:
            yield self.pool[i % len(self.pool)]
            i += 1


def normalise_u8(x):
    """vae/data.py:52-53: (x / 255.0 * 2 - 1).astype(np.float32) -- the division, scale and shift run in float64
    (NumPy promotes uint8 / float) and only the result is rounded to fp32.  Doing the arithmetic in fp32 instead
    differs in the last bit on 128 of the 256 pixel levels."""
    return (np.asarray(x) / 255.0 * 2 - 1).astype(np.float32)


def one_hot_svhn(y, depth=10):
    """vae/data.py:55-57: tf.squeeze(tf.one_hot(y - 1, 10)); SVHN stores digit 0 as label 10, so "0" is the LAST
    index.  (tf.one_hot gives an all-zero row for an index outside [0, depth).)"""
    idx = np.asarray(y).reshape(-1).astype(np.int64) - 1
    out = np.zeros((idx.shape[0], depth), np.float32)
    ok = (idx >= 0) & (idx < depth)
    out[np.nonzero(ok)[0], idx[ok]] = 1.0
    return out


def load_svhn_mat(path):
    """vae/data.py:44-53: scipy.io.loadmat(...)['X'] is [32,32,3,N] uint8 -> [N,32,32,3] fp32 in [-1,1];
    returns (x, y) with y the raw [N] labels in 1..10."""
    import scipy.io
    m = scipy.io.loadmat(path)
    x = normalise_u8(np.transpose(m["X"], (3, 0, 1, 2)))
    return x, m["y"].reshape(-1)


def _shuffled_indices(n, buffer_size, rng):
    """Index stream of tf.data's shuffle(buffer_size) over elements 0..n-1: keep `buffer_size` candidates, emit a
    uniformly random one and replace it by the next input, drain randomly at the end (same algorithm as
    tfrecord.shuffle_buffer, on indices so that an array source is gathered once per batch)."""
    buf = list(range(min(n, buffer_size)))
    for nxt in range(len(buf), n):
        j = int(rng.integers(len(buf)))
        out, buf[j] = buf[j], nxt
        yield out
    while buf:
        j = int(rng.integers(len(buf)))
        buf[j], buf[-1] = buf[-1], buf[j]
        yield buf.pop()


class ArrayDataset:
    """shuffle(buffer).repeat()?.batch(B) over in-memory arrays (vae/main.py:56-61).  With labels the batches are
    (x[B,H,W,3], y[B,10]) tuples; repeat=True never yields a partial batch (the batch spans the epoch boundary, as
    .repeat().batch() does), repeat=False ends with the remainder (Dataset.batch keeps it)."""

    def __init__(self, x, batch_size, repeat, shuffle_seed=0, device="cuda", y=None, buffer_size=SHUFFLE_BUFFER):
        self.x, self.y, self.bs, self.repeat, self.seed, self.device = x, y, batch_size, repeat, shuffle_seed, device
        self.buffer_size = buffer_size
        self.labelled = y is not None

    def _emit(self, idx):
        idx = np.asarray(idx)
        xb = torch.from_numpy(self.x[idx]).to(self.device)
        if self.y is None:
            return xb
        return xb, torch.from_numpy(self.y[idx]).to(self.device)

    def __iter__(self):
        n = self.x.shape[0]
        epoch, pend = 0, []
        while True:
            rng = np.random.default_rng([self.seed, epoch])
            for i in _shuffled_indices(n, self.buffer_size, rng):
                pend.append(i)
                if len(pend) == self.bs:
                    yield self._emit(pend)
                    pend = []
            if not self.repeat:
                if pend:
                    yield self._emit(pend)
                return
            if n == 0:
                return
            epoch += 1


class StreamDataset:
    """shuffle(buffer).repeat()?.batch(B) over a re-iterable source of single images (vae/main.py:59-61; the
    CelebA TFRecord files carry no labels, vae/data.py:102-134)."""

    def __init__(self, make_iter, batch_size, repeat, buffer_size=SHUFFLE_BUFFER, seed=0, device="cuda"):
        self.make_iter, self.bs, self.repeat, self.buffer_size, self.seed, self.device = make_iter, batch_size, repeat, buffer_size, seed, device
        self.labelled = False

    def __iter__(self):
        from .tfrecord import shuffle_buffer
        epoch, batch = 0, []
        while True:
            seen = 0
            for x in shuffle_buffer(self.make_iter(), self.buffer_size, self.seed + epoch):
                seen += 1
                batch.append(x)
                if len(batch) == self.bs:
                    yield torch.from_numpy(np.stack(batch)).to(self.device)
                    batch = []
            if not self.repeat:
                if batch:
                    yield torch.from_numpy(np.stack(batch)).to(self.device)      # Dataset.batch keeps the remainder
                return
            if seen == 0:
                return
            epoch += 1                      # .repeat().batch(): a batch may span the epoch boundary


def get_dataset(dataset="svhn", batch_size=64, synthetic=False, data_dir="data", device="cuda", test_batches=4,
                get_label=False):
    """vae/data.py:11-21 + the pipeline of vae/main.py:55-61 -> (train_iterable, test_iterable, input_shape).
    Each iterable has `.labelled`: True when its batches are (images, one-hot labels) tuples -- only the SVHN files
    carry labels (vae/data.py:54-62; get_celeba_tfrec ignores get_label)."""
    if dataset not in SHAPES:
        raise NotImplementedError('Dataset doesn\'t exit')          # vae/data.py:21
    shape = SHAPES[dataset]
    H, W = shape[1], shape[2]
    if synthetic:
        return (SyntheticDataset(H, W, batch_size, None, 0, device), SyntheticDataset(H, W, batch_size, test_batches, 77, device), shape)
    if dataset.startswith("svhn"):                                     # vae/data.py:23-75
        root = os.path.join(data_dir, "SVHN")
        tr, te, ex = (os.path.join(root, f + "_32x32.mat") for f in ("train", "test", "extra"))
        extra = dataset == "svhn"                                      # 'svhn_no_extra' leaves extra_32x32.mat out (:15-16)
        need = [tr, te] + ([ex] if extra else [])
        missing = [f for f in need if not os.path.exists(f)]
        if missing:
            raise FileNotFoundError("SVHN files %s not found (the reference would download them, vae/data.py:34-42; "
                                    "no network here); pass --synthetic%s" %
                                    (missing, " or --dataset svhn_no_extra" if missing == [ex] else ""))
        xtr, ytr = load_svhn_mat(tr)
        xte, yte = load_svhn_mat(te)
        if extra:
            xex, yex = load_svhn_mat(ex)
            xtr, ytr = np.concatenate([xtr, xex]), np.concatenate([ytr, yex])
        ltr, lte = (one_hot_svhn(ytr), one_hot_svhn(yte)) if get_label else (None, None)
        return (ArrayDataset(xtr, batch_size, True, 0, device, y=ltr), ArrayDataset(xte, batch_size, False, 1, device, y=lte), shape)
    from .tfrecord import read_celeba_tfrec                            # vae/data.py:102-131
    tr = os.path.join(data_dir, "celeba", "train_%dx%d.tfrec" % (H, W))
    te = os.path.join(data_dir, "celeba", "test_%dx%d.tfrec" % (H, W))
    if not (os.path.exists(tr) and os.path.exists(te)):
        raise FileNotFoundError("dataset files for %r not found under %r (no network here); pass --synthetic" % (dataset, data_dir))
    return (StreamDataset(lambda: read_celeba_tfrec(tr, H), batch_size, True, SHUFFLE_BUFFER, 0, device),
            StreamDataset(lambda: read_celeba_tfrec(te, H), batch_size, False, SHUFFLE_BUFFER, 1, device), shape)
