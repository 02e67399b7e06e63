"""Input side of the SPLIT-VAE path.  The metric runs on synthetic batches in the reference's data
domain (vae/data.py:52: x/255*2-1, fp32 NHWC); the on-disk formats of vae/data.py are
read without TensorFlow: SVHN .mat here, the CelebA TFRecord-of-serialize_tensor files in tfrecord.py,
both behind the 20 000-element shuffle buffer of vae/main.py:57-61."""
import os

import numpy as np
import torch

SHAPES = {"svhn": [-1, 32, 32, 3], "svhn_no_extra": [-1, 32, 32, 3], "celeba64": [-1, 64, 64, 3],
          "celeba128": [-1, 128, 128, 3]}


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

    def __iter__(self):
        i = 0
        while self.n_batches is None or i < self.n_batches:
            yield self.pool[i % len(self.pool)]
            i += 1


def load_svhn_mat(path):
    """vae/data.py:44-53: scipy.io.loadmat(...)['X'] is [32,32,3,N] uint8 -> [N,32,32,3] in [-1,1]."""
    import scipy.io
    m = scipy.io.loadmat(path)
    x = np.transpose(m["X"], (3, 0, 1, 2)).astype(np.float32) / 255.0 * 2 - 1
    y = m["y"].reshape(-1) % 10
    return x, y


class ArrayDataset:
    def __init__(self, x, batch_size, repeat, shuffle_seed=0, device="cuda"):
        self.x, self.bs, self.repeat, self.rng, self.device = x, batch_size, repeat, np.random.default_rng(shuffle_seed), device

    def __iter__(self):
        n = self.x.shape[0]
        while True:
            idx = self.rng.permutation(n)
            for s in range(0, n - self.bs + 1, self.bs):
                yield torch.from_numpy(self.x[np.sort(idx[s:s + self.bs])]).to(self.device)
            if not self.repeat:
                return


class StreamDataset:
    """shuffle(buffer).repeat().batch(B) over a re-iterable source of single images (vae/main.py:57-61)."""

    def __init__(self, make_iter, batch_size, repeat, buffer_size=20000, seed=0, device="cuda"):
        self.make_iter, self.bs, self.repeat, self.buffer_size, self.seed, self.device = make_iter, batch_size, repeat, buffer_size, seed, device

    def __iter__(self):
        from .tfrecord import shuffle_buffer
        epoch = 0
        while True:
            batch = []
            for x in shuffle_buffer(self.make_iter(), self.buffer_size, self.seed + epoch):
                batch.append(x)
                if len(batch) == self.bs:
                    yield torch.from_numpy(np.stack(batch)).to(self.device)
                    batch = []
            if batch and not self.repeat:
                yield torch.from_numpy(np.stack(batch)).to(self.device)      # Dataset.batch keeps the remainder
            if not self.repeat:
                return
            epoch += 1


def get_dataset(dataset="svhn", batch_size=64, synthetic=False, data_dir="data", device="cuda", test_batches=4):
    """vae/data.py:11-21 analogue -> (train_iterable, test_iterable, input_shape)."""
    if dataset not in SHAPES:
        raise NotImplementedError(dataset)          # vae/data.py:21
    shape = SHAPES[dataset]
    H, W = shape[1], shape[2]
    tr = os.path.join(data_dir, "train_32x32.mat")
    te = os.path.join(data_dir, "test_32x32.mat")
    if not synthetic and dataset.startswith("svhn") and os.path.exists(tr) and os.path.exists(te):
        xtr, _ = load_svhn_mat(tr)
        xte, _ = load_svhn_mat(te)
        return ArrayDataset(xtr, batch_size, True, 0, device), ArrayDataset(xte, batch_size, False, 1, device), shape
    if not synthetic and dataset.startswith("celeba"):               # vae/data.py:102-131
        from .tfrecord import read_celeba_tfrec
        tr = os.path.join(data_dir, "celeba", "train_%dx%d.tfrec" % (H, W))
        te = os.path.join(data_dir, "celeba", "test_%dx%d.tfrec" % (H, W))
        if os.path.exists(tr) and os.path.exists(te):
            return (StreamDataset(lambda: read_celeba_tfrec(tr, H), batch_size, True, 20000, 0, device),
                    StreamDataset(lambda: read_celeba_tfrec(te, H), batch_size, False, 20000, 1, device), shape)
    if not synthetic:
        raise FileNotFoundError("dataset files for %r not found under %r (no network here); pass --synthetic" % (dataset, data_dir))
    return (SyntheticDataset(H, W, batch_size, None, 0, device), SyntheticDataset(H, W, batch_size, test_batches, 77, device), shape)
