"""CPU check of the algebra behind the polyphase decoder head (csrc/conv_api.hip prep_poly, csrc/poly_fix.hip):

    Conv2D(6x6, 'same')(UpSampling2D(bilinear)(x))  ==  polyphase 5x5 conv over the edge-clamped low-res x  -  border terms

in float64 numpy, with the resize and the conv taken from the oracle restatement of the reference (vae/model.py:163-169).  The
composite weights, the ten border classes and the line construction are written here exactly as the device code builds them, so
a mistake in the derivation (coefficients, parity / tap order, excluded-tap sets, replicate vs zero line extension) fails on CPU."""
import numpy as np
import torch

from oracle import torch_ref


def _coef(p, k, t):
    """blend coefficient of hi-res tap k (0..5) of output parity p on the low-res offset t (-2..2): prep_poly's lambda."""
    h = p + k - 2
    m = h >> 1                                   # floor
    if h & 1:
        return 0.75 if t == m else 0.25 if t == m + 1 else 0.0
    return 0.25 if t == m - 1 else 0.75 if t == m else 0.0


def _line_up(v):
    """1-D 2x bilinear upsample with half-pixel centres and edge clamp (the fix kernel's line build): [n, C] -> [2n, C]."""
    n = v.shape[0]
    out = np.zeros((2 * n, v.shape[1]))
    for u in range(2 * n):
        m = u >> 1
        if u & 1:
            i0, i1, f = m, min(m + 1, n - 1), 0.25
        else:
            i0, i1, f = max(m - 1, 0), m, 0.75
        out[u] = v[i0] + (v[i1] - v[i0]) * f
    return out


def test_polyphase_identity_with_border_terms():
    rng = np.random.default_rng(0)
    B, h, C, Co = 2, 8, 5, 6
    x = rng.standard_normal((B, h, h, C))
    w = rng.standard_normal((6, 6, C, Co)) * 0.2
    bias = rng.standard_normal(Co)
    ref = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(torch.from_numpy(x)), torch.from_numpy(w), torch.from_numpy(bias), 1, None).numpy()
    H = 2 * h
    # ---- composite weights W'[py,px][ty,tx] and the polyphase conv over the edge-clamped input
    out = np.zeros((B, H, H, Co))
    xp = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="edge")
    for py in range(2):
        for px in range(2):
            Wc = np.zeros((5, 5, C, Co))
            for ty in range(-2, 3):
                for tx in range(-2, 3):
                    for ky in range(6):
                        for kx in range(6):
                            c = _coef(py, ky, ty) * _coef(px, kx, tx)
                            if c:
                                Wc[ty + 2, tx + 2] += c * w[ky, kx]
            for i in range(h):
                for j in range(h):
                    win = xp[:, i:i + 5, j:j + 5, :]                       # low-res rows i-2..i+2, cols j-2..j+2 (clamped)
                    out[:, 2 * i + py, 2 * j + px] = np.einsum("byxc,yxco->bo", win, Wc) + bias
    # ---- border terms: classes (edge coordinate, excluded taps); rows use the replicate-extended first / last upsampled row,
    # columns the zero-extended first / last upsampled column (poly_fix.hip)
    classes = [(0, (0, 1)), (1, (0,)), (H - 3, (5,)), (H - 2, (4, 5)), (H - 1, (3, 4, 5))]
    for b in range(B):
        top, bot = _line_up(x[b, 0]), _line_up(x[b, h - 1])                 # [H, C]
        left, right = _line_up(x[b, :, 0]), _line_up(x[b, :, h - 1])
        for edge, excl in classes:
            rline = top if edge < 2 else bot
            cline = left if edge < 2 else right
            for pos in range(H):
                for tap in range(6):
                    q = pos + tap - 2
                    rv = rline[min(max(q, 0), H - 1)]                       # replicate beyond the ends
                    cv = cline[q] if 0 <= q < H else np.zeros(C)            # zero beyond the ends
                    for k in excl:
                        out[b, edge, pos] -= rv @ w[k, tap]                 # row class: excluded ky = k, tap = kx
                        out[b, pos, edge] -= cv @ w[tap, k]                 # column class: excluded kx = k, tap = ky
    np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)


def test_composite_coefficients_are_a_partition_of_unity():
    """each hi-res tap draws weight 1 from its two low-res neighbours, and only offsets -2..2 are touched"""
    for p in range(2):
        for k in range(6):
            assert abs(sum(_coef(p, k, t) for t in range(-2, 3)) - 1.0) < 1e-15
            assert all(_coef(p, k, t) == 0.0 for t in (-4, -3, 3, 4))


def test_polyphase_weight_gradient_identity():
    """The same algebra for Conv2DBackpropFilter (DESIGN.md "Next kernel lever"): dW = P(dW') - dW_frame with dW' the weight
    gradient of the polyphase conv on the LOW-RES grid (edge-clamped input, dY read as its space-to-depth view), P the
    transpose of the composite-weight map, dW_frame the out-of-image taps of the five border rows / columns (1-D weight
    gradients along the fix lines).  Checked against autograd of the oracle's resize + conv."""
    rng = np.random.default_rng(1)
    B, h, C, Co = 2, 8, 4, 3
    H = 2 * h
    x = rng.standard_normal((B, h, h, C))
    w = rng.standard_normal((6, 6, C, Co)) * 0.2
    dy = rng.standard_normal((B, H, H, Co))
    wt = torch.from_numpy(w).requires_grad_(True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(torch.from_numpy(x)), wt, torch.zeros(Co, dtype=torch.float64), 1, None)
    (y * torch.from_numpy(dy)).sum().backward()
    want = wt.grad.numpy()
    # dW'[py,px][ty,tx][ci,co] = sum_{b,i,j} x~[b, i+ty, j+tx, ci] * dy[b, 2i+py, 2j+px, co]
    xp = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="edge")
    dWp = np.zeros((2, 2, 5, 5, C, Co))
    for py in range(2):
        for px in range(2):
            dys = dy[:, py::2, px::2, :]                                     # [B, h, h, Co]
            for ty in range(5):
                for tx in range(5):
                    dWp[py, px, ty, tx] = np.einsum("bijc,bijo->co", xp[:, ty:ty + h, tx:tx + h, :], dys)
    # projection P: dW[ky,kx] = sum Cy[py][ky][ty] Cx[px][kx][tx] dW'[py,px][ty,tx]
    dW = np.zeros_like(w)
    for ky in range(6):
        for kx in range(6):
            for py in range(2):
                for px in range(2):
                    for ty in range(-2, 3):
                        for tx in range(-2, 3):
                            c = _coef(py, ky, ty) * _coef(px, kx, tx)
                            if c:
                                dW[ky, kx] += c * dWp[py, px, ty + 2, tx + 2]
    # frame: the taps of the border rows / columns that leave the image
    classes = [(0, (0, 1)), (1, (0,)), (H - 3, (5,)), (H - 2, (4, 5)), (H - 1, (3, 4, 5))]
    for b in range(B):
        top, bot = _line_up(x[b, 0]), _line_up(x[b, h - 1])
        left, right = _line_up(x[b, :, 0]), _line_up(x[b, :, h - 1])
        for edge, excl in classes:
            rline = top if edge < 2 else bot
            cline = left if edge < 2 else right
            for pos in range(H):
                for tap in range(6):
                    q = pos + tap - 2
                    rv = rline[min(max(q, 0), H - 1)]
                    cv = cline[q] if 0 <= q < H else np.zeros(C)
                    for k in excl:
                        dW[k, tap] -= np.outer(rv, dy[b, edge, pos])         # row class: ky = k, kx = tap
                        dW[tap, k] -= np.outer(cv, dy[b, pos, edge])         # column class: ky = tap, kx = k
    np.testing.assert_allclose(dW, want, rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# PER-CLASS polyphase (csrc/conv_geom.h: svg_polyc; d4 = UpSampling2D -> Conv2D(32, 6), d3 = -> Conv2D(64, 4): vae/model.py:154-155,:163-165): every output
# parity class is its own conv over the edge-clamped low-res tensor with only the offsets that parity touches; any kernel size K, SAME pad before = (K-1)//2.
# The helpers below are the device code's closed forms (svg_pcoef, svg_polyc_taps, svg_polyc_excl).

def _pcoef(p, k, t, pad):
    h = p + k - pad
    m = h >> 1
    if h & 1:
        return 0.75 if t == m else 0.25 if t == m + 1 else 0.0
    return 0.25 if t == m - 1 else 0.75 if t == m else 0.0


def _polyc_taps(K, p):
    pad = (K - 1) // 2
    h0, h1 = p - pad, p + K - 1 - pad
    lo = (h0 >> 1) if (h0 & 1) else (h0 >> 1) - 1
    hi = (h1 >> 1) + 1 if (h1 & 1) else (h1 >> 1)
    return list(range(lo, hi + 1))


def _polyc_excl(K, c, k):
    pad = (K - 1) // 2
    if c < pad:
        return k < pad - c
    below = K - 1 - pad - (c - pad)
    return k > below - 1 + pad


def _class_edge(K, c, H):
    pad = (K - 1) // 2
    return c if c < pad else H - (K - 1 - pad) + (c - pad)


def _composite(w, K):
    pad = (K - 1) // 2
    out = {}
    for py in range(2):
        for px in range(2):
            for ty in _polyc_taps(K, py):
                for tx in _polyc_taps(K, px):
                    acc = np.zeros(w.shape[2:])
                    for ky in range(K):
                        for kx in range(K):
                            acc = acc + _pcoef(py, ky, ty, pad) * _pcoef(px, kx, tx, pad) * w[ky, kx]
                    out[(py, px, ty, tx)] = acc
    return out


def test_per_class_tap_sets_cover_exactly_the_nonzero_coefficients():
    for K in (4, 6):
        pad = (K - 1) // 2
        for p in range(2):
            nz = [t for t in range(-6, 7) if any(_pcoef(p, k, t, pad) for k in range(K))]
            assert _polyc_taps(K, p) == list(range(min(nz), max(nz) + 1))
    assert [len(_polyc_taps(6, p)) for p in (0, 1)] == [5, 4] and [len(_polyc_taps(4, p)) for p in (0, 1)] == [3, 4]
    for K, H in ((6, 16), (4, 16)):
        pad = (K - 1) // 2
        for c in range(K - 1):
            Y = _class_edge(K, c, H)
            assert [k for k in range(K) if _polyc_excl(K, c, k)] == [k for k in range(K) if not 0 <= Y + k - pad < H]


import pytest


@pytest.mark.parametrize("K", [6, 4])
def test_per_class_polyphase_forward_identity(K):
    rng = np.random.default_rng(K)
    B, h, C, Co = 2, 8, 5, 3
    H, pad = 2 * h, (K - 1) // 2
    x = rng.standard_normal((B, h, h, C))
    w = rng.standard_normal((K, K, C, Co)) * 0.2
    bias = rng.standard_normal(Co)
    ref = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(torch.from_numpy(x)), torch.from_numpy(w), torch.from_numpy(bias), 1, None).numpy()
    Wc = _composite(w, K)
    P = 3
    xp = np.pad(x, ((0, 0), (P, P), (P, P), (0, 0)), mode="edge")
    out = np.zeros((B, H, H, Co)) + bias
    for (py, px, ty, tx), M in Wc.items():
        out[:, py::2, px::2] += np.einsum("bijc,co->bijo", xp[:, P + ty:P + ty + h, P + tx:P + tx + h], M)
    for b in range(B):
        top, bot = _line_up(x[b, 0]), _line_up(x[b, h - 1])
        left, right = _line_up(x[b, :, 0]), _line_up(x[b, :, h - 1])
        for c in range(K - 1):
            edge = _class_edge(K, c, H)
            rline, cline = (top, left) if c < pad else (bot, right)
            for pos in range(H):
                for tap in range(K):
                    q = pos + tap - pad
                    rv = rline[min(max(q, 0), H - 1)]
                    cv = cline[q] if 0 <= q < H else np.zeros(C)
                    for k in range(K):
                        if _polyc_excl(K, c, k):
                            out[b, edge, pos] -= rv @ w[k, tap]
                            out[b, pos, edge] -= cv @ w[tap, k]
    np.testing.assert_allclose(out, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("K", [6, 4])
def test_per_class_polyphase_weight_gradient_identity(K):
    """dW = sum_c P_c(dW'_c) - dW_frame with dW'_c the weight gradient of class c's conv on the low-res grid (edge-clamped input, dY read at the class's
    sub-pixel) -- csrc/polyc_wgrad.hip -- against autograd of the oracle's resize + conv."""
    rng = np.random.default_rng(10 + K)
    B, h, C, Co = 2, 8, 4, 3
    H, pad = 2 * h, (K - 1) // 2
    x = rng.standard_normal((B, h, h, C))
    w = rng.standard_normal((K, K, C, Co)) * 0.2
    dy = rng.standard_normal((B, H, H, Co))
    wt = torch.from_numpy(w).requires_grad_(True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(torch.from_numpy(x)), wt, torch.zeros(Co, dtype=torch.float64), 1, None)
    (y * torch.from_numpy(dy)).sum().backward()
    want = wt.grad.numpy()
    P = 3
    xp = np.pad(x, ((0, 0), (P, P), (P, P), (0, 0)), mode="edge")
    dW = np.zeros_like(w)
    for py in range(2):
        for px in range(2):
            dys = dy[:, py::2, px::2, :]
            for ty in _polyc_taps(K, py):
                for tx in _polyc_taps(K, px):
                    dWp = np.einsum("bijc,bijo->co", xp[:, P + ty:P + ty + h, P + tx:P + tx + h, :], dys)
                    for ky in range(K):
                        for kx in range(K):
                            dW[ky, kx] += _pcoef(py, ky, ty, pad) * _pcoef(px, kx, tx, pad) * dWp
    for b in range(B):
        top, bot = _line_up(x[b, 0]), _line_up(x[b, h - 1])
        left, right = _line_up(x[b, :, 0]), _line_up(x[b, :, h - 1])
        for c in range(K - 1):
            edge = _class_edge(K, c, H)
            rline, cline = (top, left) if c < pad else (bot, right)
            for pos in range(H):
                for tap in range(K):
                    q = pos + tap - pad
                    rv = rline[min(max(q, 0), H - 1)]
                    cv = cline[q] if 0 <= q < H else np.zeros(C)
                    for k in range(K):
                        if _polyc_excl(K, c, k):
                            dW[k, tap] -= np.outer(rv, dy[b, edge, pos])
                            dW[tap, k] -= np.outer(cv, dy[b, pos, edge])
    np.testing.assert_allclose(dW, want, rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# POLYPHASE INPUT GRADIENT (csrc/conv_geom.h: svg_polyd; csrc/polyd_dgrad.hip): Conv2DBackpropInput + ResizeBilinearGrad as ONE stride-2 conv with (2R+1)^2
# taps over the hi-res dy, plus corrections of the first / last low-res row and column (zero padding of the upsampled image, edge clamp of the resize) and of
# the four corner pixels.  Written with the device code's index maps (svg_polyd_hitap, svg_polyd_kpm).

def _hi_taps(K):
    """hi-res offset dh = p - 2 t  ->  (parity p, low-res offset t)"""
    return {p - 2 * t: (p, t) for p in range(2) for t in _polyc_taps(K, p)}


def _kpm(K, hi_edge, q):
    pad = (K - 1) // 2
    a, b = (pad + q, pad + 1 + q) if hi_edge else (pad - q, pad - 1 - q)
    return (a if 0 <= a < K else -1), (b if 0 <= b < K else -1)


@pytest.mark.parametrize("K", [6, 4])
def test_polyphase_input_gradient_identity(K):
    pad = (K - 1) // 2
    rng = np.random.default_rng(3 + K)
    B, h, C, Co = 2, 8, 4, 3
    H = 2 * h
    x = rng.standard_normal((B, h, h, C))
    w = rng.standard_normal((K, K, C, Co)) * 0.2
    dy = rng.standard_normal((B, H, H, Co))
    xt = torch.from_numpy(x).requires_grad_(True)
    y = torch_ref.conv2d_same(torch_ref.resize_bilinear_2x(xt), torch.from_numpy(w), torch.zeros(Co, dtype=torch.float64), 1, None)
    (y * torch.from_numpy(dy)).sum().backward()
    want = xt.grad.numpy()
    ht = _hi_taps(K)
    R = max(max(ht), -min(ht))
    assert sorted(ht) == list(range(-R, R + 1)) and R == (4 if K == 6 else 3)        # every hi-res offset belongs to exactly one (parity, offset)
    E = R + 1
    dyp = np.pad(dy, ((0, 0), (E, E), (E, E), (0, 0)))
    at = lambda r, c: dyp[:, r + E, c + E]                                                # dy[:, r, c], zero outside the image
    wz = lambda ky, kx: w[ky, kx] if ky >= 0 and kx >= 0 else np.zeros((C, Co))
    got = np.zeros((B, h, h, C))
    for dyh, (py, ty) in ht.items():                                                      # main term: stride-2 conv over the hi-res dy
        for dxh, (px, tx) in ht.items():
            V = sum(_pcoef(py, ky, ty, pad) * _pcoef(px, kx, tx, pad) * w[ky, kx] for ky in range(K) for kx in range(K))
            for i in range(h):
                for j in range(h):
                    got[:, i, j] += at(2 * i + dyh, 2 * j + dxh) @ V.T
    main_only = np.abs(got - want).max()
    for hi_edge in (0, 1):                                                                # edges: rows 0 / h-1 and columns 0 / h-1
        nq = K - pad if hi_edge else pad + 1
        edge_lo = h - 1 if hi_edge else 0
        for q in range(nq):
            kp, km = _kpm(K, hi_edge, q)
            across = H - 1 - q if hi_edge else q
            for d, (pp, tt) in ht.items():
                Vr = 0.25 * sum(_pcoef(pp, k, tt, pad) * (wz(kp, k) - wz(km, k)) for k in range(K))       # row edge: x-composite of the kernel-row difference
                Vc = 0.25 * sum(_pcoef(pp, k, tt, pad) * (wz(k, kp) - wz(k, km)) for k in range(K))       # column edge
                for pos in range(h):
                    got[:, edge_lo, pos] += at(across, 2 * pos + d) @ Vr.T
                    got[:, pos, edge_lo] += at(2 * pos + d, across) @ Vc.T
    for cr in (0, 1):                                                                     # corners
        for cc in (0, 1):
            for qr in range(K - pad if cr else pad + 1):
                for qs in range(K - pad if cc else pad + 1):
                    kpr, kmr = _kpm(K, cr, qr)
                    kps, kms = _kpm(K, cc, qs)
                    Wc = 0.0625 * (wz(kpr, kps) - wz(kpr, kms) - wz(kmr, kps) + wz(kmr, kms))
                    got[:, h - 1 if cr else 0, h - 1 if cc else 0] += dy[:, H - 1 - qr if cr else qr, H - 1 - qs if cc else qs] @ Wc.T
    assert main_only > 0.1                                                                # the corrections matter
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
