"""Tensor-level wrappers over the C ABI (include/splitvae.h).

torch is plumbing only: it owns device memory and the HIP stream; every computation below is a
call into libsplitvae_hip.so with raw pointers.  All tensors must live on a HIP device.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import SV_BF16, SV_F32, ConvDesc, LGVaeDesc, StepArgs, check

_TORCH_DT = {SV_BF16: torch.bfloat16, SV_F32: torch.float32}
_SV_DT = {torch.bfloat16: SV_BF16, torch.float32: SV_F32}


def sv_dtype(dt):
    if isinstance(dt, str):
        dt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32,
              "float32": torch.float32}[dt]
    if isinstance(dt, int):
        return dt
    return _SV_DT[dt]


_STREAM_HOLD = []      # hold_stream(): the stream handle of the enclosing step (torch's lookup is ~9 us, ~75 of them per GM step)


def _stream():
    if _STREAM_HOLD:
        return _STREAM_HOLD[-1]
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class hold_stream:
    """`with ops.hold_stream():` -- every op inside launches on the stream that was current at entry."""

    def __enter__(self):
        _STREAM_HOLD.append(C.c_void_p(torch.cuda.current_stream().cuda_stream))

    def __exit__(self, *exc):
        _STREAM_HOLD.pop()
        return False


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "split_vae_amd ops need device tensors (HIP); got a CPU tensor"
    assert t.is_contiguous()
    return C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------ A1 scramble (augmentation.py:43-57)
def random_perm(B, n_patch, seed, step=0, sample_offset=0, device="cuda"):
    perm = torch.empty((B, n_patch), dtype=torch.int32, device=device)
    check(_lib.load().sv_random_perm(_p(perm), B, n_patch, seed, step, sample_offset, _stream()), "sv_random_perm")
    return perm


def scramble_gather(x, perm, patch, staged=None):
    """x[B,H,W,3] fp32, perm[B,(H/patch)^2] int32 -> [B,H,W,6] = concat([x, x_aug], axis=-1)."""
    B, H, W, Cc = x.shape
    assert Cc == 3 and x.dtype == torch.float32 and perm.dtype == torch.int32
    assert perm.shape == (B, (H // patch) * (W // patch))
    out = torch.empty((B, H, W, 6), dtype=torch.float32, device=x.device)
    if staged is not None:                       # (x8, xh8): a plan's in8_x / in8_xh buffers, filled in the same pass
        x8, xh8 = staged
        assert x8.shape == (B, H, W, 8) and xh8.shape == (B, H, W, 8) and x8.dtype == xh8.dtype
        check(_lib.load().sv_scramble_gather_staged(_p(x), _p(perm), _p(out), _p(x8), _p(xh8), sv_dtype(x8.dtype), B, H, W, patch,
                                                    _stream()), "sv_scramble_gather_staged")
        return out
    check(_lib.load().sv_scramble_gather(_p(x), _p(perm), _p(out), B, H, W, patch, _stream()), "sv_scramble_gather")
    return out


# ------------------------------------------------------------------ A6 discretised logistic (vae/trainer.py:21-38)
def dlogistic_nll(images6, ch_off, out6, grad_dtype=None, grad_scale=1.0):
    B, H, W, _ = images6.shape
    lib = _lib.load()
    nll = torch.empty((B,), dtype=torch.float32, device=images6.device)
    ws = torch.empty((lib.sv_dlogistic_nll_workspace_bytes(B, H, W) // 4,), dtype=torch.float32, device=images6.device)
    grad = None
    gdt = 0
    if grad_dtype is not None:
        gdt = sv_dtype(grad_dtype)
        grad = torch.empty((B, H, W, 8), dtype=_TORCH_DT[gdt], device=images6.device)
    check(lib.sv_dlogistic_nll(_p(images6), ch_off, _p(out6), _p(nll), _p(grad), gdt, grad_scale, B, H, W, _p(ws),
                               _stream()), "sv_dlogistic_nll")
    return nll, grad


# ------------------------------------------------------------------ A4/A7 reparam + KL
def reparam_kl_fwd(pre, bias, eps=None, z_dtype=torch.bfloat16, seed=0, step=0, stream_id=0, sample_offset=0):
    B, L2 = pre.shape
    L = L2 // 2
    dev = pre.device
    z_mean = torch.empty((B, L), dtype=torch.float32, device=dev)
    z_sig = torch.empty_like(z_mean)
    z = torch.empty_like(z_mean)
    eps_out = torch.empty_like(z_mean)
    kl = torch.empty((B,), dtype=torch.float32, device=dev)
    z_lp = torch.empty((B, L), dtype=z_dtype, device=dev)
    check(_lib.load().sv_reparam_kl_fwd(_p(pre), _p(bias), _p(eps), _p(eps_out), _p(z_mean), _p(z_sig), _p(z),
                                        _p(z_lp), sv_dtype(z_dtype), L, 0, _p(kl), B, L, seed, step, stream_id,
                                        sample_offset, _stream()), "sv_reparam_kl_fwd")
    return z_mean, z_sig, z, z_lp, kl, eps_out


def reparam_kl_bwd(dz, z_mean, z_sig, eps, kl_scale, g_dtype=torch.bfloat16, dz2=None):
    B, L = z_mean.shape
    g = torch.empty((B, 2 * L), dtype=g_dtype, device=dz.device)
    check(_lib.load().sv_reparam_kl_bwd(_p(dz), dz.shape[1], _p(dz2), 0 if dz2 is None else dz2.shape[1],
                                        _p(z_mean), _p(z_sig), _p(eps), kl_scale, _p(g), sv_dtype(g_dtype), B, L,
                                        _stream()), "sv_reparam_kl_bwd")
    return g


# ------------------------------------------------------------------ K14 Keras Adam (vae/main.py:65)
def adam_step(p, g, m, v, t, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    check(_lib.load().sv_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, t, grad_scale,
                                   _stream()), "sv_adam_step")


# ------------------------------------------------------------------ K10a bilinear 2x (vae/model.py:163-167)
def upsample2x_fwd(x):
    B, H, W, Cc = x.shape
    out = torch.empty((B, 2 * H, 2 * W, Cc), dtype=x.dtype, device=x.device)
    check(_lib.load().sv_upsample2x_fwd(_p(x), _p(out), sv_dtype(x.dtype), B, H, W, Cc, _stream()), "sv_upsample2x_fwd")
    return out


def upsample2x_bwd(g_hi, y_lo_mask=None):
    B, H2, W2, Cc = g_hi.shape
    out = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=g_hi.dtype, device=g_hi.device)
    check(_lib.load().sv_upsample2x_bwd(_p(g_hi), _p(y_lo_mask), _p(out), sv_dtype(g_hi.dtype), B, H2 // 2, W2 // 2,
                                        Cc, _stream()), "sv_upsample2x_bwd")
    return out


# ------------------------------------------------------------------ SPLIT-SPAIR spatial transformer (spair/utils.py:119-330)
def stn_sample(img, z_where, Ho, Wo, inverse=False):
    """STN.call: img [B,H,W,C] (inverse: [B,B',H,W,C]) fp32, z_where [B,Hc,Wc,4] -> (out [B,B',Ho,Wo,C], obj_bbox_mask [B,B',4])."""
    B, Hc, Wc, _ = z_where.shape
    H, W, Cc = img.shape[-3:]
    assert img.dtype == torch.float32 and z_where.dtype == torch.float32
    assert tuple(img.shape[:-3]) == ((B, Hc * Wc) if inverse else (B,))
    out = torch.empty((B, Hc * Wc, Ho, Wo, Cc), dtype=torch.float32, device=img.device)
    bbox = torch.empty((B, Hc * Wc, 4), dtype=torch.float32, device=img.device)
    check(_lib.load().sv_stn_sample_fwd(_p(img.contiguous()), _p(z_where.contiguous()), _p(out), _p(bbox), B, Hc, Wc, H, W, Cc, Ho, Wo,
                                        1 if inverse else 0, _stream()), "sv_stn_sample_fwd")
    return out, bbox


def stn_sample_bwd(img, z_where, g_out, inverse=False, need_img=True):
    """-> (g_img like img | None, g_z_where like z_where) for the upstream gradient g_out [B,B',Ho,Wo,C]."""
    B, Hc, Wc, _ = z_where.shape
    H, W, Cc = img.shape[-3:]
    Ho, Wo = g_out.shape[2:4]
    lib = _lib.load()
    g_img = None
    if need_img:
        g_img = torch.empty_like(img) if lib.sv_stn_bwd_overwrites(H, W, Cc, 1 if inverse else 0) else torch.zeros_like(img)
    g_z = torch.empty_like(z_where)
    check(_lib.load().sv_stn_sample_bwd(_p(img.contiguous()), _p(z_where.contiguous()), _p(g_out.contiguous()), _p(g_img), _p(g_z), B, Hc,
                                        Wc, H, W, Cc, Ho, Wo, 1 if inverse else 0, _stream()), "sv_stn_sample_bwd")
    return g_img, g_z


def spair_render(obj, bg, z_depth, z_pres=None, z_pres_logits=None, training=False, noise=None):
    """Renderer.call (spair/spair.py:534-579): obj [B,B',H,W,C+1], bg [B,H,W,C], z_* [B,Hc,Wc,1] -> canvas [B,H,W,C]."""
    B, Bp, H, W, C1 = obj.shape
    out = torch.empty((B, H, W, C1 - 1), dtype=torch.float32, device=obj.device)
    f = lambda t: None if t is None else t.reshape(B, Bp).contiguous()
    check(_lib.load().sv_spair_render_fwd(_p(obj.contiguous()), _p(bg.contiguous()), _p(f(z_depth)), _p(f(z_pres)), _p(f(z_pres_logits)),
                                          _p(None if noise is None else noise.contiguous()), _p(out), B, Bp, H, W, C1 - 1,
                                          1 if training else 0, _stream()), "sv_spair_render_fwd")
    return out


def spair_render_bwd(obj, bg, z_depth, z_pres, g_out, noise=None):
    """-> (g_obj, g_bg, g_z_pres [B,B'], g_z_depth [B,B']) of the training-form renderer."""
    B, Bp, H, W, C1 = obj.shape
    g_obj, g_bg = torch.empty_like(obj), torch.empty_like(bg)
    g_zp = torch.empty((B, Bp), dtype=torch.float32, device=obj.device)
    g_zd = torch.empty_like(g_zp)
    f = lambda t: t.reshape(B, Bp).contiguous()
    lib = _lib.load()
    n = lib.sv_spair_render_bwd_workspace_floats(B, H, W)
    ws = torch.empty((n,), dtype=torch.float32, device=obj.device)
    check(lib.sv_spair_render_bwd_ws(_p(obj.contiguous()), _p(bg.contiguous()), _p(f(z_depth)), _p(f(z_pres)),
                                     _p(None if noise is None else noise.contiguous()), _p(g_out.contiguous()), _p(g_obj), _p(g_bg),
                                     _p(g_zp), _p(g_zd), B, Bp, H, W, C1 - 1, _p(ws), n, _stream()), "sv_spair_render_bwd_ws")
    return g_obj, g_bg, g_zp, g_zd


def _dev_scalar(v):
    """(python float, device pointer | None): a scalar given as a 1-element fp32 device tensor is read by the kernel at run
    time (hipGraph replays see its current value)."""
    if torch.is_tensor(v):
        assert v.dtype == torch.float32 and v.numel() == 1 and v.is_cuda
        return 0.0, v
    return float(v), None


def spair_zpres_kl(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob, temperature, grad_scale=None):
    """compute_z_pres_kl_yolo_air (spair/trainer.py:45-94): inputs [B,H,W,1] -> (kl [B] per-image sums, g_pre_sigmoid, g_logits)
    with the gradients of mean_b(kl_b) (grad_scale = 1/B) unless grad_scale is given."""
    B = z_pres.shape[0]
    n = z_pres[0].numel()
    f = lambda t: t.reshape(B, n).contiguous()
    kl = torch.empty((B,), dtype=torch.float32, device=z_pres.device)
    g_pre, g_log = torch.empty((B, n), dtype=torch.float32, device=z_pres.device), torch.empty((B, n), dtype=torch.float32, device=z_pres.device)
    pp, pp_dev = _dev_scalar(prior_prob)
    check(_lib.load().sv_spair_zpres_kl_dyn(_p(f(z_pres)), _p(f(z_pres_logits)), _p(f(z_pres_pre_sigmoid)), _p(kl), _p(g_pre), _p(g_log), B, n,
                                            pp, _p(pp_dev), float(temperature), 1.0 / B if grad_scale is None else float(grad_scale),
                                            _stream()), "sv_spair_zpres_kl")
    return kl, g_pre.reshape(z_pres.shape), g_log.reshape(z_pres.shape)


def adam_step_clipnorm_tensors(p, grads, m, v, tensor_off, clipnorm, t, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, alpha_dev=None):
    """adam_step_clipnorm over a LIST of gradient tensors (one per variable, contiguous fp32, in tensor_off order): their
    addresses go to the kernels by value -- no flat copy."""
    nt = tensor_off.numel() - 1
    assert len(grads) == nt and nt <= 128
    arr = (C.c_void_p * nt)(*[g.data_ptr() for g in grads])
    ws = torch.empty((256 * nt,), dtype=torch.float32, device=p.device)
    check(_lib.load().sv_adam_step_clipnorm_ptrs(_p(p), arr, _p(m), _p(v), _p(tensor_off), nt, _p(ws), float(clipnorm), float(lr),
                                                 float(beta1), float(beta2), float(eps), int(t), _p(alpha_dev), float(grad_scale),
                                                 _stream()), "sv_adam_step_clipnorm_ptrs")


SPAIR_LOSS_MODES = {"xent": 0, "kl": 1, "kl_prior": 2}


def spair_loss(mode, a, b, prior_mean=0.0, prior_sig=1.0, grads=True):
    """Per-image loss sums of spair/trainer.py (xent_loss / kl_divergence / kl_divergence_two_gauss vs a constant prior) over
    a, b [B, ...] fp32 -> (sums [B], ga | None, gb | None) with the per-element derivatives (xent: gb only)."""
    B = a.shape[0]
    n = a[0].numel()
    a2, b2 = a.reshape(B, n).contiguous(), b.reshape(B, n).contiguous()
    sums = torch.empty((B,), dtype=torch.float32, device=a.device)
    ga = torch.empty_like(a2) if grads and mode != "xent" else None
    gb = torch.empty_like(b2) if grads else None
    pm, pm_dev = _dev_scalar(prior_mean)
    check(_lib.load().sv_spair_loss_dyn(SPAIR_LOSS_MODES[mode], _p(a2), _p(b2), _p(sums), _p(ga), _p(gb), B, n, pm, _p(pm_dev),
                                        float(prior_sig), _stream()), "sv_spair_loss")
    return sums, (None if ga is None else ga.reshape(a.shape)), (None if gb is None else gb.reshape(b.shape))


def adam_alpha(lr, beta1, beta2, t):
    """Keras Adam's bias-corrected step size of iteration t (what sv_adam_step* computes from lr and t)."""
    return float(_lib.load().sv_adam_alpha(float(lr), float(beta1), float(beta2), int(t)))


def adam_step_clipnorm(p, g, m, v, tensor_off, clipnorm, t, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, alpha_dev=None):
    """Keras Adam(clipnorm=...) over flat buffers; tensor_off: int64 device tensor of n_tensors+1 offsets.  alpha_dev: a
    1-element device tensor holding adam_alpha(lr, beta1, beta2, t) -- read at run time instead of (lr, t) (hipGraph replay)."""
    nt = tensor_off.numel() - 1
    ws = torch.empty((256 * nt,), dtype=torch.float32, device=p.device)
    check(_lib.load().sv_adam_step_clipnorm_dyn(_p(p), _p(g), _p(m), _p(v), _p(tensor_off), nt, _p(ws), float(clipnorm), float(lr),
                                                float(beta1), float(beta2), float(eps), int(t), _p(alpha_dev), float(grad_scale),
                                                _stream()), "sv_adam_step_clipnorm")


# ------------------------------------------------------------------ K3-K10 conv (vae/model.py:36-38,:153-156)
class Conv2D:
    """One Conv2D(padding='same') layer instance on the MFMA path (forward, dgrad, wgrad)."""

    def __init__(self, B, H, W, Cin, Cout, k, stride, act=None, dtype=torch.bfloat16, y_f32=False, ups_in=False):
        r8 = lambda v: (v + 7) // 8 * 8
        self.dtype = dtype
        self.desc = ConvDesc(B, H, W, Cin, Cout, k, k, stride, 1 if act == "relu" else 0, sv_dtype(dtype),
                             r8(Cin), Cout if y_f32 else r8(Cout), 1 if y_f32 else 0, 1 if ups_in else 0)
        self.OH, self.OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        lib = _lib.load()
        nf = lib.sv_conv2d_wprep_elems(C.byref(self.desc), 0)
        nd = lib.sv_conv2d_wprep_elems(C.byref(self.desc), 1)
        if nf < 0 or nd < 0:
            raise _lib.SplitVaeError("unsupported conv geometry")
        self.nf, self.nd = nf, nd
        self.w_fwd = self.w_dgrad = None

    def prep(self, w_hwio, fwd=True, dgrad=True):
        """The kernels' weight images from the Keras HWIO kernel (either may be skipped: a forward-only or gradient-only call)."""
        dev = w_hwio.device
        if fwd:
            self.w_fwd = torch.empty((self.nf,), dtype=self.dtype, device=dev)
        if dgrad:
            self.w_dgrad = torch.empty((self.nd,), dtype=self.dtype, device=dev)
        check(_lib.load().sv_conv2d_prep_weights(C.byref(self.desc), _p(w_hwio), _p(self.w_fwd if fwd else None),
                                                 _p(self.w_dgrad if dgrad else None), _stream()), "sv_conv2d_prep_weights")

    def fwd(self, x, bias, out=None, workspace=True):
        d = self.desc
        y = out if out is not None else torch.empty((d.B, self.OH, self.OW, d.ldy),
                                                    dtype=torch.float32 if d.y_f32 else self.dtype, device=x.device)
        lib = _lib.load()
        if workspace:                                   # polyphase head: border terms through a workspace (else atomics)
            n = lib.sv_conv2d_fwd_workspace_bytes(C.byref(d))
            if n > 0 and (getattr(self, "_fws", None) is None or self._fws.numel() < n or self._fws.device != x.device):
                self._fws = torch.empty((n,), dtype=torch.uint8, device=x.device)
            ws = self._fws if n > 0 else None
            check(lib.sv_conv2d_nhwc_fwd_ws(C.byref(d), _p(x), _p(self.w_fwd), _p(bias), _p(y), _p(ws), n if n > 0 else 0,
                                            _stream()), "sv_conv2d_nhwc_fwd_ws")
        else:
            check(lib.sv_conv2d_nhwc_fwd(C.byref(d), _p(x), _p(self.w_fwd), _p(bias), _p(y), _stream()), "sv_conv2d_nhwc_fwd")
        return y

    def dgrad(self, dy, relu_mask=None, f32_atomic=False, out=None):
        """out (f32_atomic only): an fp32 [B,H,W,ldx] tensor the partial sums are ADDED to (several layers
        feeding one activation accumulate into the same buffer)."""
        d = self.desc
        if out is not None:
            assert f32_atomic and out.dtype == torch.float32
            dx = out
        elif f32_atomic:
            dx = torch.zeros((d.B, d.H, d.W, d.ldx), dtype=torch.float32, device=dy.device)
        else:
            dx = torch.zeros((d.B, d.H, d.W, d.ldx), dtype=self.dtype, device=dy.device)
        check(_lib.load().sv_conv2d_nhwc_dgrad(C.byref(d), _p(dy), _p(self.w_dgrad), _p(relu_mask), _p(dx),
                                               1 if f32_atomic else 0, _stream()), "sv_conv2d_nhwc_dgrad")
        return dx

    def dgrad_lowres(self, dy, relu_mask_lo=None):
        """ups_in layers: the gradient at the LOW-RES input [B,H/2,W/2,ldx] in one launch (conv-transpose, resize adjoint, ReLU
        mask); None when the geometry has no fused kernel (then: upsample2x_bwd(dgrad(dy), mask))."""
        d = self.desc
        dx = torch.empty((d.B, d.H // 2, d.W // 2, d.ldx), dtype=self.dtype, device=dy.device)
        lib = _lib.load()
        n = lib.sv_conv2d_dgrad_lowres_workspace_bytes(C.byref(d))          # > 0: the polyphase form (fp32 d4 / d5), edge terms through a workspace
        if n > 0:
            if getattr(self, "_dws", None) is None or self._dws.numel() < n or self._dws.device != dy.device:
                self._dws = torch.empty((n,), dtype=torch.uint8, device=dy.device)
            rc = lib.sv_conv2d_nhwc_dgrad_lowres_ws(C.byref(d), _p(dy), _p(self.w_dgrad), _p(relu_mask_lo), _p(dx), _p(self._dws), n, _stream())
        else:
            rc = lib.sv_conv2d_nhwc_dgrad_lowres(C.byref(d), _p(dy), _p(self.w_dgrad), _p(relu_mask_lo), _p(dx), _stream())
        if rc == _lib.STATUS_UNSUPPORTED:
            return None
        check(rc, "sv_conv2d_nhwc_dgrad_lowres")
        return dx

    def wgrad(self, x, dy, workspace=False, dw=None, db=None):
        """dw / db: zeroed fp32 views to accumulate into (e.g. slices of a flat gradient buffer)."""
        d = self.desc
        if dw is None and db is None:                    # one zero fill for both accumulators
            n = d.KH * d.KW * d.Cin * d.Cout
            n4 = (n + 3) // 4 * 4
            z = torch.zeros((n4 + d.Cout,), dtype=torch.float32, device=x.device)
            dw, db = z[:n].view(d.KH, d.KW, d.Cin, d.Cout), z[n4:]
        if dw is None:
            dw = torch.zeros((d.KH, d.KW, d.Cin, d.Cout), dtype=torch.float32, device=x.device)
        if db is None:
            db = torch.zeros((d.Cout,), dtype=torch.float32, device=x.device)
        if workspace:
            n = _lib.load().sv_conv2d_wgrad_workspace_bytes(C.byref(d))
            if getattr(self, "_ws", None) is None or self._ws.numel() < n:
                self._ws = torch.empty((n,), dtype=torch.uint8, device=x.device)
            check(_lib.load().sv_conv2d_nhwc_wgrad_ws(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(self._ws), n, _stream()),
                  "sv_conv2d_nhwc_wgrad_ws")
        else:
            check(_lib.load().sv_conv2d_nhwc_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _stream()),
                  "sv_conv2d_nhwc_wgrad")
        return dw, db

    def wgrad_poly(self, x_lo, dy, dw=None, db=None):
        """Polyphase weight gradient of the decoder head (sv_conv2d_nhwc_wgrad_poly); None when the layer has no such form."""
        d = self.desc
        lib = _lib.load()
        n = lib.sv_conv2d_wgrad_poly_workspace_bytes(C.byref(d))
        if n <= 0:
            return None
        if getattr(self, "_pws", None) is None or self._pws.numel() < n or self._pws.device != x_lo.device:
            self._pws = torch.zeros((n,), dtype=torch.uint8, device=x_lo.device)
        if dw is None:
            dw = torch.zeros((d.KH, d.KW, d.Cin, d.Cout), dtype=torch.float32, device=x_lo.device)
        if db is None:
            db = torch.zeros((d.Cout,), dtype=torch.float32, device=x_lo.device)
        check(lib.sv_conv2d_nhwc_wgrad_poly(C.byref(d), _p(x_lo), _p(dy), _p(dw), _p(db), _p(self._pws), n, _stream()),
              "sv_conv2d_nhwc_wgrad_poly")
        return dw, db


# ------------------------------------------------------------------ the whole-step plan
class LGVaePlan:
    """Native launch plan for LGVae.call / train_step_lg_vae (vae/model.py:189-200, vae/trainer.py:120-144)."""

    def __init__(self, B, H, W, global_latent=128, local_latent=128, beta=40.0, dtype=torch.bfloat16, device="cuda",
                 external_global_encoder=False):
        lib = _lib.load()
        self.lib = lib
        self.device = torch.device(device)
        self.dtype = dtype
        self.desc = LGVaeDesc(B, H, W, global_latent, local_latent, sv_dtype(dtype), float(beta),
                              1 if external_global_encoder else 0)
        h = C.c_void_p()
        check(lib.sv_lgvae_plan_create(C.byref(self.desc), C.byref(h)), "sv_lgvae_plan_create")
        self.handle = h
        self.n_params = lib.sv_lgvae_param_count(C.byref(self.desc))
        self.param_table = param_table(self.desc)
        nbytes = lib.sv_lgvae_workspace_bytes(h)
        self.workspace = torch.zeros((nbytes,), dtype=torch.uint8, device=self.device)
        check(lib.sv_lgvae_plan_bind(h, _p(self.workspace), nbytes, _stream()), "sv_lgvae_plan_bind")
        # generation of the shared input buffers in8_x / in8_xh: bumped by every writer (a staged augmentation, the split / pad
        # pass of any step that runs the encoders' forward); a staged batch is only valid while its generation is the current one
        self.in8_gen = 0
        self.graph_on = False
        self._views = {}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.sv_lgvae_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def buffer(self, name, dtype, shape):
        """A named buffer of the plan's workspace as a tensor view (memoised: the same view object for the same request -- a step asks for half a dozen of
        them, ~25 us of ctypes + view arithmetic each, and the launch-bound steps are host-bound by now)."""
        key = (name, dtype, tuple(shape))
        v = self._views.get(key)
        if v is None:
            off, nb = C.c_int64(), C.c_int64()
            check(self.lib.sv_lgvae_buffer(self.handle, name.encode(), C.byref(off), C.byref(nb)), "sv_lgvae_buffer " + name)
            v = self._views[key] = self.workspace[off.value: off.value + nb.value].view(dtype).view(*shape)
        return v

    def step(self, phases, params=None, grads=None, adam_m=None, adam_v=None, images6=None, eps_x=None,
             eps_x_hat=None, seed=0, step=0, sample_offset=0, lr=1e-4, beta1=0.9, beta2=0.999, adam_eps=1e-7, t=1,
             grad_scale=1.0, accumulate_metrics=False):
        a = StepArgs()
        for name, tns in (("params", params), ("grads", grads), ("adam_m", adam_m), ("adam_v", adam_v),
                          ("images6", images6), ("eps_x", eps_x), ("eps_x_hat", eps_x_hat)):
            setattr(a, name, None if tns is None else _p(tns).value)
        a.seed, a.step, a.sample_offset = seed, step, sample_offset
        a.lr, a.beta1, a.beta2, a.adam_eps, a.t = lr, beta1, beta2, adam_eps, t
        a.grad_scale, a.phases, a.accumulate_metrics = grad_scale, phases, 1 if accumulate_metrics else 0
        if (phases & _lib.PHASE_FWD_ENCODERS) and not (phases & _lib.PHASE_INPUTS_STAGED):
            self.in8_gen += 1                                  # this step's split / pad pass overwrites in8_x / in8_xh
        check(self.lib.sv_lgvae_step(self.handle, C.byref(a), _stream()), "sv_lgvae_step")

    def bucket_wait(self, bucket, stream):
        """Make `stream` (a torch.cuda.Stream) wait for gradient bucket 0 (decoders) / 1 (encoder heads) / 2 (encoder convs) / 3 (1 and 2) of the last
        step that carried PHASE_BUCKET_EVENTS (sv_lgvae_bucket_wait)."""
        check(self.lib.sv_lgvae_bucket_wait(self.handle, int(bucket), C.c_void_p(stream.cuda_stream)), "sv_lgvae_bucket_wait")

    def debug(self, key, value):
        """Test hooks of this plan (include/splitvae.h: sv_lgvae_plan_debug)."""
        check(self.lib.sv_lgvae_plan_debug(self.handle, key.encode(), int(value)), "sv_lgvae_plan_debug")

    def graph_enable(self, on=True):
        """hipGraph replay of `step` (include/splitvae.h: sv_lgvae_graph_enable); effective on a non-default stream."""
        check(self.lib.sv_lgvae_graph_enable(self.handle, 1 if on else 0), "sv_lgvae_graph_enable")
        self.graph_on = bool(on)

    def graph_count(self):
        return int(self.lib.sv_lgvae_graph_count(self.handle))

    def profile_enable(self, on=True):
        check(self.lib.sv_lgvae_profile_enable(self.handle, 1 if on else 0), "sv_lgvae_profile_enable")

    def profile_filter(self, name=None):
        check(self.lib.sv_lgvae_profile_filter(self.handle, (name or "").encode()), "sv_lgvae_profile_filter")

    def profile_read(self, max_entries=128):
        names = ((C.c_char * 64) * max_entries)()
        ms = (C.c_double * max_entries)()
        n_l = (C.c_int32 * max_entries)()
        fl = (C.c_double * max_entries)()
        by = (C.c_double * max_entries)()
        n = self.lib.sv_lgvae_profile_read(self.handle, max_entries, names, ms, n_l, fl, by)
        if n < 0:
            check(n, "sv_lgvae_profile_read")
        iss = (C.c_double * max_entries)()
        ni = self.lib.sv_lgvae_profile_read_issued(self.handle, max_entries, iss)
        if ni < 0:
            check(ni, "sv_lgvae_profile_read_issued")
        # flops: the direct form's count; issued: what the scope's algorithm really multiplies (polyphase forms: fewer)
        return [dict(name=names[i].value.decode(), total_ms=ms[i], launches=n_l[i], flops=fl[i], bytes=by[i], issued=iss[i] if i < ni else fl[i])
                for i in range(n)]


def param_table(desc):
    """[(name, offset, shape)] of the 40 variables in Keras creation order."""
    lib = _lib.load()
    out = []
    for i in range(40):
        off, nd = C.c_int64(), C.c_int32()
        shp = (C.c_int64 * 4)()
        name = C.create_string_buffer(96)
        check(lib.sv_lgvae_param_info(C.byref(desc), i, C.byref(off), C.byref(nd), C.byref(shp), name), "sv_lgvae_param_info")
        out.append((name.value.decode(), off.value, tuple(shp[k] for k in range(nd.value))))
    return out


# ------------------------------------------------------------------ A9 SPLIT-GMVAE glue (vae/model.py:48-79,:116-135)
ACT = {None: _lib.SV_ACT_NONE, "relu": _lib.SV_ACT_RELU, "elu": _lib.SV_ACT_ELU}


def act_fwd(a, C_, x, act=None, y_act=None, rate=0.0, keep_in=None, keep_out=None, seed=0, step=0, stream_id=0,
            sample_offset=0, rows_per_sample=1):
    """x[r, :C] = dropout(act(a[r, :C])), padding columns of x zeroed; a, x 2-D views [rows, ld]."""
    rows = a.shape[0]
    check(_lib.load().sv_act_fwd(_p(a), sv_dtype(a.dtype), a.shape[1], _p(y_act), _p(x), sv_dtype(x.dtype), x.shape[1], rows, C_,
                                 ACT[act], float(rate), _p(keep_in), _p(keep_out), seed, step, stream_id, sample_offset,
                                 rows_per_sample, _stream()), "sv_act_fwd")
    return x


def act_bwd(gx, C_, ga, y_act=None, act=None, rate=0.0, keep=None, gx2=None):
    rows = gx.shape[0]
    check(_lib.load().sv_act_bwd(_p(gx), sv_dtype(gx.dtype), gx.shape[1], _p(gx2), sv_dtype(gx2.dtype) if gx2 is not None else 0,
                                 gx2.shape[1] if gx2 is not None else 0, _p(y_act),
                                 sv_dtype(y_act.dtype) if y_act is not None else 0, y_act.shape[1] if y_act is not None else 0,
                                 ACT[act], float(rate), _p(keep), _p(ga), sv_dtype(ga.dtype), ga.shape[1], rows, C_, _stream()),
          "sv_act_bwd")
    return ga


def add(a, b, out):
    check(_lib.load().sv_add(_p(a), _p(b), _p(out), sv_dtype(out.dtype), out.numel(), _stream()), "sv_add")
    return out


def gm_metrics(nll_x, kl_x, nll_xh, kl_xh, y_kl, beta, alpha, out6):
    """vae/trainer.py:157-173: the five batch means + total (include/splitvae.h: sv_gm_metrics)."""
    check(_lib.load().sv_gm_metrics(_p(nll_x), _p(kl_x), _p(nll_xh), _p(kl_xh), _p(y_kl), nll_x.numel(), float(beta), float(alpha),
                                    _p(out6), _stream()), "sv_gm_metrics")
    return out6


def gumbel_softmax_fwd(logits, K, tau, y, y_lp, u=None, u_out=None, seed=0, step=0, sample_offset=0):
    B = logits.shape[0]
    check(_lib.load().sv_gumbel_softmax_fwd(_p(logits), logits.shape[1], _p(u), _p(u_out), float(tau), _p(y), _p(y_lp),
                                            sv_dtype(y_lp.dtype), y_lp.shape[1], B, K, seed, step, sample_offset, _stream()),
          "sv_gumbel_softmax_fwd")


def gumbel_softmax_bwd(gy, y, logits, K, tau, alpha_over_B, g_logits, y_kl):
    B = logits.shape[0]
    check(_lib.load().sv_gumbel_softmax_bwd(_p(gy), gy.shape[1] if gy is not None else 0, _p(y), _p(logits), logits.shape[1],
                                            float(tau), float(alpha_over_B), _p(g_logits),
                                            sv_dtype(g_logits.dtype) if g_logits is not None else 0,
                                            g_logits.shape[1] if g_logits is not None else 0, _p(y_kl), B, K, _stream()),
          "sv_gumbel_softmax_bwd")


def gm_head_fwd(a_m, a_s, a_pm, a_ps, zm, zs, z, pm, ps, zcat, z_col, kl2, eps=None, eps_out=None, seed=0, step=0,
                sample_offset=0):
    B, L = a_m.shape
    check(_lib.load().sv_gm_head_fwd(_p(a_m), _p(a_s), _p(a_pm), _p(a_ps), _p(eps), _p(eps_out), _p(zm), _p(zs), _p(z), _p(pm),
                                     _p(ps), _p(zcat), sv_dtype(zcat.dtype), zcat.shape[1], z_col, _p(kl2), B, L, seed, step,
                                     sample_offset, _stream()), "sv_gm_head_fwd")


def gm_head_bwd(dz, zm, zs, pm, ps, eps, kl_scale, g_am, g_as, g_apm, g_aps):
    B, L = zm.shape
    check(_lib.load().sv_gm_head_bwd(_p(dz), dz.shape[1], _p(zm), _p(zs), _p(pm), _p(ps), _p(eps), float(kl_scale), _p(g_am),
                                     _p(g_as), _p(g_apm), _p(g_aps), sv_dtype(g_am.dtype), B, L, _stream()), "sv_gm_head_bwd")


# ---------------------------------------------------------------- SPLIT-SPAIR Dense layers (dense_f32.hip)
def _pr(t):
    """Pointer of a 2-D tensor whose rows are contiguous (any row pitch)."""
    assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32
    return C.c_void_p(t.data_ptr())


def dense_f32_fwd(x, w, bias=None, act=None):
    """y = act(x . w + bias) for x [M, K] fp32 (row pitch = x.stride(0)), w [K, N] (Keras Dense kernel): exact-fp32 MFMA."""
    M, K = x.shape
    N = w.shape[1]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(_lib.load().sv_dense_f32_fwd(_pr(x), x.stride(0), _p(w), _p(bias), _p(y), N, M, K, N, 1 if act == "relu" else 0, _stream()),
          "sv_dense_f32_fwd")
    return y


def dense_f32_dgrad(dy, w, out=None):
    """dx = dy . w^T; out: an fp32 [M, K] tensor the product is ADDED to (atomics, K split over workgroups)."""
    M, N = dy.shape
    K = w.shape[0]
    dx = out if out is not None else torch.empty((M, K), dtype=torch.float32, device=dy.device)
    check(_lib.load().sv_dense_f32_dgrad(_pr(dy), dy.stride(0), _p(w), _pr(dx), dx.stride(0), M, K, N, 1 if out is not None else 0, _stream()),
          "sv_dense_f32_dgrad")
    return dx


def dense_f32_wgrad(x, dy):
    """(dw [K, N], dbias [N]) = (x^T . dy, column sums of dy)."""
    M, K = x.shape
    N = dy.shape[1]
    dw = torch.zeros((K, N), dtype=torch.float32, device=x.device)
    db = torch.zeros((N,), dtype=torch.float32, device=x.device)
    check(_lib.load().sv_dense_f32_wgrad(_pr(x), x.stride(0), _pr(dy), dy.stride(0), _p(dw), _p(db), M, K, N, _stream()), "sv_dense_f32_wgrad")
    return dw, db
