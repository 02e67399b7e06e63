"""PyTorch-ROCm custom ops over the C ABI ("a thin torch.library shim that pulls pointers and the current
HIP stream from tensors and turns non-zero codes into RuntimeError; torch.autograd.Function wrappers pair fwd/bwd").

    import split_vae_amd.torch_ops            # registers the `split_vae::*` operators
    y = split_vae_amd.torch_ops.conv2d(x, w_hwio, bias, stride=1, act="relu")          # differentiable, NHWC
    nll = split_vae_amd.torch_ops.dlogistic_nll(images6, 0, out6)                      # differentiable in out6
    torch.ops.split_vae.adam_step(p, g, m, v, t, lr, b1, b2, eps, scale)               # raw entry points

Every operator has a HIP ("CUDA" dispatch key on ROCm) implementation only: a CPU tensor raises NotImplementedError
from the dispatcher -- there is no CPU kernel and nothing here imports the oracle.  Each implementation is one or two
calls into libsplitvae_hip.so (ops.py); torch owns the memory and the stream.  The whole-step plan (sv_lgvae_step) does
not go through these ops: it sequences the same kernels natively; these exist so that the kernels compose with autograd
in a user's own graph.
"""
import torch

from . import _lib, ops

_lib_def = torch.library.Library("split_vae", "DEF")
_lib_impl = torch.library.Library("split_vae", "IMPL", "CUDA")        # HIP devices dispatch under the CUDA key on ROCm

_lib_def.define("scramble_gather(Tensor x, Tensor perm, int patch) -> Tensor")
_lib_def.define("conv2d_nhwc_fwd(Tensor x, Tensor w_hwio, Tensor? bias, int stride, int act, bool ups_in, bool y_f32) -> Tensor")
_lib_def.define("conv2d_nhwc_dgrad(Tensor dy, Tensor w_hwio, Tensor? relu_mask, int H, int W, int ldx, int stride, bool ups_in) -> Tensor")
_lib_def.define("conv2d_nhwc_wgrad(Tensor x, Tensor dy, int KH, int KW, int Cin, int Cout, int stride, bool ups_in) -> (Tensor, Tensor)")
_lib_def.define("dlogistic_nll(Tensor images6, int ch_off, Tensor out6, float grad_scale, bool bf16_grad) -> (Tensor, Tensor)")
_lib_def.define("reparam_kl_fwd(Tensor pre, Tensor bias, Tensor? eps, int seed, int step, int stream_id, int sample_offset) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib_def.define("reparam_kl_bwd(Tensor dz, Tensor z_mean, Tensor z_sig, Tensor eps, float kl_scale) -> Tensor")
_lib_def.define("upsample2x_fwd(Tensor x) -> Tensor")
_lib_def.define("upsample2x_bwd(Tensor g_hi, Tensor? relu_mask) -> Tensor")
_lib_def.define("adam_step(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, int t, float lr, float beta1, float beta2, float eps, float grad_scale) -> ()")
# SPLIT-SPAIR operators (fp32; spair/utils.py:119-330, spair/spair.py:534-579, spair/trainer.py:28-94)
_lib_def.define("stn_sample_fwd(Tensor img, Tensor z_where, int Ho, int Wo, bool inverse) -> (Tensor, Tensor)")
_lib_def.define("stn_sample_bwd(Tensor img, Tensor z_where, Tensor g_out, bool inverse, bool need_img) -> (Tensor, Tensor)")
_lib_def.define("spair_render_fwd(Tensor obj, Tensor bg, Tensor z_depth, Tensor? z_pres, Tensor? z_pres_logits, Tensor? noise, bool training) -> Tensor")
_lib_def.define("spair_render_bwd(Tensor obj, Tensor bg, Tensor z_depth, Tensor z_pres, Tensor? noise, Tensor g_out) -> (Tensor, Tensor, Tensor, Tensor)")
_lib_def.define("spair_zpres_kl(Tensor z_pres, Tensor z_pres_logits, Tensor z_pres_pre_sigmoid, float prior_prob, float temperature, Tensor? prior_prob_dev=None) -> (Tensor, Tensor, Tensor)")

_lib_def.define("spair_loss(str mode, Tensor a, Tensor b, float prior_mean, float prior_sig, Tensor? prior_mean_dev=None) -> (Tensor, Tensor, Tensor)")

ACT = {None: 0, "none": 0, "relu": 1}


def _conv(x, Cin, Cout, KH, stride, act, ups_in, y_f32, H=None, W=None):
    B = x.shape[0]
    H = x.shape[1] * (2 if ups_in else 1) if H is None else H
    W = x.shape[2] * (2 if ups_in else 1) if W is None else W
    return ops.Conv2D(B, H, W, Cin, Cout, KH, stride, act="relu" if act == 1 else None, dtype=x.dtype, y_f32=y_f32, ups_in=ups_in)


def _impl(name):
    def deco(fn):
        _lib_impl.impl(name, fn)
        return fn
    return deco


@_impl("scramble_gather")
def _scramble_gather(x, perm, patch):
    return ops.scramble_gather(x.contiguous(), perm.to(torch.int32).contiguous(), patch)


@_impl("conv2d_nhwc_fwd")
def _conv_fwd(x, w_hwio, bias, stride, act, ups_in, y_f32):
    """x [B,h,w,ldx] NHWC (ldx = Cin rounded up to 8; with ups_in the conv sees the 2x bilinear upsample of x),
    w_hwio [KH,KW,Cin,Cout] fp32 Keras layout -> y [B,OH,OW,ldy] (ldy = Cout rounded up to 8, or Cout fp32 when y_f32)."""
    KH, KW, Cin, Cout = w_hwio.shape
    c = _conv(x, Cin, Cout, KH, stride, act, ups_in, y_f32)
    c.prep(w_hwio.contiguous(), dgrad=False)
    if bias is None:
        bias = torch.zeros((Cout,), dtype=torch.float32, device=x.device)
    return c.fwd(x.contiguous(), bias.contiguous())


@_impl("conv2d_nhwc_dgrad")
def _conv_dgrad(dy, w_hwio, relu_mask, H, W, ldx, stride, ups_in):
    """dL/dx of the conv for x [B,H,W,ldx] (H, W: the conv's logical input size, i.e. hi-res when ups_in)."""
    KH, KW, Cin, Cout = w_hwio.shape
    c = _conv(dy, Cin, Cout, KH, stride, 0, ups_in, False, H=H, W=W)
    c.prep(w_hwio.contiguous(), fwd=False)
    return c.dgrad(dy.contiguous(), relu_mask)


@_impl("conv2d_nhwc_wgrad")
def _conv_wgrad(x, dy, KH, KW, Cin, Cout, stride, ups_in):
    assert KH == KW
    if ups_in:
        # the fused-resize weight gradient exists for the model's decoder geometries (wgrad_tile.hip); any other layer
        # materialises the 2x bilinear resize first (same numbers: the fused staging uses the resize kernel's arithmetic)
        try:
            return _conv(x, Cin, Cout, KH, stride, 0, True, False).wgrad(x.contiguous(), dy.contiguous(), workspace=True)
        except _lib.SplitVaeError as e:
            if "SV_E_UNSUPPORTED" not in str(e):
                raise
        x = ops.upsample2x_fwd(x.contiguous())
    c = _conv(x, Cin, Cout, KH, stride, 0, False, False)
    return c.wgrad(x.contiguous(), dy.contiguous(), workspace=True)


@_impl("dlogistic_nll")
def _dlogistic(images6, ch_off, out6, grad_scale, bf16_grad):
    """-> (nll [B] per-image sums, grad [B,H,W,8] = grad_scale * d nll / d out6 in channels 0..5)."""
    return ops.dlogistic_nll(images6.contiguous(), ch_off, out6.contiguous(), torch.bfloat16 if bf16_grad else torch.float32,
                             grad_scale)


@_impl("reparam_kl_fwd")
def _reparam_fwd(pre, bias, eps, seed, step, stream_id, sample_offset):
    z_mean, z_sig, z, _, kl, eps_out = ops.reparam_kl_fwd(pre.contiguous(), bias.contiguous(), eps, torch.float32, seed, step,
                                                          stream_id, sample_offset)
    return z, z_mean, z_sig, kl, eps_out


@_impl("reparam_kl_bwd")
def _reparam_bwd(dz, z_mean, z_sig, eps, kl_scale):
    return ops.reparam_kl_bwd(dz.contiguous(), z_mean, z_sig, eps, kl_scale, torch.float32)


@_impl("upsample2x_fwd")
def _ups_fwd(x):
    return ops.upsample2x_fwd(x.contiguous())


@_impl("upsample2x_bwd")
def _ups_bwd(g_hi, relu_mask):
    return ops.upsample2x_bwd(g_hi.contiguous(), relu_mask)


@_impl("adam_step")
def _adam(p, g, m, v, t, lr, beta1, beta2, eps, grad_scale):
    ops.adam_step(p, g, m, v, t, lr, beta1, beta2, eps, grad_scale)


# ---------------------------------------------------------------------------------------------- autograd pairing
class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_hwio, bias, stride, act, ups_in, y_f32):
        a = ACT[act]
        y = torch.ops.split_vae.conv2d_nhwc_fwd(x, w_hwio, bias, stride, a, ups_in, y_f32)
        ctx.save_for_backward(x, w_hwio, y if a else None)
        ctx.cfg = (stride, a, ups_in, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, a, ups_in, has_bias = ctx.cfg
        KH, KW, Cin, Cout = w.shape
        gy = gy.to(x.dtype)
        if gy.shape[-1] != (Cout + 7) // 8 * 8:                       # fp32 head (y_f32): pad the gradient to the 8-channel pitch
            gy = torch.nn.functional.pad(gy, (0, (Cout + 7) // 8 * 8 - gy.shape[-1]))
        if a:                                                         # this layer's ReLU: dY * (y > 0)
            gy = gy * (y > 0).to(gy.dtype)
        gy = gy.contiguous()
        H, W = x.shape[1] * (2 if ups_in else 1), x.shape[2] * (2 if ups_in else 1)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gy_d, w_d = gy, w
            P = 1 << (gy.shape[-1] - 1).bit_length()
            if P != gy.shape[-1]:                                     # the input-gradient kernels index dY by a power-of-two pixel pitch
                gy_d = torch.nn.functional.pad(gy, (0, P - gy.shape[-1]))          # (SPAIR's 100-channel z3): zero channels, zero filters
                w_d = torch.nn.functional.pad(w, (0, P - Cout))
            gx = torch.ops.split_vae.conv2d_nhwc_dgrad(gy_d, w_d, None, H, W, x.shape[-1], stride, ups_in)
            if ups_in:                                                # adjoint of the fused bilinear resize
                gx = torch.ops.split_vae.upsample2x_bwd(gx, None)
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw, gb = torch.ops.split_vae.conv2d_nhwc_wgrad(x, gy, KH, KW, Cin, Cout, stride, ups_in)
        return gx, gw, (gb if has_bias else None), None, None, None, None


def conv2d(x, w_hwio, bias=None, stride=1, act=None, ups_in=False, y_f32=False):
    """Conv2D(padding='same') of vae/model.py:36-38,:153-156 on the MFMA kernels, differentiable in x, w, bias.
    x is NHWC with its channel pitch padded to a multiple of 8 (pad channels must be zero)."""
    return _Conv2dFn.apply(x, w_hwio, bias, stride, act, ups_in, y_f32)


class _DlogisticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images6, ch_off, out6):
        nll, grad = torch.ops.split_vae.dlogistic_nll(images6, ch_off, out6, 1.0, False)
        ctx.save_for_backward(grad)
        return nll

    @staticmethod
    def backward(ctx, g_nll):
        (grad,) = ctx.saved_tensors                                   # d nll_b / d out6 (fused into the forward kernel)
        return None, None, grad[..., :6] * g_nll.view(-1, 1, 1, 1)


def dlogistic_nll(images6, ch_off, out6):
    """Per-image discretised-logistic NLL sums (vae/trainer.py:21-38 + reduce_sum[1,2,3], :127): images6 [B,H,W,6] fp32
    (x at channels ch_off..ch_off+2), out6 [B,H,W,6] fp32 (mean | log_scale) -> [B]; differentiable in out6."""
    return _DlogisticFn.apply(images6, ch_off, out6)


class _ReparamKlFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, bias, eps):
        z, z_mean, z_sig, kl, eps_out = torch.ops.split_vae.reparam_kl_fwd(pre, bias, eps, 0, 0, 0, 0)
        ctx.save_for_backward(z_mean, z_sig, eps_out)
        ctx.mark_non_differentiable(z_mean, z_sig)
        return z, kl, z_mean, z_sig

    @staticmethod
    def backward(ctx, gz, gkl, _gm, _gs):
        z_mean, z_sig, eps = ctx.saved_tensors
        # the kernel applies ONE scalar KL weight: d(sum_b w_b kl_b) needs a uniform w_b (batch mean * beta: what the step uses)
        w = gkl.reshape(-1)
        if not bool((w == w[0]).all()):
            raise NotImplementedError("reparam_kl backward takes a uniform KL cotangent (e.g. beta * kl.mean())")
        g = torch.ops.split_vae.reparam_kl_bwd(gz.contiguous(), z_mean, z_sig, eps, float(w[0]))
        L = z_mean.shape[1]
        gb = g.sum(dim=0)
        return g, torch.cat([gb[:L], gb[L:]]), None


def reparam_kl(pre, bias, eps):
    """Sampling + KL (vae/model.py:9-13,:111-113; vae/trainer.py:11-15) on the head pre-activations pre [B,2L] (mean |
    sd, before bias and softplus), bias [2L], eps [B,L] -> (z, kl [B] per-image terms, z_mean, z_sig); differentiable in
    pre and bias through z and a uniformly weighted kl."""
    return _ReparamKlFn.apply(pre, bias, eps)


class _Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return torch.ops.split_vae.upsample2x_fwd(x)

    @staticmethod
    def backward(ctx, g):
        return torch.ops.split_vae.upsample2x_bwd(g.contiguous(), None)


def upsample2x(x):
    """tf.image.resize to twice the size (bilinear, half-pixel centres; vae/model.py:163-167, spair/spair.py:175-180) as its own
    op, for layers whose geometry the fused-resize conv (ups_in) does not cover.  x NHWC, channels % 8 == 0 (fp32: % 4)."""
    return _Upsample2xFn.apply(x)


# ---------------------------------------------------------------------------------------------- SPLIT-SPAIR operators
@_impl("stn_sample_fwd")
def _stn_fwd(img, z_where, Ho, Wo, inverse):
    return ops.stn_sample(img, z_where, Ho, Wo, inverse=inverse)


@_impl("stn_sample_bwd")
def _stn_bwd(img, z_where, g_out, inverse, need_img):
    g_img, g_z = ops.stn_sample_bwd(img, z_where, g_out, inverse=inverse, need_img=need_img)
    return (g_img if need_img else img.new_empty((0,))), g_z


@_impl("spair_render_fwd")
def _render_fwd(obj, bg, z_depth, z_pres, z_pres_logits, noise, training):
    return ops.spair_render(obj, bg, z_depth, z_pres=z_pres, z_pres_logits=z_pres_logits, training=training, noise=noise)


@_impl("spair_render_bwd")
def _render_bwd(obj, bg, z_depth, z_pres, noise, g_out):
    return ops.spair_render_bwd(obj, bg, z_depth, z_pres, g_out, noise=noise)


@_impl("spair_zpres_kl")
def _zpres_kl(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob, temperature, prior_prob_dev=None):
    return ops.spair_zpres_kl(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob if prior_prob_dev is None else prior_prob_dev,
                              temperature, grad_scale=1.0)


class _StnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, z_where, Ho, Wo, inverse):
        out, bbox = torch.ops.split_vae.stn_sample_fwd(img, z_where, Ho, Wo, inverse)
        ctx.save_for_backward(img, z_where)
        ctx.inverse = inverse
        ctx.mark_non_differentiable(bbox)          # obj_bbox_mask only feeds metrics / drawings in the reference
        return out, bbox

    @staticmethod
    def backward(ctx, g_out, _g_bbox):
        img, z_where = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0]                   # the glimpse STN reads the input image: no gradient, no scatter
        g_img, g_z = torch.ops.split_vae.stn_sample_bwd(img, z_where, g_out.contiguous(), ctx.inverse, need_img)
        return (g_img if need_img else None), g_z, None, None, None


def stn_sample(img, z_where, Ho, Wo, inverse=False):
    """STN.call (spair/utils.py:119-200): glimpses / pasted objects [B,B',Ho,Wo,C] and obj_bbox_mask; differentiable in img
    and z_where."""
    return _StnFn.apply(img, z_where, Ho, Wo, inverse)


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, bg, z_depth, z_pres, noise):
        out = torch.ops.split_vae.spair_render_fwd(obj, bg, z_depth, z_pres, None, noise, True)
        ctx.save_for_backward(obj, bg, z_depth, z_pres, noise)
        return out

    @staticmethod
    def backward(ctx, g_out):
        obj, bg, z_depth, z_pres, noise = ctx.saved_tensors
        g_obj, g_bg, g_zp, g_zd = torch.ops.split_vae.spair_render_bwd(obj, bg, z_depth, z_pres, noise, g_out.contiguous())
        return g_obj, g_bg, g_zd.reshape(z_depth.shape), g_zp.reshape(z_pres.shape), None


def spair_render(obj, bg, z_depth, z_pres, noise=None):
    """Renderer.call, training form (spair/spair.py:534-579); differentiable in obj, bg, z_depth, z_pres.  (The test form
    rounds the presences: call torch.ops.split_vae.spair_render_fwd with training=False.)"""
    return _RenderFn.apply(obj, bg, z_depth, z_pres, noise)


class _ZpresKlFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob, temperature):
        dev = prior_prob if torch.is_tensor(prior_prob) else None      # a device scalar: read at run time (hipGraph replay)
        kl, g_pre, g_log = torch.ops.split_vae.spair_zpres_kl(z_pres, z_pres_logits, z_pres_pre_sigmoid, 0.0 if dev is not None else prior_prob,
                                                              temperature, dev)
        ctx.save_for_backward(g_pre, g_log)
        return kl

    @staticmethod
    def backward(ctx, g_kl):
        g_pre, g_log = ctx.saved_tensors
        w = g_kl.view(-1, *([1] * (g_pre.dim() - 1)))
        return None, g_log * w, g_pre * w, None, None     # no gradient to z_pres: the count prior sees thresholded samples only


def spair_zpres_kl(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob, temperature):
    """compute_z_pres_kl_yolo_air (spair/trainer.py:45-94) as per-image sums kl [B] (tf_mean_sum = kl.mean()); differentiable
    in the logits and the pre-sigmoid sample."""
    return _ZpresKlFn.apply(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob if torch.is_tensor(prior_prob) else float(prior_prob),
                            float(temperature))


@_impl("spair_loss")
def _spair_loss(mode, a, b, prior_mean, prior_sig, prior_mean_dev=None):
    sums, ga, gb = ops.spair_loss(mode, a, b, prior_mean if prior_mean_dev is None else prior_mean_dev, prior_sig)
    return sums, (ga if ga is not None else a.new_empty((0,))), gb


class _SpairLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mode, a, b, prior_mean, prior_sig):
        dev = prior_mean if torch.is_tensor(prior_mean) else None
        sums, ga, gb = torch.ops.split_vae.spair_loss(mode, a, b, 0.0 if dev is not None else prior_mean, prior_sig, dev)
        ctx.save_for_backward(ga, gb)
        ctx.has_ga = mode != "xent"
        return sums

    @staticmethod
    def backward(ctx, g_sums):
        ga, gb = ctx.saved_tensors
        w = g_sums.view(-1, *([1] * (gb.dim() - 1)))
        return None, (ga * w if ctx.has_ga and ctx.needs_input_grad[1] else None), gb * w, None, None


def spair_xent(label, pred):
    """Per-image sums of xent_loss (spair/trainer.py:103-104) [B]; differentiable in pred."""
    return _SpairLossFn.apply("xent", label, pred, 0.0, 1.0)


def spair_kl(z_mean, z_sig):
    """Per-image sums of kl_divergence's integrand (spair/trainer.py:13-21) [B]; differentiable in both."""
    return _SpairLossFn.apply("kl", z_mean, z_sig, 0.0, 1.0)


def spair_kl_prior(mean, sig, prior_mean, prior_sig):
    """Per-image sums of kl_divergence_two_gauss (spair/trainer.py:23-24) against the constant prior N(prior_mean, prior_sig)."""
    return _SpairLossFn.apply("kl_prior", mean, sig, prior_mean if torch.is_tensor(prior_mean) else float(prior_mean), float(prior_sig))
