/*
 * splitvae.h -- C ABI of libsplitvae_hip.so: the MI355X (gfx950) SPLIT-VAE training path.
 *
 * The reference (51616/split-vae) has no FFI/plugin layer: its hot path is Python calling
 * TensorFlow-2.0 library kernels.  Every entry point below therefore cites the reference *call
 * site(s)* whose TF op sequence it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain C, raw DEVICE pointers + explicit sizes, `stream` is a hipStream_t passed as void*;
 *   - the caller owns every buffer including the workspace; no entry point allocates device
 *     memory, synchronises the host, or keeps global mutable state (plans are caller-owned);
 *   - return 0 on success, <0 = sv_status error, >0 = a hipError_t from a launch;
 *   - activations are NHWC; conv kernels HWIO fp32; dense kernels [in,out] fp32 (Keras layouts);
 *   - `dtype` selects the arithmetic type of the MFMA contractions (SV_BF16: bf16 operands,
 *     fp32 accumulate; SV_F32: exact fp32 MFMA).  Master weights, ELBO terms, KL, Adam are fp32.
 */
#ifndef SPLITVAE_H
#define SPLITVAE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { SV_OK = 0, SV_E_BADARG = -1, SV_E_UNSUPPORTED = -2, SV_E_WORKSPACE = -3, SV_E_STATE = -4 } sv_status;
typedef enum { SV_F32 = 0, SV_BF16 = 1 } sv_dtype;
typedef enum { SV_ACT_NONE = 0, SV_ACT_RELU = 1, SV_ACT_ELU = 2 /* pointwise kernels of the GMVAE encoder only */ } sv_act;

const char* sv_version(void);

/* Fixed-order reductions for every launch that FOLLOWS: 1 = on (no split-K / m-split fp32 atomics anywhere: same inputs twice ->
 * identical bits; SURVEY section 5's determinism test), 0 = off (the faster atomics-ordered dense layers), -1 = what the environment
 * variable SV_DETERMINISTIC said at load time (the default).  Host-side, per launch: plans, workspaces and prepared weights do not
 * depend on it.  The reference has no such switch (TF-2.0 GPU kernels are nondeterministic, no seed is set anywhere: SURVEY 0);
 * the parity tests use it to compare against the oracle at the bounds of SURVEY 8c without a summation-order allowance. */
int sv_set_deterministic(int32_t mode);
int sv_get_deterministic(void);

/* ---------------------------------------------------------------- A1: patch scramble
 * Replaces Augmentator.scramble (augmentation.py:43-57) as wired at vae/main.py:54-61:
 * extract_patches -> reshape -> shuffle -> split/unstack/concat -> concat([x, x_aug], axis=2).
 * x[B,H,W,3] fp32, perm[B,(H/patch)*(W/patch)] int32 (destination patch n takes source patch
 * perm[n]); images6[B,H,W,6] fp32 = [x | x_aug].  Pure index bookkeeping: bit-exact. */
int sv_scramble_gather(const float* x, const int32_t* perm, float* images6,
                       int32_t B, int32_t H, int32_t W, int32_t patch, void* stream);
/* The same, also writing the two zero-padded 8-channel NHWC tensors (x | x_aug, sv_dtype `dtype`) the first encoder layers of
 * sv_lgvae_step read -- pass the plan's in8_x / in8_xh buffers (sv_lgvae_buffer) and SV_PHASE_INPUTS_STAGED to the step: images6 is
 * read once less and the step's split / pad pass drops out (vae/main.py:57-61 + vae/model.py:190 in one kernel). */
int sv_scramble_gather_staged(const float* x, const int32_t* perm, float* images6, void* x8, void* xh8, int32_t dtype, int32_t B,
                              int32_t H, int32_t W, int32_t patch, void* stream);
/* tf.random.shuffle (augmentation.py:49) stand-in: one uniform permutation per image from a
 * counter-based Philox stream keyed by (seed, step, global sample index = sample_offset + b),
 * so 1-GPU and N-GPU runs draw identical permutations.  n_patch <= 4096. */
int sv_random_perm(int32_t* perm, int32_t B, int32_t n_patch, uint64_t seed, uint64_t step,
                   int64_t sample_offset, void* stream);

/* ---------------------------------------------------------------- A6: discretised logistic NLL
 * Replaces discretised_logistic_loss (vae/trainer.py:21-38) + reduce_sum[1,2,3] (:127-128) and,
 * when grad != NULL, its adjoint under tape.gradient (:137).
 * images6[B,H,W,6] fp32; channels [ch_off, ch_off+3) are the targets (0: x, 3: x_hat).
 * out6[B,H,W,6] fp32 decoder head: ch 0-2 mean, 3-5 log_scale (vae/model.py:169).
 * nll[B] fp32 per-image sums.  grad[B,H,W,8] (dtype) = d(mean_b nll)/d(out6) * grad_scale in
 * ch 0-5, zeros in ch 6-7 (the padded layout the conv dgrad/wgrad kernels consume). */
int sv_dlogistic_nll(const float* images6, int32_t ch_off, const float* out6, float* nll,
                     void* grad, int32_t grad_dtype, float grad_scale,
                     int32_t B, int32_t H, int32_t W, float* partial_ws, void* stream);
int64_t sv_dlogistic_nll_workspace_bytes(int32_t B, int32_t H, int32_t W);

/* ---------------------------------------------------------------- A4+A7: reparameterise + KL
 * Replaces Sampling.call (vae/model.py:9-13), the Dense bias/softplus epilogues of e4_mean/e4_sd
 * (:41-42,:111-112) and kl_divergence (vae/trainer.py:11-15).
 * pre[B,2L] fp32 = [f@W_mean | f@W_sd] WITHOUT bias; bias[2L]; eps[B,L] (NULL: drawn from Philox
 * keyed by (seed, step, stream_id, sample_offset+b, j) and written to eps_out).
 * Outputs: z_mean,z_sig,z [B,L] fp32; z_lp (dtype) [B, ldz] at column z_col (decoder input,
 * the tf.concat of vae/model.py:197 is this write); kl[B] = -1/2 sum_j(1+log sig^2-mu^2-sig^2). */
int sv_reparam_kl_fwd(const float* pre, const float* bias, const float* eps, float* eps_out,
                      float* z_mean, float* z_sig, float* z, void* z_lp, int32_t z_dtype,
                      int32_t ldz, int32_t z_col, float* kl, int32_t B, int32_t L,
                      uint64_t seed, uint64_t step, int32_t stream_id, int64_t sample_offset,
                      void* stream);
/* Adjoint: dz[B,L] fp32 (ld_dz, optional second addend dz2) -> g_pre[B,2L] (dtype) =
 * [dz + kl_scale*mu | (dz*eps + kl_scale*(sig-1/sig)) * (1-exp(-sig))]; kl_scale = beta/B. */
int sv_reparam_kl_bwd(const float* dz, int32_t ld_dz, const float* dz2, int32_t ld_dz2,
                      const float* z_mean, const float* z_sig, const float* eps, float kl_scale,
                      void* g_pre, int32_t g_dtype, int32_t B, int32_t L, void* stream);

/* ---------------------------------------------------------------- K14: Keras Adam
 * Replaces tf.keras.optimizers.Adam(lr).apply_gradients (vae/main.py:65, vae/trainer.py:138):
 * alpha=lr*sqrt(1-b2^t)/(1-b1^t); m+=(g-m)(1-b1); v+=(g*g-v)(1-b2); p-=alpha*m/(sqrt(v)+eps).
 * One launch over the flat parameter buffer; g is multiplied by grad_scale first (1/world). */
int sv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t t, float grad_scale, void* stream);

/* The same with Keras `clipnorm` (SPLIT-SPAIR: Adam(..., clipnorm=1.0), spair/main.py:109): every gradient tensor is scaled by
 * clipnorm / max(||g * grad_scale||_2, clipnorm) first (tf.clip_by_norm).  tensor_off: DEVICE array of n_tensors+1 element
 * offsets into the flat buffers; norm_ws: DEVICE scratch of 256*n_tensors floats.  Deterministic (fixed-order norms). */
int sv_adam_step_clipnorm(float* p, const float* g, float* m, float* v, const int64_t* tensor_off, int32_t n_tensors,
                          float* norm_ws, float clipnorm, float lr, float beta1, float beta2, float eps, int64_t t,
                          float grad_scale, void* stream);

/* hipGraph-replay form: the step size alpha = lr*sqrt(1-b2^t)/(1-b1^t) is read from DEVICE memory (alpha_dev, one float the
 * caller refreshes before each replay with sv_adam_alpha(lr, beta1, beta2, t)); alpha_dev == NULL = the call above. */
int sv_adam_step_clipnorm_dyn(float* p, const float* g, float* m, float* v, const int64_t* tensor_off, int32_t n_tensors,
                              float* norm_ws, float clipnorm, float lr, float beta1, float beta2, float eps, int64_t t,
                              const float* alpha_dev, float grad_scale, void* stream);
float sv_adam_alpha(float lr, float beta1, float beta2, int64_t t);
/* The same over SEPARATE gradient tensors: grads = HOST array of n_tensors (<= 128) DEVICE pointers, 16-byte aligned, tensor i
 * holding tensor_off[i+1] - tensor_off[i] floats.  The addresses are passed to the kernels by value: no flat copy of the
 * gradients, and a captured hipGraph stays valid while the tensors keep their addresses.  alpha_dev may be NULL. */
int sv_adam_step_clipnorm_ptrs(float* p, const float* const* grads, float* m, float* v, const int64_t* tensor_off, int32_t n_tensors,
                               float* norm_ws, float clipnorm, float lr, float beta1, float beta2, float eps, int64_t t,
                               const float* alpha_dev, float grad_scale, void* stream);

/* ---------------------------------------------------------------- K10a: bilinear 2x
 * Replaces tf.image.resize(x,[2H,2W]) (vae/model.py:163,:165,:167; bilinear, half-pixel centres,
 * edge clamp) and its adjoint (ResizeBilinearGrad) fused with the ReLU mask of the producer. */
int sv_upsample2x_fwd(const void* in, void* out, int32_t dtype, int32_t B, int32_t H, int32_t W,
                      int32_t C, void* stream);
int sv_upsample2x_bwd(const void* g_hi, const void* y_lo_mask, void* g_lo, int32_t dtype,
                      int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* ---------------------------------------------------------------- SPLIT-SPAIR: spatial transformer (fp32)
 * Replaces spair.utils.STN.call + bilinear_sampler + get_pixel_value (spair/utils.py:119-200, :202-272, :274-330) and the
 * gradient tape.gradient takes through them.  z_where [B,Hc,Wc,4] pre-activations; inverse = 0: img [B,H,W,C] -> one
 * [Ho,Wo,C] glimpse per cell, out [B,Hc*Wc,Ho,Wo,C]; inverse = 1 (the renderer's STN): img [B,Hc*Wc,H,W,C] -> each object
 * pasted on its own [Ho,Wo,C] canvas.  bbox (may be NULL) = obj_bbox_mask [B,Hc*Wc,4].  Backward: g_z_where [B,Hc,Wc,4] and
 * g_img (same shape as img; may be NULL when the image takes no gradient).  g_img is scatter-ADDED with atomics -- zero it
 * first -- unless sv_stn_bwd_overwrites(H, W, C, inverse) is 1 (inverse form, small objects: each cell's gradient is
 * accumulated in LDS and stored whole). */
int sv_stn_sample_fwd(const float* img, const float* z_where, float* out, float* bbox, int32_t B, int32_t Hc, int32_t Wc,
                      int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t inverse, void* stream);
int sv_stn_bwd_overwrites(int32_t H, int32_t W, int32_t C, int32_t inverse);
int sv_stn_sample_bwd(const float* img, const float* z_where, const float* g_out, float* g_img, float* g_z_where, int32_t B,
                      int32_t Hc, int32_t Wc, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t inverse,
                      void* stream);

/* SPLIT-SPAIR: Renderer.call (spair/spair.py:534-579), fp32.  obj [B,B',H,W,C+1] = every object's rgb + alpha on its own
 * canvas (sv_stn_sample_fwd, inverse = 1), bg [B,H,W,C], z_depth / z_pres / z_pres_logits [B,B'] -> out [B,H,W,C]; B' <= 16,
 * C <= 4.  training != 0: z_pres as given and `noise` [B,B',H,W,C] (the GaussianNoise(0.01) draw, may be NULL) added to the
 * object images before their clip; training == 0: z_pres = max(round(sigmoid(z_pres_logits)), 1e-8), noise must be NULL.
 * Backward (training form): g_obj like obj, g_bg like bg, g_z_pres / g_z_depth [B,B'] (all written, none accumulated). */
int sv_spair_render_fwd(const float* obj, const float* bg, const float* z_depth, const float* z_pres,
                        const float* z_pres_logits, const float* noise, float* out, int32_t B, int32_t Bp, int32_t H,
                        int32_t W, int32_t C, int32_t training, void* stream);
int sv_spair_render_bwd(const float* obj, const float* bg, const float* z_depth, const float* z_pres, const float* noise,
                        const float* g_out, float* g_obj, float* g_bg, float* g_z_pres, float* g_z_depth, int32_t B,
                        int32_t Bp, int32_t H, int32_t W, int32_t C, void* stream);

/* The same with a workspace of sv_spair_render_bwd_workspace_floats(B, H, W) floats: a workgroup per 256-pixel chunk instead of
 * one per image (288 instead of 32 workgroups at batch 32); the chunks' partial g_z sums are added in chunk order (deterministic). */
int64_t sv_spair_render_bwd_workspace_floats(int32_t B, int32_t H, int32_t W);
int sv_spair_render_bwd_ws(const float* obj, const float* bg, const float* z_depth, const float* z_pres, const float* noise,
                           const float* g_out, float* g_obj, float* g_bg, float* g_z_pres, float* g_z_depth, int32_t B,
                           int32_t Bp, int32_t H, int32_t W, int32_t C, float* ws, int64_t ws_floats, void* stream);

/* SPLIT-SPAIR: compute_z_pres_kl_yolo_air + concrete_binary_sample_kl (spair/trainer.py:28-42, :45-94), fp32: the sequential
 * count-prior KL of the n_cells <= 16 presence variables, cells in raster order.  z_pres / z_pres_logits / z_pres_pre_sigmoid
 * [B,n_cells] -> kl [B] (per-image sums: tf_mean_sum = their batch mean); g_pre_sigmoid / g_logits (may be NULL) =
 * grad_scale * d kl_b / d input (the prior depends on the thresholded samples only: no gradient to z_pres). */
int sv_spair_zpres_kl(const float* z_pres, const float* z_pres_logits, const float* z_pres_pre_sigmoid, float* kl,
                      float* g_pre_sigmoid, float* g_logits, int32_t B, int32_t n_cells, float prior_prob, float temperature,
                      float grad_scale, void* stream);

/* hipGraph-replay form: prior_prob_dev (may be NULL) = one DEVICE float that overrides prior_prob (the prior anneals with the
 * step, spair/trainer.py:153). */
int sv_spair_zpres_kl_dyn(const float* z_pres, const float* z_pres_logits, const float* z_pres_pre_sigmoid, float* kl,
                          float* g_pre_sigmoid, float* g_logits, int32_t B, int32_t n_cells, float prior_prob,
                          const float* prior_prob_dev, float temperature, float grad_scale, void* stream);

/* SPLIT-SPAIR: the per-image loss sums of spair/trainer.py with their gradients, fp32, one launch each (spair_loss.hip):
 *   mode 0: xent_loss :103-104, a = label, b = prediction;  mode 1: kl_divergence :13-21, a = z_mean, b = z_sig;
 *   mode 2: kl_divergence_two_gauss :23-24 against the constant prior N(prior_mean, prior_sig) (:156-157), a = mean, b = sig.
 * a, b [B,n] -> sums [B] (tf_mean_sum :107-109 = their batch mean); ga / gb [B,n] (may be NULL) = the per-element derivatives
 * (mode 0: ga is ignored).  tf_safe_log :97-101 semantics (log(v + 1e-8); NaN / inf -> -100, no gradient). */
int sv_spair_loss(int32_t mode, const float* a, const float* b, float* sums, float* ga, float* gb, int32_t B, int32_t n,
                  float prior_mean, float prior_sig, void* stream);
/* hipGraph-replay form: prior_mean_dev (may be NULL) = one DEVICE float that overrides prior_mean (mode 2: the zoom prior
 * anneals with the step, spair/trainer.py:156). */
int sv_spair_loss_dyn(int32_t mode, const float* a, const float* b, float* sums, float* ga, float* gb, int32_t B, int32_t n,
                      float prior_mean, const float* prior_mean_dev, float prior_sig, void* stream);

/* ---------------------------------------------------------------- K3-K10: NHWC conv (implicit GEMM on MFMA)
 * Replaces tf.keras.layers.Conv2D(padding='same') forward (vae/model.py:36-38,:153-156) and the
 * Conv2DBackpropInput / Conv2DBackpropFilter / BiasAddGrad / ReluGrad nodes of tape.gradient
 * (vae/trainer.py:137).  Dense layers (vae/model.py:41-42,:152) are the H=W=KH=KW=1 case.
 * Channel counts are padded to a multiple of 8 in the low-precision activation tensors. */
typedef struct {
  int32_t B, H, W;          /* input spatial size (power-of-two H, W) */
  int32_t Cin, Cout;        /* real channel counts (HWIO weight shape = [KH,KW,Cin,Cout]) */
  int32_t KH, KW, stride;   /* TF 'SAME' padding is implied: out=ceil(in/stride) */
  int32_t act;              /* sv_act fused into the forward epilogue */
  int32_t dtype;            /* sv_dtype of activations / prepared weights */
  int32_t ldx;              /* channels per input pixel in memory  (>= Cin, multiple of 8) */
  int32_t ldy;              /* channels per output pixel in memory (>= Cout; multiple of 8 unless y_f32) */
  int32_t y_f32;            /* forward output written as fp32 (decoder head) */
  int32_t ups_in;           /* x is the LOW-RES tensor [B,H/2,W/2,ldx] and the layer input is its 2x bilinear
                               upsample (tf.image.resize, vae/model.py:163-167), produced on the fly while the
                               tile is staged: forward and wgrad only, stride 1, tile kernels (else SV_E_UNSUPPORTED) */
} sv_conv_desc;

/* element counts (of `dtype`) of the prepared forward / dgrad weight images */
int64_t sv_conv2d_wprep_elems(const sv_conv_desc* d, int32_t for_dgrad);
/* fp32 HWIO master -> MFMA-ready [Cout_pad][taps][Cin_pad] (fwd) and per-parity-class
 * [Cin_pad][taps'][Cout_pad] (dgrad) images in `dtype`. */
int sv_conv2d_prep_weights(const sv_conv_desc* d, const float* w_hwio, void* w_fwd, void* w_dgrad,
                           void* stream);
int sv_conv2d_nhwc_fwd(const sv_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                       void* y, void* stream);
/* Same with a caller-owned workspace (sv_conv2d_fwd_workspace_bytes; 0 for most layers).  The bf16 decoder head
 * (UpSampling2D(bilinear) -> Conv2D(6, 6x6, 'same'), vae/model.py:163-167 + d5) runs in POLYPHASE form: one 5x5 conv over
 * the low-res tensor whose four output parities are four column classes, plus 1-D border terms for the taps that leave the
 * zero-padded hi-res image; with a workspace those terms are computed first and added by the conv's epilogue, without one
 * they are added to y with atomics afterwards (same result, slower).  The fp32 upsample -> 6 x 6 conv layers with 32 output channels (d4) run per output-parity
 * class over the low-res tensor and need the workspace for their border terms; WITHOUT one the call runs the direct fused-resize form on a second
 * weight image that sv_conv2d_prep_weights keeps behind the class images (sv_conv2d_wprep_elems counts it) -- same result to fp32 rounding, 1.4x the time.
 * Form selection is a pure function of the descriptor (and of the SV_NO_POLY* / SV_POLYC_K tuning variables, which must not change between
 * sv_conv2d_prep_weights and the calls that consume its images). */
int64_t sv_conv2d_fwd_workspace_bytes(const sv_conv_desc* d);
int sv_conv2d_nhwc_fwd_ws(const sv_conv_desc* d, const void* x, const void* w_fwd, const float* bias, void* y,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* dx = conv-transpose(dy, w) * (mask>0 if mask!=NULL); dy[B,OH,OW,ldy], dx[B,H,W,ldx] (dtype).
 * dx_f32_atomic!=0: K is split over workgroups and dx (fp32, pre-zeroed) is accumulated atomically. */
int sv_conv2d_nhwc_dgrad(const sv_conv_desc* d, const void* dy, const void* w_dgrad,
                         const void* relu_mask, void* dx, int32_t dx_f32_atomic, void* stream);
/* The same for a layer with ups_in (its input is the 2x bilinear upsample of a low-res tensor, vae/model.py:163-167),
 * delivered AT THE LOW-RES TENSOR: dx_lo[B,H/2,W/2,ldx] = (relu_mask_lo > 0) * resize_adjoint(conv-transpose(dy, w)) in
 * one launch -- ResizeBilinearGrad + ReluGrad of tape.gradient (vae/trainer.py:137) fused into the Conv2DBackpropInput
 * kernel; the hi-res gradient never reaches HBM.  SV_E_UNSUPPORTED when the geometry has no fused kernel: call
 * sv_conv2d_nhwc_dgrad followed by sv_upsample2x_bwd instead (bitwise the same result). */
int sv_conv2d_nhwc_dgrad_lowres(const sv_conv_desc* d, const void* dy, const void* w_dgrad,
                                const void* relu_mask_lo, void* dx_lo, void* stream);
/* The same with a caller-owned workspace (sv_conv2d_dgrad_lowres_workspace_bytes; 0: the layer needs none).  At the reference's precision (SV_F32) the
 * 6 x 6 upsample -> conv layers (vae/model.py:155-156 behind :165 / :167: d4, d5) take the POLYPHASE form: the transposed conv and the resize adjoint collapse
 * into one stride-2 conv with 9 x 9 taps over the hi-res dy (81 tap products per low-res pixel instead of 4 x 36), the zero-padding / edge-clamp terms of the
 * first and last low-res row and column travel through the workspace (csrc/polyd_dgrad.hip).  Without a (large enough) workspace: sv_conv2d_nhwc_dgrad_lowres. */
int64_t sv_conv2d_dgrad_lowres_workspace_bytes(const sv_conv_desc* d);
int sv_conv2d_nhwc_dgrad_lowres_ws(const sv_conv_desc* d, const void* dy, const void* w_dgrad, const void* relu_mask_lo, void* dx_lo,
                                   void* workspace, int64_t workspace_bytes, void* stream);
/* dw[KH,KW,Cin,Cout] += x^T*dy, dbias[Cout] += colsum(dy) (fp32 HWIO, atomically accumulated:
 * zero them first). */
int sv_conv2d_nhwc_wgrad(const sv_conv_desc* d, const void* x, const void* dy, float* dw,
                         float* dbias, void* stream);
/* Same with a caller-owned partial-sum workspace (sv_conv2d_wgrad_workspace_bytes): the tile kernel
 * then writes per-split slabs and reduces them in a fixed order -- faster than atomics and
 * bitwise reproducible.  dw is still accumulated into (zero it first). */
int64_t sv_conv2d_wgrad_workspace_bytes(const sv_conv_desc* d);
int sv_conv2d_nhwc_wgrad_ws(const sv_conv_desc* d, const void* x, const void* dy, float* dw,
                            float* dbias, void* workspace, int64_t workspace_bytes, void* stream);

/* Weight gradient of the bf16 decoder head (UpSampling2D(bilinear) -> Conv2D(6x6), vae/model.py:163-169) in POLYPHASE form: the
 * weight gradient of the 5x5 low-res conv (no blend arithmetic, a quarter of the pixels staged), projected back onto the 6x6
 * kernel, minus the out-of-image taps of the five border rows / columns (tests/test_polyphase_math.py).  Same result
 * as sv_conv2d_nhwc_wgrad on the same layer to bf16 accuracy (closer to the fp64 gradient: the upsampled activations are never
 * rounded).  Workspace: sv_conv2d_wgrad_poly_workspace_bytes (0 = the layer has no polyphase form), ZEROED before its first use. */
int64_t sv_conv2d_wgrad_poly_workspace_bytes(const sv_conv_desc* d);
int sv_conv2d_nhwc_wgrad_poly(const sv_conv_desc* d, const void* x_lo, const void* dy, float* dw, float* dbias, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- F3: input files (host side, no device work)
 * CRC-32C of the TFRecord framing the reference's CelebA files use (tf.io.TFRecordWriter at vae/data.py:93-100,
 * TFRecordDataset at :123-131): record = uint64 length | masked_crc32c(length) | data | masked_crc32c(data),
 * masked(c) = rotr(c, 15) + 0xa282ead8.  sv_crc32c("123456789") == 0xE3069283. */
uint32_t sv_crc32c(const void* data, int64_t n);
uint32_t sv_masked_crc32c(const void* data, int64_t n);

/* ---------------------------------------------------------------- A9: SPLIT-GMVAE global encoder glue
 * The contractions of Encoder(type='gmvae') (vae/model.py:48-79, call_gmvae :116-135) run on the
 * sv_conv2d_* kernels (dense layers = 1x1 convs on a 1x1 grid); these are the pointwise pieces between
 * them.  `dtype` arguments are sv_dtype; "lp" tensors are in the contraction dtype with the row pitch
 * the next MFMA layer reads (padding columns are written as zeros).
 *
 * sv_act_fwd: x[r,c] = dropout(act(a[r,c])), c < C  (Conv2D/Dense activation='elu' :50-52,:55,:57,:64,:73;
 *   Dropout(rate) in training = x * keep / (1-rate) :56,:72 as called at :118,:129).  y_act (optional) receives
 *   the pre-dropout activation (act' is recovered from it in the backward).  keep_in (optional, [rows*C] 0/1)
 *   pins the mask; otherwise it is drawn from Philox keyed by (seed, step, stream_id, sample_offset + r /
 *   rows_per_sample, column) and written to keep_out.
 * sv_act_bwd: ga = (gx * keep/(1-rate) + gx2) * act'(y_act)   (gx2, y_act optional). */
int sv_act_fwd(const void* a, int32_t a_dtype, int32_t lda, void* y_act, void* x, int32_t x_dtype, int32_t ldx,
               int64_t rows, int32_t C, int32_t act, float drop_rate, const float* keep_in, float* keep_out,
               uint64_t seed, uint64_t step, int32_t stream_id, int64_t sample_offset, int32_t rows_per_sample,
               void* stream);
int sv_act_bwd(const void* gx, int32_t gx_dtype, int32_t ldg, const void* gx2, int32_t gx2_dtype, int32_t ldg2,
               const void* y_act, int32_t y_dtype, int32_t ldy, int32_t act, float drop_rate, const float* keep,
               void* ga, int32_t ga_dtype, int32_t ldga, int64_t rows, int32_t C, void* stream);
/* out = a + b (vae/model.py:130: h = h + h_top) */
int sv_add(const void* a, const void* b, void* out, int32_t dtype, int64_t n, void* stream);
/* The five Mean metrics of train_step_lg_gm_vae / test_step_lg_gm_vae (vae/trainer.py:157-173) and the total loss from the per-image terms
 * [B] fp32: out6 = {mean nll_x, mean kl_x, mean nll_xh, mean kl_xh, mean y_kl, out[0] + out[2] + beta (out[1] + out[3]) + alpha out[4]}.
 * One launch, fixed-order sums. */
int sv_gm_metrics(const float* nll_x, const float* kl_x, const float* nll_xh, const float* kl_xh, const float* y_kl, int32_t B,
                  float beta, float alpha, float* out6, void* stream);
/* Gumbel-softmax (vae/model.py:121-122): y = softmax((logits - log(-log u)) / tau, axis=1); u[B,K] (NULL: Philox,
 * written to u_out).  y[B,K] fp32 and y_lp[B,ld_lp] (zero padded).  K, ld_lp <= 128. */
int sv_gumbel_softmax_fwd(const float* logits, int32_t ld_logits, const float* u, float* u_out, float tau, float* y,
                          void* y_lp, int32_t lp_dtype, int32_t ld_lp, int32_t B, int32_t K, uint64_t seed,
                          uint64_t step, int64_t sample_offset, void* stream);
/* its adjoint plus the categorical term of train_step_lg_gm_vae (vae/trainer.py:161-165):
 * g_logits = (1/tau) y (gy - <y,gy>) + alpha_over_B * d/dlogits sum_k p_k (log(p_k + 1e-8) - log(1/K)), p = softmax(logits);
 * y_kl[b] = that sum for image b.  g_logits NULL: only y_kl (evaluation). */
int sv_gumbel_softmax_bwd(const float* gy, int32_t ldg, const float* y, const float* logits, int32_t ld_logits,
                          float tau, float alpha_over_B, void* g_logits, int32_t g_dtype, int32_t ld_out,
                          float* y_kl, int32_t B, int32_t K, void* stream);
/* posterior and prior heads (vae/model.py:124-125,:131-133; Sampling :9-13) from the four fp32 pre-activations
 * (bias included): z_mean = a_mean, z_sig = softplus(a_sig), z = z_mean + z_sig*eps, prior likewise; z_lp is the
 * decoder input (columns [z_col, z_col+L) of `zcat`); kl2[b] = kl_divergence_two_gauss row sum (vae/trainer.py:17-18). */
int sv_gm_head_fwd(const float* a_mean, const float* a_sig, const float* a_prior_mean, const float* a_prior_sig,
                   const float* eps, float* eps_out, float* z_mean, float* z_sig, float* z, float* prior_mean,
                   float* prior_sig, void* z_lp, int32_t lp_dtype, int32_t ldz, int32_t z_col, float* kl2,
                   int32_t B, int32_t L, uint64_t seed, uint64_t step, int64_t sample_offset, void* stream);
/* adjoint: dL/dz (decoder) + kl_scale * d kl2, through the two softplus heads, as the dY of the four Dense layers */
int sv_gm_head_bwd(const float* dz, int32_t ld_dz, const float* z_mean, const float* z_sig, const float* prior_mean,
                   const float* prior_sig, const float* eps, float kl_scale, void* g_a_mean, void* g_a_sig,
                   void* g_a_prior_mean, void* g_a_prior_sig, int32_t g_dtype, int32_t B, int32_t L, void* stream);

/* ---------------------------------------------------------------- A9: the SPLIT-GMVAE global encoder as one object
 * Replaces Encoder(type='gmvae').call_gmvae (vae/model.py:48-79, :116-135) and its adjoint in train_step_lg_gm_vae
 * (vae/trainer.py:146-173) with native launch sequences over one caller-owned workspace.  It plugs into an LGVae plan
 * created with external_global_encoder = 1: FWD_ENCODERS phase -> sv_gm_encoder_forward (writes z_x into zcat[:, :L])
 * -> FWD_DECODERS/LOSS/BWD_DECODERS phases -> sv_gm_encoder_backward (reads dL/dz_x from the plan's gz_x) -> ... */
typedef struct {
  int32_t B, H, W;          /* H == W, power of two >= 8 */
  int32_t latent;           /* global latent size L (power of two) */
  int32_t y_size;           /* categorical size K (2..128) */
  float tau;                /* Gumbel-softmax temperature */
  int32_t dtype;            /* sv_dtype of the contractions */
} sv_gm_desc;
typedef struct {
  const float* params;      /* flat fp32 [sv_gm_param_count]: the 24 arrays in the reference's variable order */
  float* grads;             /* backward: same layout, ACCUMULATED into (zero it first) */
  const void* in8_x;        /* [B,H,W,8] network input in `dtype` (the plan's buffer "in8_x") */
  void* zcat; int32_t ldz;  /* forward: decoder input rows (the plan's "zcat"), row pitch in elements */
  const float* gz; int32_t ld_gz;   /* backward: dL/dz_x rows (the plan's "gz_x"), columns [0, L) */
  const float* eps;         /* [B,L]     or NULL -> Philox        (pins for parity tests) */
  const float* u;           /* [B,K]     Gumbel uniforms or NULL */
  const float* keep1;       /* [B,1024]  y_block dropout mask (0/1) or NULL */
  const float* keep5;       /* [B,F]     do5 dropout mask or NULL */
  int32_t training;         /* 1: y_block's Dropout and do5 act (vae/model.py:56,:72,:129).  train_step_lg_gm_vae passes
                             training=True (vae/trainer.py:149) but LGGMVae.call drops it before encoder_x (:241): under the
                             pinned tensorflow 2.0.0 the dropouts never fire (pass 0), under >= 2.1 they do (pass 1) */
  float beta, alpha;        /* backward: weights of the two-Gaussian KL and of the categorical KL */
  uint64_t seed, step;
  int64_t sample_offset;
} sv_gm_args;
typedef struct sv_gm_encoder sv_gm_encoder;
int64_t sv_gm_param_count(const sv_gm_desc* d);
int sv_gm_param_info(const sv_gm_desc* d, int32_t index, int64_t* offset, int32_t* ndim, int64_t shape[4], char name[96]);
int sv_gm_encoder_create(const sv_gm_desc* d, sv_gm_encoder** enc);
void sv_gm_encoder_destroy(sv_gm_encoder* enc);
int64_t sv_gm_encoder_workspace_bytes(const sv_gm_encoder* enc);
int sv_gm_encoder_bind(sv_gm_encoder* enc, void* workspace, int64_t bytes, void* stream);   /* zero-fills the workspace, uploads the job table (as sv_lgvae_plan_bind) */
/* named activation / gradient buffers inside the workspace (z, zm, zs, pm, ps, y, logits, kl2, ykl, keep1, ...) */
int sv_gm_encoder_buffer(const sv_gm_encoder* enc, const char* name, int64_t* offset, int64_t* bytes);
/* fp32 masters -> MFMA-ready weight images of the twelve layers (after every parameter update) */
int sv_gm_encoder_prep(sv_gm_encoder* enc, const float* params, void* stream);
int sv_gm_encoder_forward(sv_gm_encoder* enc, const sv_gm_args* a, void* stream);
int sv_gm_encoder_backward(sv_gm_encoder* enc, const sv_gm_args* a, void* stream);
/* evaluation: the per-image categorical KL term "ykl" of the last forward, no gradients */
int sv_gm_encoder_y_kl(sv_gm_encoder* enc, void* stream);

/* ---------------------------------------------------------------- A2/A3/A5/A8: the whole LGVae step
 * Replaces LGVae.call (vae/model.py:189-200) and train_step_lg_vae (vae/trainer.py:120-144)
 * with one native launch sequence on `stream` (captured into a hipGraph by the caller if wanted). */
typedef struct {
  int32_t B, H, W;                       /* per-device batch; H==W power of two, multiple of 8 */
  int32_t global_latent, local_latent;   /* vae/main.py:16-17 (multiples of 8) */
  int32_t dtype;                         /* sv_dtype of the contractions */
  float beta;                            /* vae/main.py:19 */
  int32_t external_global_encoder;       /* 1: SPLIT-GMVAE (LGGMVae, vae/model.py:221-246): encoder_x is the caller's
                                            (type 'gmvae', :48-79).  The plan then skips encoder_x in every phase: the
                                            caller stores z_x into columns [0, global_latent) of the `zcat` buffer between
                                            FWD_ENCODERS and FWD_DECODERS and reads dL/dz_x from columns [0, global_latent)
                                            of `gz_x` after BWD_DECODERS.  The encoder_x slots of the parameter table stay
                                            (unused, zero gradient) so the flat layout is the same in both modes. */
} sv_lgvae_desc;

typedef struct sv_lgvae_plan sv_lgvae_plan;

/* the 40 trainable variables, Keras creation order, flat fp32 buffer */
int64_t sv_lgvae_param_count(const sv_lgvae_desc* d);
int sv_lgvae_param_info(const sv_lgvae_desc* d, int32_t index, int64_t* offset, int32_t* ndim,
                        int64_t shape[4], char name[96]);

int sv_lgvae_plan_create(const sv_lgvae_desc* d, sv_lgvae_plan** plan);
void sv_lgvae_plan_destroy(sv_lgvae_plan* plan);
int64_t sv_lgvae_workspace_bytes(const sv_lgvae_plan* plan);
/* carve the caller-owned workspace: ZERO-FILLS it (pad channels / rows that no kernel writes are read as zeros; a few accumulators count up from zero) and uploads the
 * (tiny) weight-prep job table, both on `stream`; returns after the upload has finished.  The workspace must not be written by anything else while the plan is bound. */
int sv_lgvae_plan_bind(sv_lgvae_plan* plan, void* workspace, int64_t bytes, void* stream);
/* byte offset/size of a named workspace buffer (for zero-copy views of outputs); <0 if unknown */
int sv_lgvae_buffer(const sv_lgvae_plan* plan, const char* name, int64_t* offset, int64_t* bytes);

enum { SV_PHASE_PREP = 1,           /* fp32 master weights -> MFMA-ready images */
       SV_PHASE_FWD_ENCODERS = 2,   /* split/pad, e1-e3, heads, reparameterisation + KL */
       SV_PHASE_FWD_DECODERS = 4,   /* d1-d5 of both decoders from the latents in `zcat` */
       SV_PHASE_LOSS = 8,           /* ELBO terms (+ their gradients and grad zeroing when grads != NULL) */
       SV_PHASE_BWD_DECODERS = 16,  /* fills the decoder_x / decoder_x_hat gradient ranges */
       SV_PHASE_BWD_ENC_HEADS = 32, /* reparam adjoint, e4_mean/e4_sd gradients, dgrad into a3 */
       SV_PHASE_BWD_ENC_CONVS = 64, /* e3, e2, e1 gradients */
       SV_PHASE_ADAM = 128,
       SV_PHASE_NO_RECON = 256,     /* modifier: a call that runs the decoders, the loss and its gradients together evaluates the loss in
                                       the head conv's epilogue; the reconstruction tensors out6_x / out6_xh (x_mean | x_log_scale) are
                                       then dead -- train_step_lg_vae (vae/trainer.py:121-144) returns nothing -- and with this bit they
                                       are not stored (100 MB of HBM writes per 512-image step).  Losses and gradients are unchanged. */
       SV_PHASE_INPUTS_STAGED = 512, /* modifier: the plan buffers in8_x / in8_xh already hold this call's images6 as padded 8-channel
                                       tensors in the plan's dtype (written by sv_scramble_gather_staged): the split / pad pass is skipped */
       SV_PHASE_BUCKET_EVENTS = 1024, /* modifier (data parallelism): the plan records an event set when each gradient bucket is complete -- 0: both decoders
                                       (after SV_PHASE_BWD_DECODERS), 1: the encoder heads (e4_mean / e4_sd, after SV_PHASE_BWD_ENC_HEADS), 2: the encoder convs --
                                       on the compute stream AND on its weight-gradient side streams, so ONE call can run the whole backward with the
                                       single-GPU stream overlap while the caller's communication stream picks every bucket up as it completes
                                       (sv_lgvae_bucket_wait); no phase-split host round trips */
       SV_PHASE_FORWARD = 6, SV_PHASE_BACKWARD = 112, SV_PHASE_ALL = 255 };

typedef struct {
  float* params;            /* flat fp32 [param_count] */
  float* grads;             /* flat fp32 [param_count] (zeroed by PHASE_LOSS) */
  float* adam_m;            /* flat fp32 */
  float* adam_v;            /* flat fp32 */
  const float* images6;     /* [B,H,W,6] fp32, output of sv_scramble_gather */
  const float* eps_x;       /* [B,global_latent] or NULL -> Philox */
  const float* eps_x_hat;   /* [B,local_latent]  or NULL -> Philox */
  uint64_t seed, step;
  int64_t sample_offset;    /* global index of this rank's first sample */
  float lr, beta1, beta2, adam_eps;
  int64_t t;                /* Adam iteration (iterations+1) */
  float grad_scale;         /* applied in Adam (1/world_size for DP) */
  int32_t phases;           /* SV_PHASE_* mask */
  int32_t accumulate_metrics; /* add the 5 scalars into the running-mean accumulators (K15) */
} sv_lgvae_step_args;

int sv_lgvae_step(sv_lgvae_plan* plan, const sv_lgvae_step_args* a, void* stream);

/* hipGraph replay of sv_lgvae_step (no reference counterpart: the reference's tf.function traces its step once,
 * vae/trainer.py:117 / :146; this is the HIP equivalent for the launch-bound small-batch configurations).  With it on,
 * the first step of each distinct (phase mask, buffer set, Adam constants) runs eagerly, the second is captured on
 * `stream`, later ones are ONE hipGraphLaunch; seed / step / sample_offset / lr / t stay per-step values (they reach
 * the kernels through a small device record).  Steps issued on the legacy default stream (stream == NULL) are never
 * captured; at most 16 distinct graphs are kept (callers that rotate buffers beyond that stay eager).  Disabling destroys
 * the captured graphs.  Profiling turns replay off.  Measured (scripts/bench_graph.py, SVHN-32 B=64 .. CelebA-64
 * B=512): replay == eager to 1 %: the kernels already run back to back, the small configurations are bound by the
 * ~10 us dependent-kernel latency of an in-order stream, which a graph of the same chain keeps. */
int sv_lgvae_graph_enable(sv_lgvae_plan* plan, int32_t enable);
/* Make `stream` wait (hipStreamWaitEvent, no host sync) until gradient bucket `bucket` of the most recent sv_lgvae_step call that carried
 * SV_PHASE_BUCKET_EVENTS is complete: 0 decoders, 1 encoder heads, 2 encoder convs, 3 = 1 and 2 (the whole encoders' range).  The gradient
 * all-reduce of that bucket (RCCL over xGMI) is then enqueued on `stream`.  SV_E_STATE: no such events were recorded.
 * Every call that runs a backward phase invalidates the events of the steps before it; a captured (hipGraph) step records none -- callers then order the
 * collective behind the compute stream instead (split_vae_amd/trainer.py falls back to one all-reduce after the backward).  Weight-gradient slab reduces
 * are never deferred past the events of a call that records them (SV_DEFER_REDUCE is ignored for that call). */
int sv_lgvae_bucket_wait(sv_lgvae_plan* plan, int32_t bucket, void* stream);
/* Test hooks of ONE plan (tests/test_gpu_dist.py); an explicit call, never an environment variable, so nothing a training job inherits can switch them on:
 *   "side_delay_us" = n    the first weight-gradient side-stream launch of every step is held back n microseconds (a consumer that misses the side
 *                          stream's part of a bucket then reads gradients that do not exist yet);
 *   "bucket_skip_side" = 1 sv_lgvae_bucket_wait drops the side streams' events (the negative control of the dependency test).
 * SV_E_BADARG: unknown key / value out of range. */
int sv_lgvae_plan_debug(sv_lgvae_plan* plan, const char* key, int64_t value);
/* The library's side streams: hipStream_t `index` (0 .. 2) of the CURRENT device, created once per process and shared by every plan (weight-gradient
 * streams), tape (lanes) and by the SPLIT-GMVAE step's second encoder stream -- a stream per object put a later object's stream on the hardware queue of the
 * compute stream it should run beside (csrc/streams.hip).  A host binding that wants a library-compatible side stream of its own takes it from here. */
int sv_side_stream(int32_t index, void** stream);
/* number of captured (instantiated) step graphs, or a negative SV_E_* */
int sv_lgvae_graph_count(const sv_lgvae_plan* plan);

/* per-kernel hipEvent timing (bench roofline): enable, run steps, read average ms per launch */
int sv_lgvae_profile_enable(sv_lgvae_plan* plan, int32_t enable);
/* restrict the event brackets to launches whose label equals `name` (NULL or "" = all launches) */
int sv_lgvae_profile_filter(sv_lgvae_plan* plan, const char* name);
int sv_lgvae_profile_read(sv_lgvae_plan* plan, int32_t max_entries, char names[][64],
                          double* total_ms, int32_t* launches, double* flops_per_launch,
                          double* bytes_per_launch);
/* FLOPs per launch the scope's algorithm really multiplies on the matrix pipe, entry i = entry i of sv_lgvae_profile_read.  flops_per_launch above is the DIRECT
 * form's count (2 B OH OW Cout KH KW Cin: SURVEY 8d); the polyphase forms of the upsample -> conv layers multiply 81 (per-class forms, the
 * stride-2 input gradient) or 100 (the head's merged 25-tap form) of the direct form's 144 tap products per low-res pixel, plus their border terms: a scope's
 * rate against the MFMA peak is a utilisation figure only at THIS count.  Equal to flops_per_launch for every direct-form scope. */
int sv_lgvae_profile_read_issued(sv_lgvae_plan* plan, int32_t max_entries, double* issued_flops_per_launch);

/* ---------------------------------------------------------------- SPLIT-SPAIR: Dense layers, exact fp32 on the matrix cores
 * tf.keras.layers.Dense (spair/spair.py:135-154, :185-202, :246-273, :341-366, :424-467) for ANY fan-in / fan-out, reading the Keras
 * [in, out] kernel as it lies in the variable buffer (dense_f32.hip).  x [M, ldx], y / dy [M, ldy], w [K, N] row-major.
 *   fwd:   y = act(x . w + bias)                       act: SV_ACT_NONE | SV_ACT_RELU
 *   dgrad: dx = dy . w^T (accumulate != 0: added to dx with atomics, K may be split over workgroups)
 *   wgrad: dw += x^T . dy, dbias += column sums of dy (atomics onto the zeroed gradients; dbias may be NULL) */
int sv_dense_f32_fwd(const float* x, int32_t ldx, const float* w, const float* bias, float* y, int32_t ldy, int32_t M, int32_t K,
                     int32_t N, int32_t act, void* stream);
int sv_dense_f32_dgrad(const float* dy, int32_t ldy, const float* w, float* dx, int32_t ldx, int32_t M, int32_t K, int32_t N,
                       int32_t accumulate, void* stream);
int sv_dense_f32_wgrad(const float* x, int32_t ldx, const float* dy, int32_t ldy, float* dw, float* dbias, int32_t M, int32_t K,
                       int32_t N, void* stream);

/* ---------------------------------------------------------------- SPLIT-SPAIR: the train step as one native launch sequence
 * Replaces, for spair/: SPAIR.call / LGSPAIR.call (spair/spair.py:35-49, :84-106), the loss assembly and tape.gradient of train_step
 * (spair/trainer.py:136-228) and optimizer.apply_gradients (:226-227, spair/main.py:109).  The host records the model ONCE as a list
 * of nodes over fp32 2-D tensors [rows, cols | row pitch ld floats] (split_vae_amd/spair_native.py mirrors the reference's classes);
 * sv_tape_run walks it forwards and backwards natively: no Python, autograd engine or library GEMM between the launches (tape.hip).
 * Tensor ids index the tape's tensors; *_off are float offsets into the flat variable / gradient buffers (-1: none). */
typedef struct sv_tape sv_tape;
enum { SV_TAPE_DENSE = 0, SV_TAPE_CONV, SV_TAPE_UNARY, SV_TAPE_SAMPLE, SV_TAPE_LOGITNOISE, SV_TAPE_UPSAMPLE, SV_TAPE_STN, SV_TAPE_RENDER,
       SV_TAPE_ZPRES, SV_TAPE_LOSS, SV_TAPE_NOISE };
enum { SV_TAPE_COPY = 0, SV_TAPE_RELU, SV_TAPE_SIGMOID, SV_TAPE_SOFTPLUS /* softplus(x + p0) */, SV_TAPE_CLAMP /* [p0, p1] */,
       SV_TAPE_SCALE /* p0 * x */ };
enum { SV_TAPE_PHASE_FORWARD = 1, SV_TAPE_PHASE_BACKWARD = 2, SV_TAPE_PHASE_ADAM = 4 };
typedef struct {
  int32_t kind;                 /* SV_TAPE_* node kind */
  int32_t x, y, t2, t3, t4, t5, t6;   /* tensors: main input, output, extra inputs (-1: none); per kind:
      DENSE       y[M,N] = act(x[M,K] . W + b)                         (Dense; also the backbone's 1x1 convolutions)
      CONV        y = act(conv2d_same(x, W) + b); B,H,W,C = input extent, Cout, k, stride       (spair/spair.py Conv2D layers)
      UNARY       y[:, yo:yo+n] = op(x[:, xo:xo+n]), `rep` output rows per input row (tf.tile / concat / slice / activations)
      SAMPLE      y[:, yo:] = x[:, xo:] + t2[:, o2:] * t3[:, o3:]      (Sampling, spair/utils.py:19-24: mean, sig, eps)
      LOGITNOISE  y = (x + log(t2 + 1e-8) - log(1 - t2 + 1e-8)) / p0   (concrete_binary_pre_sigmoid_sample, spair/utils.py:14-17)
      UPSAMPLE    y = tf.image.resize(x, 2x) on [B,H,W,C=ld]           (spair/spair.py:175-180, :360-364)
      STN         y (, t3 = obj_bbox_mask) = STN(x = images, t2 = z_where [B*Hc*Wc,4]); B,H,W,C input, Ho,Wo output, Hc,Wc cells, inverse
      RENDER      y = Renderer(x = objects, t2 = bg, t3 = z_depth, t4 = z_pres, t5 = z_pres_logits, t6 = GaussianNoise draw (-1)); R cells
      ZPRES       loss[loss_idx] = compute_z_pres_kl_yolo_air(x = z_pres, t2 = logits, t3 = pre_sigmoid); prior_prob = dyn[dyn_idx], p0 = tau
      LOSS        loss[loss_idx] = per-image sums of mode 0 xent(x = label, t2 = pred) | 1 kl(x = mean, t2 = sig) | 2 kl vs N(dyn[dyn_idx] | p0, p1)
                  over rows [b*R, (b+1)*R) x columns [xo | o2, +n)
      NOISE       y <- Philox draws, op 0: normal * p0, 1: uniform (skipped when the run pins the noise tensors) */
  int32_t xo, yo, o2, o3;       /* column offsets */
  int32_t n, rep, op, act;
  float p0, p1;
  int64_t w_off, b_off;
  int32_t B, H, W, C, Cout, k, stride, Ho, Wo, Hc, Wc, inverse, training;
  int32_t loss_idx, dyn_idx, mode, R, stream_id;
  int32_t group;                /* UNARY: consecutive nodes with the same non-zero group are independent of each other and run as ONE launch */
  int32_t lane;                 /* 0: the caller's stream; 1..3: a HIP stream of the tape's own.  Independent branches of the model (LG-SPAIR's x-hat / background
                                   networks beside the object pipeline, spair/spair.py:84-104) recorded on another lane run concurrently with lane 0; sv_tape_finalize
                                   derives every cross-lane dependency from the nodes' tensors and sv_tape_run orders conflicting accesses in tape order with events,
                                   so any lane assignment computes the single-stream step bit for bit (SV_TAPE_LANES=0: everything on the caller's stream) */
} sv_tape_node;
typedef struct {
  float* params;                /* flat fp32 variables (updated by the ADAM phase) */
  float* grads;                 /* flat fp32 gradients, same layout (zeroed and filled by the BACKWARD phase) */
  const float* loss_weights;    /* HOST array [n_weights]: total = sum_i loss_weights[i] * mean_b loss_i (spair/trainer.py:165-207) */
  int32_t n_weights;
  float dyn[8];                 /* step-dependent scalars the nodes reference (prior_z_pres_prob, the zoom prior's mean: :153, :156) */
  uint64_t seed, step;          /* Philox key of the NOISE nodes */
  int32_t pinned_noise;         /* != 0: the caller filled the noise tensors (tests) */
  int32_t phases;               /* SV_TAPE_PHASE_* */
  int32_t accumulate_metrics;
  float* adam_m; float* adam_v; int64_t n_params;
  float lr, beta1, beta2, adam_eps; int64_t t;
  float clipnorm;               /* > 0: tf.clip_by_norm per variable before the update (TF >= 2.4 apply_gradients); 0: plain Keras Adam */
  const int64_t* tensor_off; int32_t n_tensors; float* norm_ws;   /* clipnorm only: device [n_tensors + 1] offsets, [n_tensors * 256] floats */
} sv_tape_run_args;
int sv_tape_create(sv_tape** out, int32_t batch, int32_t conv_dtype);
void sv_tape_destroy(sv_tape* t);
int32_t sv_tape_tensor(sv_tape* t, int64_t rows, int32_t cols, int32_t ld, int32_t need_grad);       /* -> tensor id (>= 0) or SV_E_* */
int32_t sv_tape_view(sv_tape* t, int32_t src, int64_t rows, int32_t cols, int32_t ld);               /* same storage, other 2-D shape */
int sv_tape_add(sv_tape* t, const sv_tape_node* node);
/* The cross-lane schedule of a finalized tape, node by node (pass 0 forward, 1 backward): returns how many nodes' events the launch of `node` waits for (their indices in
 * waits[0 .. max_waits)), *records = 1 when an event is recorded behind it for another lane.  A UNARY group is one launch (waits on its first node forwards / its last
 * backwards).  Host logic only -- no GPU needed: what tests/test_abi.py pins.  0 for a single-lane tape. */
int sv_tape_schedule(const sv_tape* t, int32_t pass, int32_t node, int32_t* waits, int32_t max_waits, int32_t* records);
/* reported[j] = sum_i matrix[j * 16 + i] * mean_b loss_i: the `losses` list of train_step (spair/trainer.py:158-160, :208-216) */
int sv_tape_set_report(sv_tape* t, const float* matrix, int32_t n_report);
int sv_tape_finalize(sv_tape* t);
int64_t sv_tape_workspace_bytes(const sv_tape* t);
int sv_tape_bind(sv_tape* t, void* workspace, int64_t bytes, void* stream);                           /* zero-fills the workspace */
int sv_tape_tensor_info(const sv_tape* t, int32_t id, int64_t* offset, int64_t* grad_offset);        /* byte offsets (-1: no gradient) */
/* out block (floats): [total, reported x 16, mean loss_i x 16]; metric block: running sums of [total, reported x 16], then the count */
int sv_tape_loss_info(const sv_tape* t, int64_t* out_offset, int64_t* metric_offset, int32_t* n_loss);
int sv_tape_run(sv_tape* t, const sv_tape_run_args* args, void* stream);

/* ---- K16 data-parallel gradient exchange: absent in the reference (single device) -----------
 * One process per GPU; gradients of the batch-mean loss (vae/trainer.py:13,:127-128,:137) are averaged over equal
 * shards by an in-place all-reduce(sum) of the flat fp32 gradient buffer; 1/world is sv_adam_step's grad_scale.
 * RCCL is loaded at run time (the process's own copy if it has one, e.g. PyTorch's): SV_E_UNSUPPORTED without it.
 *   sv_comm_unique_id   rank 0 draws the 128-byte rendezvous id and ships it to the others out of band
 *   sv_comm_init        collective over all ranks; binds the calling thread's current HIP device
 *   sv_comm_allreduce   enqueue on `stream` (asynchronous): buf[0..count) <- sum over ranks
 *   sv_comm_allreduce_ranges   n disjoint element ranges [begin[i], end[i]) of one buffer as one RCCL group
 *                       (a bucket = the contiguous runs of the parameters whose gradients one backward phase completes) */
typedef struct sv_comm sv_comm;
int sv_comm_unique_id(void* id128);
int sv_comm_init(const void* id128, int32_t rank, int32_t world, sv_comm** out);
int sv_comm_allreduce(sv_comm* comm, float* buf, int64_t count, void* stream);
int sv_comm_allreduce_ranges(sv_comm* comm, float* base, const int64_t* begin, const int64_t* end, int32_t n, void* stream);
int sv_comm_destroy(sv_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
