// Native launch sequence of the SPLIT-GMVAE global encoder: Encoder(type='gmvae') (vae/model.py:48-79), its call
// (call_gmvae :116-135) and the adjoint that tape.gradient builds for train_step_lg_gm_vae (vae/trainer.py:146-173).
//
// The contractions are the sv_conv2d_* entry points (three stride-2 ELU convs, nine Dense layers = 1x1 convs on a 1x1
// grid), the glue is gm_pointwise.hip; this file only sequences them over one caller-owned workspace, the way
// lgvae_plan.hip does for the rest of the model (sv_lgvae_desc.external_global_encoder).  It replaces a per-layer Python
// loop (~110 ctypes calls per step: the step was host-bound at 3.2 ms for any batch size).
//
// Variables, in the reference's layer-tracking order (kernel, bias each): h_block conv2d x3, y_block dense x2, y_dense,
// h_top_dense, z_prior_mean, z_prior_sig, e1, z_mean, z_sig.  The Dropout layers do1-4, do6, do7 exist in the reference
// but are never called (:59-75 vs :116-135); only y_block's Dropout and do5 act, in training.
#include <map>
#include <string>
#include <vector>
#include <stdio.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"
#include "conv_geom.h"

#define SV_TRY(x)            \
  do {                       \
    const int rc_ = (x);     \
    if (rc_ != SV_OK) return rc_; \
  } while (0)

namespace {

constexpr float GM_RATE = 0.2f;   // Dropout(rate=0.2): y_block (vae/model.py:56) and do5 (:72)
enum { C1, C2, C3, D1, D2, YD, HT, PM, PS, E1, ZM, ZS, NLAYER };
const char* const LAYER_NAMES[NLAYER] = {
    "encoder_x/h_block/conv2d", "encoder_x/h_block/conv2d_1", "encoder_x/h_block/conv2d_2", "encoder_x/y_block/dense",
    "encoder_x/y_block/dense_1", "encoder_x/y_dense", "encoder_x/h_top_dense", "encoder_x/z_prior_mean",
    "encoder_x/z_prior_sig", "encoder_x/e1", "encoder_x/z_mean", "encoder_x/z_sig"};

struct GmParam { std::string name; int64_t off; int ndim; int64_t shape[4]; int64_t count; };
struct GmBuf { int64_t off, bytes; };

inline int r8(int v) { return (v + 7) / 8 * 8; }

int check_desc(const sv_gm_desc* d) {
  if (!d) return SV_E_BADARG;
  if (d->B <= 0 || d->H < 8 || d->H != d->W || ilog2_exact(d->H) < 0) return SV_E_UNSUPPORTED;
  if (d->latent < 8 || ilog2_exact(d->latent) < 0) return SV_E_UNSUPPORTED;
  if (d->y_size < 2 || d->y_size > 128) return SV_E_UNSUPPORTED;
  if (d->dtype != SV_BF16 && d->dtype != SV_F32) return SV_E_BADARG;
  if (!(d->tau > 0.f)) return SV_E_BADARG;
  return SV_OK;
}

std::vector<GmParam> build_params(const sv_gm_desc* d) {
  const int64_t F = (int64_t)(d->H / 8) * (d->W / 8) * 128, K = d->y_size, L = d->latent;
  const int64_t kshape[NLAYER][4] = {{6, 6, 3, 128}, {6, 6, 128, 128}, {4, 4, 128, 128}, {F, 1024, 0, 0}, {1024, 128, 0, 0},
                                     {128, K, 0, 0}, {K, 512, 0, 0}, {K, L, 0, 0}, {K, L, 0, 0}, {F, 512, 0, 0},
                                     {512, L, 0, 0}, {512, L, 0, 0}};
  std::vector<GmParam> v;
  int64_t off = 0;
  for (int l = 0; l < NLAYER; ++l) {
    GmParam k;
    k.name = std::string(LAYER_NAMES[l]) + "/kernel"; k.off = off; k.ndim = l < 3 ? 4 : 2; k.count = 1;
    for (int i = 0; i < 4; ++i) { k.shape[i] = i < k.ndim ? kshape[l][i] : 1; k.count *= k.shape[i]; }
    off += (k.count + 3) / 4 * 4;
    v.push_back(k);
    GmParam b;
    b.name = std::string(LAYER_NAMES[l]) + "/bias"; b.off = off; b.ndim = 1;
    b.shape[0] = kshape[l][k.ndim - 1]; b.shape[1] = b.shape[2] = b.shape[3] = 1; b.count = b.shape[0];
    off += (b.count + 3) / 4 * 4;
    v.push_back(b);
  }
  return v;
}

}  // namespace

struct sv_gm_encoder {
  sv_gm_desc d;
  std::vector<GmParam> params;
  sv_conv_desc conv[NLAYER];
  int64_t wf_off[NLAYER], wd_off[NLAYER];   // prepared forward / input-gradient images (element offsets in the arena)
  std::vector<PrepJob> jobs;
  int prep_blocks;
  std::map<std::string, GmBuf> bufs;
  int64_t ws_bytes = 0;
  char* ws = nullptr;
  float rate = 0.f;                          // dropout rate of the last forward (its backward uses the same)
  int64_t F;
  int Kp;
  hipEvent_t ev_dy[NLAYER] = {};             // "dY of layer l is ready": its weight gradient runs on the library's shared side stream 0 behind this
  hipEvent_t ev_wjoin = nullptr;
  ~sv_gm_encoder() {
    for (auto ev : ev_dy)
      if (ev) (void)hipEventDestroy(ev);
    if (ev_wjoin) (void)hipEventDestroy(ev_wjoin);
  }

  size_t esz() const { return d.dtype == SV_BF16 ? 2 : 4; }
  void add(const std::string& n, int64_t bytes) {
    bufs[n] = GmBuf{ws_bytes, bytes};
    ws_bytes += (bytes + 255) / 256 * 256;
  }
  char* bp(const char* n) const { return ws + bufs.at(n).off; }
  float* fp(const char* n) const { return (float*)bp(n); }
  const float* kernel(const float* flat, int l) const { return flat + params[2 * l].off; }
  const float* bias(const float* flat, int l) const { return flat + params[2 * l + 1].off; }
  void* wfwd(int l) const { return bp("warena") + wf_off[l] * esz(); }
  void* wdgrad(int l) const { return bp("warena") + wd_off[l] * esz(); }
};

extern "C" int64_t sv_gm_param_count(const sv_gm_desc* d) {
  if (check_desc(d) != SV_OK) return -1;
  auto v = build_params(d);
  return v.back().off + (v.back().count + 3) / 4 * 4;
}

extern "C" int sv_gm_param_info(const sv_gm_desc* d, int32_t index, int64_t* offset, int32_t* ndim, int64_t shape[4],
                                char name[96]) {
  const int rc = check_desc(d);
  if (rc) return rc;
  auto v = build_params(d);
  if (index < 0 || index >= (int)v.size()) return SV_E_BADARG;
  if (offset) *offset = v[index].off;
  if (ndim) *ndim = v[index].ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = v[index].shape[i];
  if (name) snprintf(name, 96, "%s", v[index].name.c_str());
  return SV_OK;
}

extern "C" int sv_gm_encoder_create(const sv_gm_desc* d, sv_gm_encoder** out) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!out) return SV_E_BADARG;
  sv_gm_encoder* e = new sv_gm_encoder();
  e->d = *d;
  e->params = build_params(d);
  const int B = d->B, H = d->H, K = d->y_size, L = d->latent;
  const int64_t F = (int64_t)(H / 8) * (H / 8) * 128;
  e->F = F;
  e->Kp = r8(K);
  auto conv = [&](int h, int cin, int cout, int k) {
    sv_conv_desc c;
    memset(&c, 0, sizeof(c));
    c.B = B; c.H = h; c.W = h; c.Cin = cin; c.Cout = cout; c.KH = k; c.KW = k; c.stride = 2; c.act = SV_ACT_NONE;
    c.dtype = d->dtype; c.ldx = r8(cin); c.ldy = r8(cout);
    return c;
  };
  auto dense = [&](int64_t cin, int cout) {   // Dense = 1x1 conv on a 1x1 grid, fp32 pre-activation out (bias included)
    sv_conv_desc c;
    memset(&c, 0, sizeof(c));
    c.B = B; c.H = 1; c.W = 1; c.Cin = (int)cin; c.Cout = cout; c.KH = 1; c.KW = 1; c.stride = 1; c.act = SV_ACT_NONE;
    c.dtype = d->dtype; c.ldx = r8((int)cin); c.ldy = cout; c.y_f32 = 1;
    return c;
  };
  e->conv[C1] = conv(H, 3, 128, 6); e->conv[C2] = conv(H / 2, 128, 128, 6); e->conv[C3] = conv(H / 4, 128, 128, 4);
  e->conv[D1] = dense(F, 1024); e->conv[D2] = dense(1024, 128); e->conv[YD] = dense(128, K); e->conv[HT] = dense(K, 512);
  e->conv[PM] = dense(K, L); e->conv[PS] = dense(K, L); e->conv[E1] = dense(F, 512); e->conv[ZM] = dense(512, L);
  e->conv[ZS] = dense(512, L);
  for (int l = 0; l < NLAYER; ++l)
    if ((rc = svg_check(&e->conv[l])) != SV_OK) { delete e; return rc; }

  // weight-preparation jobs: one launch re-lays all 24 images per step (layout of sv_conv2d_prep_weights: the classes of
  // an input-gradient image are contiguous)
  int64_t arena = 0;
  int blocks = 0;
  auto push = [&](PrepJob j) { j.first_block = blocks; blocks += j.nblocks; e->jobs.push_back(j); };
  for (int l = 0; l < NLAYER; ++l) {
    PrepJob j;
    svg_prep_job_fwd(&e->conv[l], &j);
    j.src_off = e->params[2 * l].off;
    arena = (arena + 127) / 128 * 128;
    e->wf_off[l] = arena; j.dst_off = arena;
    arena += (int64_t)j.rows * j.ntaps * j.inner;
    push(j);
    arena = (arena + 127) / 128 * 128;
    e->wd_off[l] = arena;
    for (int c = 0; c < svg_dgrad_classes(&e->conv[l]); ++c) {
      PrepJob jd;
      svg_prep_job_dgrad(&e->conv[l], c, &jd);
      jd.src_off = e->params[2 * l].off;
      jd.dst_off = arena;
      arena += (int64_t)jd.rows * jd.ntaps * jd.inner;
      push(jd);
    }
  }
  e->prep_blocks = blocks;

  const int64_t es = (int64_t)e->esz(), Kp = e->Kp;
  const int64_t P1 = (int64_t)B * (H / 2) * (H / 2) * 128, P2 = (int64_t)B * (H / 4) * (H / 4) * 128, BF = (int64_t)B * F;
  e->add("jobs", (int64_t)e->jobs.size() * sizeof(PrepJob));
  e->add("warena", (arena + 128) * es);
  // forward (compute dtype unless noted)
  e->add("h1", P1 * es); e->add("h2", P2 * es); e->add("h3", BF * es);
  e->add("a1", (int64_t)B * 1024 * 4); e->add("yh1a", (int64_t)B * 1024 * es); e->add("yh1", (int64_t)B * 1024 * es);
  e->add("keep1", (int64_t)B * 1024 * 4);
  e->add("a2", (int64_t)B * 128 * 4); e->add("yh2", (int64_t)B * 128 * es);
  e->add("logits", (int64_t)B * K * 4); e->add("y", (int64_t)B * K * 4); e->add("y_lp", (int64_t)B * Kp * es); e->add("u", (int64_t)B * K * 4);
  e->add("a_pm", (int64_t)B * L * 4); e->add("a_ps", (int64_t)B * L * 4); e->add("a_t", (int64_t)B * 512 * 4);
  e->add("h_top", (int64_t)B * 512 * es);
  e->add("h5", BF * es); e->add("keep5", BF * 4); e->add("a_e", (int64_t)B * 512 * 4); e->add("he", (int64_t)B * 512 * es);
  e->add("hh", (int64_t)B * 512 * es);
  e->add("a_m", (int64_t)B * L * 4); e->add("a_s", (int64_t)B * L * 4);
  for (const char* n : {"zm", "zs", "z", "pm", "ps", "eps"}) e->add(n, (int64_t)B * L * 4);
  e->add("kl2", (int64_t)B * 4); e->add("ykl", (int64_t)B * 4);
  // backward
  for (const char* n : {"g_am", "g_as", "g_apm", "g_aps"}) e->add(n, (int64_t)B * L * es);
  e->add("g_ae", (int64_t)B * 512 * es); e->add("g_at", (int64_t)B * 512 * es);
  e->add("g_logits", (int64_t)B * Kp * es); e->add("g_a2", (int64_t)B * 128 * es); e->add("g_a1", (int64_t)B * 1024 * es);
  e->add("g_c3", BF * es); e->add("g_h2", P2 * es); e->add("g_c2", P2 * es); e->add("g_h1d", P1 * es); e->add("g_c1", P1 * es);
  // fp32 accumulation targets of the split-K input gradients: adjacent, zeroed with one memset per backward
  e->add("g_hh", (int64_t)B * 512 * 4); e->add("g_h5", BF * 4); e->add("g_y", (int64_t)B * Kp * 4);
  e->add("g_yh2", (int64_t)B * 128 * 4); e->add("g_yh1", (int64_t)B * 1024 * 4); e->add("g_h1", BF * 4);
  e->add("acc_end", 0);
  // partial-sum slabs of the tile weight gradients (two-stage flush: faster than fp32 atomics and run-to-run identical)
  int64_t wsb = 0;
  for (int l = 0; l < NLAYER; ++l) { const int64_t b = sv_conv2d_wgrad_workspace_bytes(&e->conv[l]); wsb = b > wsb ? b : wsb; }
  e->add("wgrad_ws", wsb);
  *out = e;
  return SV_OK;
}

extern "C" void sv_gm_encoder_destroy(sv_gm_encoder* e) { delete e; }

extern "C" int64_t sv_gm_encoder_workspace_bytes(const sv_gm_encoder* e) { return e ? e->ws_bytes : -1; }

extern "C" int sv_gm_encoder_bind(sv_gm_encoder* e, void* workspace, int64_t bytes, void* stream) {
  if (!e || !workspace) return SV_E_BADARG;
  if (bytes < e->ws_bytes) return SV_E_WORKSPACE;
  if ((uintptr_t)workspace & 255) return SV_E_BADARG;
  e->ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, (size_t)e->ws_bytes, st) != hipSuccess) return (int)hipGetLastError();     // pad channels and accumulation targets start from zero (as sv_lgvae_plan_bind)
  if (hipMemcpyAsync(e->bp("jobs"), e->jobs.data(), e->jobs.size() * sizeof(PrepJob), hipMemcpyHostToDevice, st) != hipSuccess)
    return (int)hipGetLastError();
  if (hipStreamSynchronize(st) != hipSuccess) return (int)hipGetLastError();   // the host vector may go away
  return SV_OK;
}

extern "C" int sv_gm_encoder_buffer(const sv_gm_encoder* e, const char* name, int64_t* offset, int64_t* bytes) {
  if (!e || !name) return SV_E_BADARG;
  auto it = e->bufs.find(name);
  if (it == e->bufs.end()) return SV_E_BADARG;
  if (offset) *offset = it->second.off;
  if (bytes) *bytes = it->second.bytes;
  return SV_OK;
}

// fp32 HWIO masters -> MFMA-ready images of all twelve layers (one launch)
extern "C" int sv_gm_encoder_prep(sv_gm_encoder* e, const float* params, void* stream) {
  if (!e || !e->ws || !params) return SV_E_BADARG;
  return svk_prep_weights(params, e->bp("warena"), e->d.dtype, (const PrepJob*)e->bp("jobs"), (int)e->jobs.size(),
                          e->prep_blocks, (hipStream_t)stream);
}

// call_gmvae (vae/model.py:116-135).  Needs sv_gm_encoder_prep after every change of `params`.
extern "C" int sv_gm_encoder_forward(sv_gm_encoder* e, const sv_gm_args* a, void* stream) {
  if (!e || !e->ws || !a || !a->params || !a->in8_x || !a->zcat) return SV_E_BADARG;
  const sv_gm_desc& d = e->d;
  const int B = d.B, H = d.H, K = d.y_size, L = d.latent, dt = d.dtype, Kp = e->Kp;
  const int64_t F = e->F;
  const float* P = a->params;
  const float rate = a->training ? GM_RATE : 0.f;
  auto fwd = [&](int l, const void* x, void* y) { return sv_conv2d_nhwc_fwd(&e->conv[l], x, e->wfwd(l), e->bias(P, l), y, stream); };
  auto elu_inplace = [&](const char* n, int64_t rows) {   // Conv2D(activation='elu') :50-52
    return sv_act_fwd(e->bp(n), dt, 128, nullptr, e->bp(n), dt, 128, rows, 128, SV_ACT_ELU, 0.f, nullptr, nullptr, 0, 0, 0, 0, 1, stream);
  };
  // h_block: three stride-2 convs with ELU
  SV_TRY(fwd(C1, a->in8_x, e->bp("h1"))); SV_TRY(elu_inplace("h1", (int64_t)B * (H / 2) * (H / 2)));
  SV_TRY(fwd(C2, e->bp("h1"), e->bp("h2"))); SV_TRY(elu_inplace("h2", (int64_t)B * (H / 4) * (H / 4)));
  SV_TRY(fwd(C3, e->bp("h2"), e->bp("h3"))); SV_TRY(elu_inplace("h3", (int64_t)B * (H / 8) * (H / 8)));
  // y_block (:54-58) -> y_dense (:60) -> Gumbel-softmax (:121-122)
  SV_TRY(fwd(D1, e->bp("h3"), e->bp("a1")));
  SV_TRY(sv_act_fwd(e->bp("a1"), SV_F32, 1024, e->bp("yh1a"), e->bp("yh1"), dt, 1024, B, 1024, SV_ACT_ELU, rate, a->keep1,
                    e->fp("keep1"), a->seed, a->step, 11, a->sample_offset, 1, stream));
  SV_TRY(fwd(D2, e->bp("yh1"), e->bp("a2")));
  SV_TRY(sv_act_fwd(e->bp("a2"), SV_F32, 128, nullptr, e->bp("yh2"), dt, 128, B, 128, SV_ACT_ELU, 0.f, nullptr, nullptr, 0, 0, 0, 0, 1, stream));
  SV_TRY(fwd(YD, e->bp("yh2"), e->bp("logits")));
  SV_TRY(sv_gumbel_softmax_fwd(e->fp("logits"), K, a->u, e->fp("u"), d.tau, e->fp("y"), e->bp("y_lp"), dt, Kp, B, K, a->seed,
                               a->step, a->sample_offset, stream));
  // prior (:124-125), h_top (:127), encoder block (:128-133)
  SV_TRY(fwd(PM, e->bp("y_lp"), e->bp("a_pm")));
  SV_TRY(fwd(PS, e->bp("y_lp"), e->bp("a_ps")));
  SV_TRY(fwd(HT, e->bp("y_lp"), e->bp("a_t")));
  SV_TRY(sv_act_fwd(e->bp("a_t"), SV_F32, 512, nullptr, e->bp("h_top"), dt, 512, B, 512, SV_ACT_ELU, 0.f, nullptr, nullptr, 0, 0, 0, 0, 1, stream));
  SV_TRY(sv_act_fwd(e->bp("h3"), dt, (int)F, nullptr, e->bp("h5"), dt, (int)F, B, (int)F, SV_ACT_NONE, rate, a->keep5, e->fp("keep5"),
                    a->seed, a->step, 15, a->sample_offset, 1, stream));
  SV_TRY(fwd(E1, e->bp("h5"), e->bp("a_e")));
  SV_TRY(sv_act_fwd(e->bp("a_e"), SV_F32, 512, nullptr, e->bp("he"), dt, 512, B, 512, SV_ACT_ELU, 0.f, nullptr, nullptr, 0, 0, 0, 0, 1, stream));
  SV_TRY(sv_add(e->bp("he"), e->bp("h_top"), e->bp("hh"), dt, (int64_t)B * 512, stream));
  SV_TRY(fwd(ZM, e->bp("hh"), e->bp("a_m")));
  SV_TRY(fwd(ZS, e->bp("hh"), e->bp("a_s")));
  SV_TRY(sv_gm_head_fwd(e->fp("a_m"), e->fp("a_s"), e->fp("a_pm"), e->fp("a_ps"), a->eps, e->fp("eps"), e->fp("zm"), e->fp("zs"),
                        e->fp("z"), e->fp("pm"), e->fp("ps"), a->zcat, dt, a->ldz, 0, e->fp("kl2"), B, L, a->seed, a->step,
                        a->sample_offset, stream));
  e->rate = rate;
  return SV_OK;
}

// The adjoint (tape.gradient, vae/trainer.py:167): gz [B, >= L] fp32 = dL/dz_x from the decoder (columns [0, L)).
// ACCUMULATES the 24 gradients into `grads` (zero them first) and fills ykl (the per-image categorical KL term).
extern "C" int sv_gm_encoder_backward(sv_gm_encoder* e, const sv_gm_args* a, void* stream) {
  if (!e || !e->ws || !a || !a->params || !a->grads || !a->in8_x || !a->gz) return SV_E_BADARG;
  const sv_gm_desc& d = e->d;
  const int B = d.B, H = d.H, K = d.y_size, L = d.latent, dt = d.dtype, Kp = e->Kp;
  const int64_t F = e->F;
  const float rate = e->rate;
  hipStream_t st = (hipStream_t)stream;
  {
    char* z0 = e->bp("g_hh");
    char* z1 = e->bp("acc_end");
    if (hipMemsetAsync(z0, 0, (size_t)(z1 - z0), st) != hipSuccess) return (int)hipGetLastError();
  }
  // The weight gradients feed only Adam: they go to the library's shared side stream 0 behind a "dY is ready" event, the input-gradient chain (the critical
  // path: 12 layers, one after the other) continues at once; joined at the end of the call.  Every dY buffer is written once per call, the forward activations
  // are read-only here, each layer's variable gradients have one writer.  SV_GM_WGRAD_SIDE=0: everything on `stream` (A/B).  profiles/r06_gm_streams.txt
  static const bool wside = !(getenv("SV_GM_WGRAD_SIDE") && atoi(getenv("SV_GM_WGRAD_SIDE")) == 0);
  hipStream_t ws2 = wside ? sv_shared_stream(0) : nullptr;
  if (ws2 == st) ws2 = nullptr;
  bool forked = false;
  auto wg = [&](int l, const void* x, const void* dy) -> int {
    void* wst = stream;
    if (ws2) {
      if (!e->ev_dy[l] && hipEventCreateWithFlags(&e->ev_dy[l], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
      if (hipEventRecord(e->ev_dy[l], st) != hipSuccess || hipStreamWaitEvent(ws2, e->ev_dy[l], 0) != hipSuccess) return (int)hipGetLastError();
      wst = (void*)ws2;
      forked = true;
    }
    return sv_conv2d_nhwc_wgrad_ws(&e->conv[l], x, dy, a->grads + e->params[2 * l].off, a->grads + e->params[2 * l + 1].off, e->bp("wgrad_ws"),
                                   e->bufs.at("wgrad_ws").bytes, wst);
  };
  auto dg_acc = [&](int l, const void* dy, const char* acc) {   // split-K input gradient added into an fp32 buffer
    return sv_conv2d_nhwc_dgrad(&e->conv[l], dy, e->wdgrad(l), nullptr, e->bp(acc), 1, stream);
  };
  auto dg = [&](int l, const void* dy, const char* out) { return sv_conv2d_nhwc_dgrad(&e->conv[l], dy, e->wdgrad(l), nullptr, e->bp(out), 0, stream); };
  auto act_bwd = [&](const char* gx, int gx_dt, int ld, const char* gx2, const char* y_act, float r, const float* keep,
                     const char* ga, int64_t rows, int C) {
    return sv_act_bwd(e->bp(gx), gx_dt, ld, gx2 ? e->bp(gx2) : nullptr, gx2 ? SV_F32 : 0, gx2 ? ld : 0, e->bp(y_act), dt, ld,
                      SV_ACT_ELU, r, keep, e->bp(ga), dt, ld, rows, C, stream);
  };
  SV_TRY(sv_gm_head_bwd(a->gz, a->ld_gz, e->fp("zm"), e->fp("zs"), e->fp("pm"), e->fp("ps"), e->fp("eps"), a->beta / (float)B,
                        e->bp("g_am"), e->bp("g_as"), e->bp("g_apm"), e->bp("g_aps"), dt, B, L, stream));
  SV_TRY(wg(ZM, e->bp("hh"), e->bp("g_am"))); SV_TRY(wg(ZS, e->bp("hh"), e->bp("g_as")));
  SV_TRY(dg_acc(ZM, e->bp("g_am"), "g_hh")); SV_TRY(dg_acc(ZS, e->bp("g_as"), "g_hh"));
  SV_TRY(act_bwd("g_hh", SV_F32, 512, nullptr, "he", 0.f, nullptr, "g_ae", B, 512));
  SV_TRY(act_bwd("g_hh", SV_F32, 512, nullptr, "h_top", 0.f, nullptr, "g_at", B, 512));
  SV_TRY(wg(E1, e->bp("h5"), e->bp("g_ae")));
  SV_TRY(dg_acc(E1, e->bp("g_ae"), "g_h5"));
  SV_TRY(wg(HT, e->bp("y_lp"), e->bp("g_at"))); SV_TRY(dg_acc(HT, e->bp("g_at"), "g_y"));
  SV_TRY(wg(PM, e->bp("y_lp"), e->bp("g_apm"))); SV_TRY(dg_acc(PM, e->bp("g_apm"), "g_y"));
  SV_TRY(wg(PS, e->bp("y_lp"), e->bp("g_aps"))); SV_TRY(dg_acc(PS, e->bp("g_aps"), "g_y"));
  SV_TRY(sv_gumbel_softmax_bwd(e->fp("g_y"), Kp, e->fp("y"), e->fp("logits"), K, d.tau, a->alpha / (float)B, e->bp("g_logits"), dt,
                               Kp, e->fp("ykl"), B, K, stream));
  SV_TRY(wg(YD, e->bp("yh2"), e->bp("g_logits"))); SV_TRY(dg_acc(YD, e->bp("g_logits"), "g_yh2"));
  SV_TRY(act_bwd("g_yh2", SV_F32, 128, nullptr, "yh2", 0.f, nullptr, "g_a2", B, 128));
  SV_TRY(wg(D2, e->bp("yh1"), e->bp("g_a2"))); SV_TRY(dg_acc(D2, e->bp("g_a2"), "g_yh1"));
  SV_TRY(act_bwd("g_yh1", SV_F32, 1024, nullptr, "yh1a", rate, e->fp("keep1"), "g_a1", B, 1024));
  SV_TRY(wg(D1, e->bp("h3"), e->bp("g_a1"))); SV_TRY(dg_acc(D1, e->bp("g_a1"), "g_h1"));
  // h feeds y_block (g_h1) and, through do5, e1 (g_h5): combine, then ELU' of conv3
  SV_TRY(act_bwd("g_h5", SV_F32, (int)F, "g_h1", "h3", rate, e->fp("keep5"), "g_c3", B, (int)F));
  SV_TRY(wg(C3, e->bp("h2"), e->bp("g_c3")));
  SV_TRY(dg(C3, e->bp("g_c3"), "g_h2"));
  SV_TRY(act_bwd("g_h2", dt, 128, nullptr, "h2", 0.f, nullptr, "g_c2", (int64_t)B * (H / 4) * (H / 4), 128));
  SV_TRY(wg(C2, e->bp("h1"), e->bp("g_c2")));
  SV_TRY(dg(C2, e->bp("g_c2"), "g_h1d"));
  SV_TRY(act_bwd("g_h1d", dt, 128, nullptr, "h1", 0.f, nullptr, "g_c1", (int64_t)B * (H / 2) * (H / 2), 128));
  SV_TRY(wg(C1, a->in8_x, e->bp("g_c1")));
  if (forked) {
    if (!e->ev_wjoin && hipEventCreateWithFlags(&e->ev_wjoin, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    if (hipEventRecord(e->ev_wjoin, ws2) != hipSuccess || hipStreamWaitEvent(st, e->ev_wjoin, 0) != hipSuccess) return (int)hipGetLastError();
  }
  return SV_OK;
}

// evaluation: only the per-image categorical KL term ykl (no gradients)
extern "C" int sv_gm_encoder_y_kl(sv_gm_encoder* e, void* stream) {
  if (!e || !e->ws) return SV_E_BADARG;
  return sv_gumbel_softmax_bwd(nullptr, 0, e->fp("y"), e->fp("logits"), e->d.y_size, e->d.tau, 0.f, nullptr, 0, 0, e->fp("ykl"),
                               e->d.B, e->d.y_size, stream);
}
