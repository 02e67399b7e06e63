// The SPLIT-VAE (LGVae) training step as one native launch sequence.
//
// Replaces LGVae.call (vae/model.py:189-200) + train_step_lg_vae (vae/trainer.py:120-144) +
// Adam.apply_gradients (vae/main.py:65): weight preparation, both encoders, reparameterisation/KL,
// both decoders, discretised-logistic ELBO, the full backward pass and the Keras-Adam update are
// enqueued on one HIP stream from C++ (no Python per kernel), over a caller-owned workspace.
// The phase mask lets the data-parallel driver slip its RCCL all-reduce between the decoder
// backward, the encoder backward and Adam.
#include <string>
#include <algorithm>
#include <vector>
#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.hip.h"
#include "kernels.h"
#include "conv_geom.h"

namespace {

// test hook only (sv_lgvae_plan_debug "side_delay_us"): holds a stream back for a while
__global__ void sv_plan_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
static void sv_plan_spin(hipStream_t st, int us) { hipLaunchKernelGGL(sv_plan_spin_kernel, dim3(1), dim3(64), 0, st, (long long)us * 100); }   // wall_clock64: 100 MHz

struct Buf { std::string name; int64_t off, bytes; };

struct ParamInfo { std::string name; int64_t off; int ndim; int64_t shape[4]; int64_t count; };

static std::vector<ParamInfo> build_params(const sv_lgvae_desc* d) {
  std::vector<ParamInfo> v;
  int64_t off = 0;
  auto add = [&](const std::string& n, std::initializer_list<int64_t> shp) {
    ParamInfo p;
    p.name = n; p.off = off; p.ndim = (int)shp.size(); p.count = 1;
    int i = 0;
    for (auto s : shp) { p.shape[i++] = s; p.count *= s; }
    for (; i < 4; ++i) p.shape[i] = 1;
    off += (p.count + 3) / 4 * 4;   // keep every tensor 16-B aligned in the flat buffer
    v.push_back(p);
  };
  const int64_t F = (int64_t)(d->H / 8) * (d->W / 8) * 128;
  // vae/model.py:152 writes image_shape[1]//8*image_shape[2]//8*128 = ((H//8)*W)//8*128
  const int64_t D1 = ((int64_t)(d->H / 8) * d->W) / 8 * 128;
  auto enc = [&](const std::string& pre, int64_t L) {
    add(pre + "/e1/kernel", {6, 6, 3, 32}); add(pre + "/e1/bias", {32});
    add(pre + "/e2/kernel", {6, 6, 32, 64}); add(pre + "/e2/bias", {64});
    add(pre + "/e3/kernel", {4, 4, 64, 128}); add(pre + "/e3/bias", {128});
    add(pre + "/e4_mean/kernel", {F, L}); add(pre + "/e4_mean/bias", {L});
    add(pre + "/e4_sd/kernel", {F, L}); add(pre + "/e4_sd/bias", {L});
  };
  auto dec = [&](const std::string& pre, int64_t L) {
    add(pre + "/d1/kernel", {L, D1}); add(pre + "/d1/bias", {D1});
    add(pre + "/d2/kernel", {4, 4, 128, 128}); add(pre + "/d2/bias", {128});
    add(pre + "/d3/kernel", {4, 4, 128, 64}); add(pre + "/d3/bias", {64});
    add(pre + "/d4/kernel", {6, 6, 64, 32}); add(pre + "/d4/bias", {32});
    add(pre + "/d5/kernel", {6, 6, 32, 6}); add(pre + "/d5/bias", {6});
  };
  enc("encoder_x", d->global_latent);
  enc("encoder_x_hat", d->local_latent);
  dec("decoder_x", d->global_latent + d->local_latent);
  dec("decoder_x_hat", d->local_latent);
  return v;
}

static int check_desc(const sv_lgvae_desc* d) {
  if (!d) return SV_E_BADARG;
  if (d->B <= 0 || d->H < 8 || d->W < 8) return SV_E_BADARG;
  if (d->H != d->W || ilog2_exact(d->H) < 0) return SV_E_UNSUPPORTED;
  if (d->global_latent <= 0 || d->local_latent <= 0) return SV_E_BADARG;
  if (ilog2_exact(d->global_latent) < 0 || ilog2_exact(d->local_latent) < 0 || d->global_latent < 8 ||
      d->local_latent < 8 || d->global_latent != d->local_latent)
    return SV_E_UNSUPPORTED;   // MFMA K-pieces want power-of-two channel counts (reference default 128/128)
  if (d->dtype != SV_BF16 && d->dtype != SV_F32) return SV_E_BADARG;
  return SV_OK;
}

struct ProfEntry { std::string name; double flops, bytes, total_ms; int launches; double issued; };   // issued: see issued_flops() below
struct ProfPending { int entry; hipEvent_t a, b; };

// one conv-like layer instance with everything needed for fwd / dgrad / wgrad
struct Layer {
  std::string name;
  sv_conv_desc d;
  int kparam, bparam;           // indices into the param table (kernel, bias); head uses two kernels
  int64_t wf_off;               // prepared forward weights (element offset in arena)
  int64_t wd_off[4];            // prepared dgrad weights per parity class
  int64_t wdp_off;              // polyphase input gradient (svg_polyd): main / edge / corner images, or -1
  bool need_dgrad;
  bool polyc;                   // forward in per-class polyphase form (conv_geom.h: svg_polyc), LATCHED when the plan is created: the arena then holds the class images
                                // only, and a later change of the tuning environment must not send the direct kernel to them
};

}  // namespace

struct sv_lgvae_plan {
  sv_lgvae_desc d;
  std::vector<ParamInfo> params;
  int64_t nparams;
  std::vector<Buf> bufs;
  std::map<std::string, int> bufidx;
  int64_t ws_bytes;
  char* ws;
  bool bound;
  // layers: [enc_x e1,e2,e3,head][enc_xh ...][dec_x d1..d5][dec_xh ...]
  Layer enc[2][4];
  Layer dec[2][5];
  std::vector<PrepJob> jobs;
  int prep_blocks;
  int dec_block0 = 0;           // first block of the decoders' jobs in the table (the encoders' jobs come first): the step prepares the two halves on two streams
  int64_t arena_elems;
  // slab reduces of the tile weight gradients, deferred: every layer keeps its partial sums in its own workspace region and ONE launch
  // sums them all after the backward pass (after the side stream has joined).  Opt-in (SV_DEFER_REDUCE=1): see run_wgrad_layers for the measurement
  WgradReduceDesc red_pending[64];
  int n_pending = 0;
  bool nll_fused = false;  // the last decoder forward evaluated the loss in the head's epilogue (nllpart_*, g5_* are valid)
  bool gz_clean = false;   // dz accumulators zeroed by the last encoder-forward phase and not yet used
  // the K-slice slabs of d1's input gradient are still unsummed in lat_ws_x / lat_ws_xh: reparam_kl_bwd sums them itself (one launch less)
  bool dz_slabs = false;
  bool dz_valid = false;        // the decoders' backward has run since the last encoder forward (dz is what reparam_kl_bwd may read)
  bool gz_zero_skipped = false; // that forward did not zero dz (slab path): an encoder backward WITHOUT the decoders' (KL terms only) zeroes it first
  unsigned polycw_bad = 0;      // decoder layers (bit = index in dec[]) whose polyphase weight gradient was refused once: not tried again (one profile scope per launch)
  bool lat_head_ok = false, lat_d1_ok = false;     // the heads' forward / d1's input gradient of this plan run on latent_gemm.hip (shapes are fixed per plan)
  int dz_S[2] = {0, 0};
  int64_t dz_stride[2] = {0, 0};
  // weight gradients on a second stream (they feed only Adam / the all-reduce; the input-gradient chain is the critical
  // path): fork = the side stream waits for the event recorded on the main stream when dY is ready, join before Adam and
  // at the end of every sv_lgvae_step call
  enum { SIDE_MAX = 4 };
  hipStream_t side[SIDE_MAX] = {nullptr, nullptr, nullptr, nullptr};   // SV_SIDE_STREAMS of them, taken round-robin: the weight
  hipEvent_t ev_fork = nullptr, ev_join[SIDE_MAX] = {nullptr, nullptr, nullptr, nullptr};   // gradients of different layers are independent
  int side_use = 1;        // streams the current call hands layers to (run_phases)
  int side_count = 0;      // layers forked since the last join
  int nside = 0, side_next = 0, side_slot = 0;   // side_slot: which stream (and which slab workspace) the last wgrad_stream() gave out
  bool side_pending = false;
  // EARLY SIDE WORK (round 6): what the step's first launches do not need runs on side stream 0 beside them -- the decoders' weight images (the encoders' forward
  // reads only its own) and the zero fill of the gradient buffer (first written by the backward).  Both are HBM-bound, the encoder convs beside them matrix-bound;
  // batch-independent time off the critical path of every shard size.  ev_early is waited for before the decoders' forward / the backward / the end of the call.
  hipEvent_t ev_early = nullptr;
  bool early_pending = false;
  // captured steps: SV_GRAPH_SIDE=1 lets the capture fork to the side streams too (every fork / join on its OWN event and only the streams a call really used are
  // joined: round 3's capture, which re-recorded one fork event and joined every stream, replayed wrongly on ROCm 7.2).  Off by default: see DESIGN.md section 7.
  static bool graph_side() { static const bool on = getenv("SV_GRAPH_SIDE") && atoi(getenv("SV_GRAPH_SIDE")) != 0; return on; }
  bool side_allowed() const {
    static const bool off = getenv("SV_NO_SIDE") != nullptr;
    return !(off || (prof_on && prof_filter.empty()) || ((graph_on || dyn) && !graph_side()));
  }
  // fork / join events: one per use while graph replay is on (a captured event node per dependency), the two fixed ones otherwise
  enum { CAP_EV = 96 };
  hipEvent_t ev_cap[CAP_EV] = {};
  int n_cap = 0;
  bool side_used[SIDE_MAX] = {false, false, false, false};
  hipEvent_t fresh_event(hipEvent_t fixed) {
    if (!(graph_on || dyn) || n_cap >= CAP_EV) return fixed;
    if (!ev_cap[n_cap] && hipEventCreateWithFlags(&ev_cap[n_cap], hipEventDisableTiming) != hipSuccess) return fixed;
    return ev_cap[n_cap++];
  }
  hipStream_t early_stream(hipStream_t st) {          // side stream 0, ordered behind everything `st` holds now; nullptr: no side streams (serial modes)
    // OPT-IN (SV_EARLY_SIDE=1).  Measured (profiles/r06_ab.txt): the step is no shorter for it at any shard size -- fp32 512 images 9.152 against 9.138 ms without,
    // 64 images 1.690 / 1.689, bf16 512 images 1.661 / 1.646: the encoders' first layers are HBM-bound themselves (e1 reads the whole batch), the weight images' 165 MB
    // beside them slow them by what the overlap saves (fwd.e1 25 -> 52 us at 64 images).  Same finding as round 3's early optimizer tail: no idle resource to hide it in.
    static const bool on = getenv("SV_EARLY_SIDE") && atoi(getenv("SV_EARLY_SIDE")) != 0;
    if (!on || !side_allowed() || !ensure_side()) return nullptr;
    if (!ev_early && hipEventCreateWithFlags(&ev_early, hipEventDisableTiming) != hipSuccess) return nullptr;
    hipEvent_t ef = fresh_event(ev_fork);
    if (hipEventRecord(ef, st) != hipSuccess || hipStreamWaitEvent(side[0], ef, 0) != hipSuccess) return nullptr;
    side_used[0] = true;
    return side[0];
  }
  hipEvent_t ev_early_now = nullptr;
  int early_done(hipStream_t side0) {                 // the early work is enqueued: mark it
    ev_early_now = fresh_event(ev_early);
    if (hipEventRecord(ev_early_now, side0) != hipSuccess) return (int)hipGetLastError();
    early_pending = true;
    return SV_OK;
  }
  int early_wait(hipStream_t st) {
    if (!early_pending) return SV_OK;
    early_pending = false;
    return hipStreamWaitEvent(st, ev_early_now, 0) == hipSuccess ? SV_OK : (int)hipGetLastError();
  }
  bool ensure_side() {
    if (nside) return true;
    {
      // (priority: streams.hip creates the shared streams at the lowest priority; SV_SIDE_PRIO_NORMAL is read there)
      static const int want = getenv("SV_SIDE_STREAMS") ? atoi(getenv("SV_SIDE_STREAMS")) : 2;      // taken; `side_use` of them are used per call
      const int k = want < 1 ? 1 : want > SIDE_MAX - 1 ? SIDE_MAX - 1 : want;   // the last workspace slot belongs to the main stream
      if (!ev_fork && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return false;
      for (int i = 0; i < k && i < SV_SHARED_STREAMS; ++i) {
        side[i] = sv_shared_stream(i);               // the process's shared side streams (streams.hip): never a stream per plan
        if (!side[i]) break;
        (void)hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming);
        ++nside;
      }
    }
    return nside > 0;
  }
  hipStream_t wgrad_stream(hipStream_t st) {
    side_slot = 0;
    if (!side_allowed()) return st;     // the full per-kernel table wants serial launches; a captured fork/join replayed wrongly on ROCm 7.2 (corrupt gradients, then a crash): captures stay single-stream
    if (!ensure_side()) return st;
    const int use = side_use < nside ? side_use : nside;
    static const char* order = getenv("SV_SIDE_ORDER");        // experiment: stream per forked layer in launch order, e.g. "0110" (repeats)
    int slot = side_next % use;
    if (order && order[0]) { const int c = order[side_count % (int)strlen(order)] - '0'; if (c >= 0 && c < use) slot = c; }
    ++side_count;
    hipEvent_t ef = fresh_event(ev_fork);
    if (hipEventRecord(ef, st) != hipSuccess || hipStreamWaitEvent(side[slot], ef, 0) != hipSuccess) return st;
    side_used[slot] = true;
    // test hook (sv_lgvae_plan_debug "side_delay_us", tests/test_gpu_dist.py): hold the side stream back at its first use of a step, so that a consumer
    // of the gradients that does not wait for the side stream's part (a missing bucket dependency) reads them before they exist
    if (dbg_side_delay_us > 0 && side_count == 1) sv_plan_spin(side[slot], dbg_side_delay_us);
    side_next = slot + 1;
    side_slot = slot;
    side_pending = true;
    return side[slot];
  }
  int join_side(hipStream_t st) {
    const bool cap = graph_on || dyn;
    if (!side_pending && !(cap && side_used[0])) return SV_OK;        // (a captured call joins the early work's stream too: every forked stream must rejoin the capture)
    side_pending = false;
    side_next = 0;                     // every step hands the layers to the same streams
    side_count = 0;
    for (int i = 0; i < nside; ++i) {
      if (cap && !side_used[i]) continue;                              // (an event of a stream outside the capture must not be waited for inside it)
      hipEvent_t ej = fresh_event(ev_join[i]);
      if (hipEventRecord(ej, side[i]) != hipSuccess || hipStreamWaitEvent(st, ej, 0) != hipSuccess) return (int)hipGetLastError();
    }
    return SV_OK;
  }
  // per-plan test hooks (sv_lgvae_plan_debug): never read from the environment, so nothing a job inherits can switch them on
  int dbg_side_delay_us = 0;
  bool dbg_bucket_skip_side = false;
  // gradient-bucket events (SV_PHASE_BUCKET_EVENTS): [bucket][0 = compute stream, 1 + i = side stream i]
  hipEvent_t ev_bucket[3][1 + SIDE_MAX] = {};
  bool bucket_rec[3][1 + SIDE_MAX] = {};
  bool buckets_now = false;   // the running call records bucket events: weight-gradient reduces must not be deferred past them
  void clear_buckets() {
    for (auto& row : bucket_rec)
      for (auto& b : row) b = false;
  }
  int record_bucket(int k, hipStream_t st) {
    for (int i = 0; i <= SIDE_MAX; ++i) bucket_rec[k][i] = false;
    for (int i = 0; i <= nside; ++i) {                    // every stream that may hold work of this bucket, in its own order
      hipStream_t s = i == 0 ? st : side[i - 1];
      if (i > 0 && !s) continue;
      if (!ev_bucket[k][i] && hipEventCreateWithFlags(&ev_bucket[k][i], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
      if (hipEventRecord(ev_bucket[k][i], s) != hipSuccess) return (int)hipGetLastError();
      bucket_rec[k][i] = true;
    }
    return SV_OK;
  }
  // captured steps (sv_lgvae_graph_enable): one executable graph per distinct (phase mask, buffers, baked scalars)
  bool graph_on = false;
  const SvDynArgs* dyn = nullptr;   // non-null while a step is being captured
  // host-side flags that a phase reads on entry and leaves behind on exit: they select launches (which buffer reparam_kl_bwd reads, whether dz is
  // re-zeroed, whether the ELBO gradient comes from the fused head), so they are part of a captured graph's key and a replay must leave the
  // plan in the state the captured phases left it in
  struct HostState { bool gz_clean, dz_slabs, dz_valid, gz_zero_skipped, nll_fused; };
  HostState host_state() const { return {gz_clean, dz_slabs, dz_valid, gz_zero_skipped, nll_fused}; }
  void set_host_state(const HostState& h) { gz_clean = h.gz_clean; dz_slabs = h.dz_slabs; dz_valid = h.dz_valid; gz_zero_skipped = h.gz_zero_skipped; nll_fused = h.nll_fused; }
  uint64_t host_state_bits() const { return (uint64_t)gz_clean | (uint64_t)dz_slabs << 1 | (uint64_t)dz_valid << 2 | (uint64_t)gz_zero_skipped << 3 | (uint64_t)nll_fused << 4; }
  struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; HostState exit_state{}; };
  std::map<std::vector<uint64_t>, GraphEntry> graphs;
  // profiling
  bool prof_on;
  std::string prof_filter;
  std::vector<ProfEntry> prof;
  std::map<std::string, int> profidx;
  std::vector<ProfPending> pending;
  std::vector<hipEvent_t> event_pool;

  size_t esz() const { return d.dtype == SV_BF16 ? 2 : 4; }
  int64_t add_buf(const std::string& n, int64_t bytes) {
    Buf b;
    b.name = n; b.off = ws_bytes; b.bytes = bytes;
    ws_bytes += (bytes + 255) / 256 * 256;
    bufidx[n] = (int)bufs.size();
    bufs.push_back(b);
    return b.off;
  }
  void* bp(const std::string& n) const {
    auto it = bufidx.find(n);
    return it == bufidx.end() ? nullptr : (void*)(ws + bufs[it->second].off);
  }
  int64_t bbytes(const std::string& n) const { return bufs[bufidx.at(n)].bytes; }
};

namespace {

// SV_ROCTX=1: every plan scope ("fwd.d4", "wgrad.d4", "adam_step" ...) is also a roctx range, so a `rocprofv3 --marker-trace
// --kernel-trace` timeline names the layers directly.  libroctx64 is bound at run time (no link dependency); off by default.
#include <dlfcn.h>
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = getenv("SV_ROCTX");
    if (!e || !atoi(e)) return;
    void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static const Roctx& roctx() { static const Roctx r; return r; }

struct Scope {   // hipEvent bracket around one launch when profiling is on
  sv_lgvae_plan* p; hipStream_t st; int entry = -1, entry2 = -1; hipEvent_t a = nullptr, b = nullptr, m1 = nullptr, m2 = nullptr;
  bool on, on2 = false;
  hipEvent_t get() {
    hipEvent_t e;
    if (!p->event_pool.empty()) { e = p->event_pool.back(); p->event_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  }
  int find(const std::string& name, double flops, double bytes) {
    auto it = p->profidx.find(name);
    if (it != p->profidx.end()) return it->second;
    const int e = (int)p->prof.size();
    p->profidx[name] = e;
    p->prof.push_back(ProfEntry{name, flops, bytes, 0.0, 0, flops});
    return e;
  }
  bool marked = false;
  void issued(double fl) { if (entry >= 0) p->prof[entry].issued = fl; }      // the launch set runs a strength-reduced form: FLOPs it really multiplies
  Scope(sv_lgvae_plan* p_, hipStream_t st_, const std::string& name, double flops, double bytes) : p(p_), st(st_), on(p_->prof_on) {
    if (roctx().push) { roctx().push(name.c_str()); marked = true; }
    if (on && !p->prof_filter.empty() && p->prof_filter != name) on = false;
    if (!on) return;
    entry = find(name, flops, bytes);
    a = get(); b = get();
    (void)hipEventRecord(a, st);
  }
  // a launch with a second stage (wgrad tile kernel -> slab reduce): the callee records ev[0], ev[1] between
  // the two, and the second stage is booked under its own name
  void split(const std::string& name2, double bytes2, hipEvent_t ev[2]) {
    ev[0] = ev[1] = nullptr;
    if (!p->prof_on) return;
    on2 = p->prof_filter.empty() || p->prof_filter == name2;
    if (!on && !on2) return;
    if (on2) entry2 = find(name2, 0, bytes2);
    m1 = ev[0] = get(); m2 = ev[1] = get();
    if (!b) b = get();
  }
  ~Scope() {
    if (marked) roctx().pop();
    if (!b) return;
    (void)hipEventRecord(b, st);
    if (!m1) { p->pending.push_back(ProfPending{entry, a, b}); return; }
    if (on) p->pending.push_back(ProfPending{entry, a, m1}); else p->event_pool.push_back(m1);
    if (on2) p->pending.push_back(ProfPending{entry2, m2, b});
    else { p->event_pool.push_back(m2); p->event_pool.push_back(b); }
  }
};

static void build_layers(sv_lgvae_plan* p) {
  const sv_lgvae_desc& d = p->d;
  const int H = d.H, W = d.W, B = d.B;
  const int F = (H / 8) * (W / 8) * 128;
  const int Lg = d.global_latent, Ll = d.local_latent;
  auto mk = [&](const std::string& name, int h, int w, int cin, int cout, int k, int s, int act, int ldx,
                int ldy, int yf32, int kparam, bool need_dgrad) {
    Layer L;
    L.name = name;
    L.d = sv_conv_desc{B, h, w, cin, cout, k, k, s, act, d.dtype, ldx, ldy, yf32, 0};
    L.kparam = kparam; L.bparam = kparam + 1; L.need_dgrad = need_dgrad;
    L.wf_off = 0; L.wdp_off = -1; L.polyc = false;
    for (int i = 0; i < 4; ++i) L.wd_off[i] = 0;
    return L;
  };
  for (int e = 0; e < 2; ++e) {
    const int pb = e * 10;
    const int L = e == 0 ? Lg : Ll;
    const std::string pre = e == 0 ? "enc_x." : "enc_xh.";
    p->enc[e][0] = mk(pre + "e1", H, W, 3, 32, 6, 2, SV_ACT_RELU, 8, 32, 0, pb + 0, false);
    p->enc[e][1] = mk(pre + "e2", H / 2, W / 2, 32, 64, 6, 2, SV_ACT_RELU, 32, 64, 0, pb + 2, true);
    p->enc[e][2] = mk(pre + "e3", H / 4, W / 4, 64, 128, 4, 2, SV_ACT_RELU, 64, 128, 0, pb + 4, true);
    // e4_mean | e4_sd as ONE dense GEMM with N = 2L (vae/model.py:41-42,:111-112)
    p->enc[e][3] = mk(pre + "head", 1, 1, F, 2 * L, 1, 1, SV_ACT_NONE, F, 2 * L, 1, pb + 6, true);
  }
  for (int k = 0; k < 2; ++k) {
    const int pb = 20 + k * 10;
    const int Lz = k == 0 ? Lg + Ll : Ll;
    const std::string pre = k == 0 ? "dec_x." : "dec_xh.";
    p->dec[k][0] = mk(pre + "d1", 1, 1, Lz, F, 1, 1, SV_ACT_RELU, Lg + Ll, F, 0, pb + 0, true);
    p->dec[k][1] = mk(pre + "d2", H / 8, W / 8, 128, 128, 4, 1, SV_ACT_RELU, 128, 128, 0, pb + 2, true);
    p->dec[k][2] = mk(pre + "d3", H / 4, W / 4, 128, 64, 4, 1, SV_ACT_RELU, 128, 64, 0, pb + 4, true);
    p->dec[k][3] = mk(pre + "d4", H / 2, W / 2, 64, 32, 6, 1, SV_ACT_RELU, 64, 32, 0, pb + 6, true);
    p->dec[k][4] = mk(pre + "d5", H, W, 32, 6, 6, 1, SV_ACT_NONE, 32, 6, 1, pb + 8, true);
    // the three bilinear resizes are fused into the staging of d3/d4/d5's forward and wgrad tiles (needs >= 16 output pixels per image for
    // the tile kernels: any H >= 16 here); fp32 since round 4 too (wgrad_tile_f32.hip; SV_F32_MATERIALISE_UPSAMPLE=1: u2 / u3 / u4 written out)
    static const bool no_fuse = getenv("SV_NO_FUSED_UPSAMPLE") != nullptr;
    static const bool f32_mat = getenv("SV_F32_MATERIALISE_UPSAMPLE") != nullptr;
    if ((d.dtype == SV_BF16 || !f32_mat) && !no_fuse && H >= 16)
      for (int l = 2; l <= 4; ++l) p->dec[k][l].d.ups_in = 1;
  }
}

static void build_prep_jobs(sv_lgvae_plan* p) {
  int64_t arena = 0;
  int blocks = 0;
  auto push = [&](PrepJob j) {
    j.first_block = blocks;
    blocks += j.nblocks;
    p->jobs.push_back(j);
  };
  auto align = [&]() { arena = (arena + 127) / 128 * 128; };
  auto polyd_jobs = [&](Layer& L) {                       // polyphase input gradient (conv_geom.h: svg_polyd): main, edge and corner images
    if (!L.need_dgrad || !svg_polyd(&L.d)) return;
    align(); L.wdp_off = arena;
    for (int which = 0; which < 3; ++which) {
      PrepJob j;
      svg_prep_job_polyd(&L.d, which, &j);
      j.src_off = p->params[L.kparam].off; j.dst_off = arena;
      arena += svg_polyd_elems(&L.d, which);
      push(j);
    }
  };
  auto do_layer = [&](Layer& L, bool is_head) {
    L.polyc = !is_head && svg_polyc(&L.d) && svk_polyc_fwd_plannable(&L.d);     // (the class problems must plan on the tile kernel: else the direct form is prepared)
    if (L.polyc) {
      // per-class polyphase forward (conv_geom.h: svg_polyc): four class images + the border-class image, contiguous from wf_off in the
      // order svk_polyc_fwd_multi expects; the input / weight gradients keep their own forms
      align(); L.wf_off = arena;
      for (int c = 0; c < 4; ++c) {
        PrepJob j;
        svg_prep_job_polyc(&L.d, c, &j);
        j.src_off = p->params[L.kparam].off; j.dst_off = arena;
        arena += svg_polyc_class_elems(&L.d, c);
        push(j);
      }
      PrepJob jf;
      svg_prep_job_polyc_fix(&L.d, &jf);
      jf.src_off = p->params[L.kparam].off; jf.dst_off = arena;
      arena += svg_polyc_fix_elems(&L.d);
      push(jf);
      if (L.need_dgrad)
        for (int c = 0; c < svg_dgrad_classes(&L.d); ++c) {
          PrepJob jd;
          svg_prep_job_dgrad(&L.d, c, &jd);
          jd.src_off = p->params[L.kparam].off;
          align(); L.wd_off[c] = arena; jd.dst_off = arena;
          arena += (int64_t)jd.rows * jd.ntaps * jd.inner;
          push(jd);
        }
      polyd_jobs(L);
    } else if (!is_head) {
      PrepJob j;
      svg_prep_job_fwd(&L.d, &j);
      j.src_off = p->params[L.kparam].off;
      align(); L.wf_off = arena; j.dst_off = arena;
      arena += (int64_t)j.rows * j.ntaps * j.inner;
      push(j);
      if (svg_poly(&L.d)) {                             // border-fix image of the polyphase head, right behind the composite image
        PrepJob jf;
        svg_prep_job_polyfix(&L.d, &jf);
        jf.src_off = p->params[L.kparam].off;
        jf.dst_off = arena;
        arena += (int64_t)jf.rows * jf.ntaps * jf.inner;
        push(jf);
      }
      if (L.need_dgrad)
        for (int c = 0; c < svg_dgrad_classes(&L.d); ++c) {
          PrepJob jd;
          svg_prep_job_dgrad(&L.d, c, &jd);
          jd.src_off = p->params[L.kparam].off;
          align(); L.wd_off[c] = arena; jd.dst_off = arena;
          arena += (int64_t)jd.rows * jd.ntaps * jd.inner;
          push(jd);
        }
      polyd_jobs(L);
    } else {
      // two Keras tensors [F,L] (mean at kparam, sd at kparam+2) -> one [2L][F] forward image
      // and one [F][2L] dgrad image
      const int F = L.d.Cin, L2 = L.d.Cout, Lh = L2 / 2;
      align(); L.wf_off = arena;
      for (int h = 0; h < 2; ++h) {
        PrepJob j;
        memset(&j, 0, sizeof(j));
        j.src_off = p->params[L.kparam + 2 * h].off;
        j.ntaps = 1; j.Cin = F; j.Cout = Lh; j.rows = Lh; j.inner = F; j.inner_ld = F; j.inner_off = 0;
        j.transpose = 0; j.srctap[0] = 0;
        j.dst_off = arena + (int64_t)h * Lh * F;
        j.nblocks = svg_prep_nblocks(&j);
        push(j);
      }
      arena += (int64_t)L2 * F;
      align(); L.wd_off[0] = arena;
      for (int h = 0; h < 2; ++h) {
        PrepJob j;
        memset(&j, 0, sizeof(j));
        j.src_off = p->params[L.kparam + 2 * h].off;
        j.ntaps = 1; j.Cin = F; j.Cout = Lh; j.rows = F; j.inner = Lh; j.inner_ld = L2; j.inner_off = h * Lh;
        j.transpose = 1; j.srctap[0] = 0;
        j.dst_off = arena;
        j.nblocks = svg_prep_nblocks(&j);
        push(j);
      }
      arena += (int64_t)F * L2;
    }
  };
  for (int e = 0; e < 2; ++e)
    for (int l = 0; l < 4; ++l) do_layer(p->enc[e][l], l == 3);
  p->dec_block0 = blocks;
  for (int k = 0; k < 2; ++k)
    for (int l = 0; l < 5; ++l) do_layer(p->dec[k][l], false);
  p->arena_elems = arena + 128;
  p->prep_blocks = blocks;
}

static void build_buffers(sv_lgvae_plan* p) {
  const sv_lgvae_desc& d = p->d;
  const int64_t B = d.B, H = d.H, W = d.W, es = (int64_t)p->esz();
  const int64_t F = (H / 8) * (W / 8) * 128;
  const int Lg = d.global_latent, Ll = d.local_latent, Lc = Lg + Ll;
  p->ws_bytes = 0;
  p->add_buf("jobs", (int64_t)p->jobs.size() * sizeof(PrepJob));
  p->add_buf("warena", p->arena_elems * es);
  p->add_buf("wgrad_ws", SV_WGRAD_WS_BYTES * SV_WGRAD_MAX_MULTI * sv_lgvae_plan::SIDE_MAX);   // per side stream, per problem
  if (getenv("SV_DEFER_REDUCE")) p->add_buf("wslab", SV_WGRAD_WS_BYTES * SV_WGRAD_MAX_MULTI * 7);    // per tile-wgrad layer (e1 e2 e3 d2 d3 d4 d5) and problem: deferred reduces
  p->add_buf("polyfix_x", svk_poly_fix_ws_bytes((int)B, (int)H / 2, (int)W / 2));   // border terms of the polyphase head (poly_fix.hip)
  p->add_buf("polyfix_xh", svk_poly_fix_ws_bytes((int)B, (int)H / 2, (int)W / 2));
  {   // border terms of the per-class polyphase layers (d3, d4): one region per network, sized for the largest layer (the layers run one after the other)
    int64_t need = 256;
    for (int l = 2; l <= 4; ++l)
      if (p->dec[0][l].polyc) need = std::max<int64_t>(need, svg_polyc_fix_ws_bytes(&p->dec[0][l].d));
    p->add_buf("polycfix_x", need);
    p->add_buf("polycfix_xh", need);
    int64_t needd = 256;                                  // edge terms of the polyphase input gradients (d4, d5): one region per network
    for (int l = 2; l <= 4; ++l)
      if (svg_polyd(&p->dec[0][l].d)) needd = std::max<int64_t>(needd, svg_polyd_ws_bytes(&p->dec[0][l].d));
    p->add_buf("polyd_x", needd);
    p->add_buf("polyd_xh", needd);
    // polyphase weight gradients at fp32 (polyc_wgrad.hip): dW' + frame slabs per layer and network (the layers' launches may overlap across streams)
    for (int l = 2; l <= 4; ++l) {
      const int64_t fl = svk_polyc_wgrad_ws_floats(&p->dec[0][l].d);
      if (!fl) continue;
      p->add_buf("polycw" + std::to_string(l) + "_x", fl * 4);
      p->add_buf("polycw" + std::to_string(l) + "_xh", fl * 4);
    }
  }
  p->add_buf("polyw_x", svk_poly_wgrad_ws_floats(32, SV_POLY_WGRAD_NWG) * 4);        // polyphase weight gradient of the head: dW', dbias', frame slabs
  p->add_buf("polyw_xh", svk_poly_wgrad_ws_floats(32, SV_POLY_WGRAD_NWG) * 4);
  {   // K-slice slabs of the heads' forward and d1's input gradient (latent_gemm.hip): [S][B][N] fp32 per network
    const int64_t Fd = (H / 8) * (W / 8) * 128;
    // the four launches that write them -- heads of x / x-hat (N = 2 Lg / 2 Ll), d1's input gradient of decoder_x / decoder_x-hat (N = Lg + Ll / Ll) --
    // pick their split PER PROBLEM: sized as the maximum of the four actual (split, N) products (with unequal latent sizes a
    // per-problem split x N could exceed a product formed from two of them); the launch sites check the capacity again and fall back
    int64_t need = 0;
    for (const int64_t Nq : {2 * Lg, 2 * Ll, Lg + Ll, Ll}) {
      const int64_t b = (int64_t)svk_nt_gemm_pick_splitk((int)B, (int)Nq, (int)Fd, 2) * B * Nq * 4;
      need = b > need ? b : need;
    }
    p->add_buf("lat_ws_x", need);
    p->add_buf("lat_ws_xh", need);
  }
  p->add_buf("dyn", sizeof(SvDynArgs));
  p->add_buf("losses", 8 * 4);
  p->add_buf("metric_acc", 8 * 4);
  p->add_buf("zcat", B * Lc * es);
  // per-network buffers: the x and x-hat twins of one kind are adjacent (x-hat exactly `bytes` after x
  // when the size is a multiple of 256 B), so pointwise kernels can take both as one batch of 2B
  auto twin = [&](const char* kind, int64_t bytes_x, int64_t bytes_xh) {
    p->add_buf(std::string(kind) + "x", bytes_x);
    p->add_buf(std::string(kind) + "xh", bytes_xh);
  };
  auto same = [&](const char* kind, int64_t bytes) { twin(kind, bytes, bytes); };
  same("in8_", B * H * W * 8 * es);
  same("a1_", B * (H / 2) * (W / 2) * 32 * es);
  same("a2_", B * (H / 4) * (W / 4) * 64 * es);
  same("a3_", B * F * es);
  // zeroed every step by ONE memset (split-K / atomic accumulation targets): pre_x .. gz_xh are adjacent
  twin("pre_", B * 2 * Lg * 4, B * 2 * Ll * 4);
  twin("gz_", B * Lc * 4, B * Ll * 4);
  twin("z_mean_", B * Lg * 4, B * Ll * 4);
  twin("z_sig_", B * Lg * 4, B * Ll * 4);
  twin("z_", B * Lg * 4, B * Ll * 4);
  twin("eps_", B * Lg * 4, B * Ll * 4);
  same("kl_", B * 4);
  twin("ghead_", B * 2 * Lg * es, B * 2 * Ll * es);
  same("ga3_", B * F * es);
  same("ga2_", B * (H / 4) * (W / 4) * 64 * es);
  same("ga1_", B * (H / 2) * (W / 2) * 32 * es);
  // decoders
  same("h1_", B * F * es);
  same("h2_", B * F * es);
  same("u2_", B * (H / 4) * (W / 4) * 128 * es);
  same("h3_", B * (H / 4) * (W / 4) * 64 * es);
  same("u3_", B * (H / 2) * (W / 2) * 64 * es);
  same("h4_", B * (H / 2) * (W / 2) * 32 * es);
  same("u4_", B * H * W * 32 * es);
  same("out6_", B * H * W * 6 * 4);
  same("nll_", B * 4);
  same("nllpart_", sv_dlogistic_nll_workspace_bytes((int)B, (int)H, (int)W));
  same("g5_", B * H * W * 8 * es);
  same("gu4_", B * H * W * 32 * es);
  same("g4_", B * (H / 2) * (W / 2) * 32 * es);
  same("gu3_", B * (H / 2) * (W / 2) * 64 * es);
  same("g3_", B * (H / 4) * (W / 4) * 64 * es);
  same("gu2_", B * (H / 4) * (W / 4) * 128 * es);
  same("g2_", B * F * es);
  same("g1_", B * F * es);
}

// the latent block's GEMMs on latent_gemm.hip (LDS-DMA phases, split-K through fp32 slabs summed in slice order) instead of the im2col
// kernel's split-K atomics; SV_NO_LATENT_GEMM restores the old launches (A/B)
// the slab sums of the two split-K launches live in their consumers (Sampling + KL forward / backward) instead of nt_slab_reduce_kernel
static bool latent_fuse_on() {
  static const bool off = getenv("SV_NO_LATENT_FUSE") != nullptr;
  return !off;
}
static bool latent_gemm_on(const sv_lgvae_plan* p) {
  static const bool off = getenv("SV_NO_LATENT_GEMM") != nullptr;
  static const bool off32 = getenv("SV_NO_LATENT_GEMM_F32") != nullptr;       // (A/B: the fp32 forms of latent_gemm.hip)
  return !off && (p->d.dtype == SV_BF16 || !off32);
}

static double conv_flops(const sv_conv_desc& d) {
  return 2.0 * d.B * svg_oh(&d) * svg_ow(&d) * (double)d.Cout * d.KH * d.KW * d.Cin;
}

// FLOPs a POLYPHASE form of an upsample -> K x K conv layer really multiplies on the matrix pipe (real channels only; DESIGN.md section 5 "issued"):
//   main term: `taps` tap products per LOW-RES pixel where the direct form has 4 K^2 (per-class forms and the stride-2 input gradient: (5+4)^2 = 81 of 144 for
//   K = 6; the head's merged 25-tap form multiplies all 4 x 25 = 100 -- 19 of them structural zeros of the composite image -- of 144);
//   border terms: forward / weight gradient 2 (K-1) edge lines of H (W) hi-res pixels x K taps (poly_fix.hip, polyc_wgrad.hip's frame kernel); input gradient
//   four low-res edge lines x 4 strip rows x 9 taps (polyd_dgrad.hip).  Corner terms, the projection and the frame sum (< 1 % of a layer) are not counted.
static double poly_issued(const sv_conv_desc& d, int taps, bool dgrad) {
  const double main_ = conv_flops(d) * taps / (4.0 * d.KH * d.KW);
  const double border = 2.0 * d.B * (double)(d.H + d.W) * d.Cin * d.Cout * (dgrad ? 36.0 : (double)(d.KH - 1) * d.KH);
  return main_ + border;
}

// algorithmic HBM bytes of one conv-like launch (what a perfect kernel must move once): kind 0 forward (input [low-res when the
// resize is fused] + output + weights), 1 input gradient (dY + mask + dX), 2 weight gradient (input + dY + fp32 dW)
static double conv_bytes(const sv_conv_desc& d, int kind, size_t es) {
  const double in_px = (double)d.B * (d.ups_in ? d.H / 2 : d.H) * (d.ups_in ? d.W / 2 : d.W);
  const double out_px = (double)d.B * svg_oh(&d) * svg_ow(&d);
  const double w = (double)d.KH * d.KW * d.Cin * d.Cout;
  const double x = in_px * d.Cin * es, y = out_px * d.Cout * (d.y_f32 ? 4 : es);
  if (kind == 0) return x + y + w * es;
  if (kind == 1) return out_px * d.Cout * es + 2 * x + w * es;
  return x + out_px * d.Cout * es + w * 4;
}

#define SV_TRY(x)            \
  do {                       \
    int rc__ = (x);          \
    if (rc__) return rc__;   \
  } while (0)

// The x and x-hat networks have twin layers of identical geometry: n of them go out as ONE launch
// (blockIdx.z picks the problem), which halves the per-launch fixed cost that dominates at B<=512.
struct FusedNll { const float* images6; void* grad[2]; float* part[2]; float gscale; int noout; };   // loss in the head's epilogue (tile_conv.hip)
static int run_fwd_layers(sv_lgvae_plan* p, int n, Layer* const* L, const void* const* x, const float* params,
                          void* const* y, hipStream_t st, const FusedNll* nll = nullptr) {
  TapGemmArgs a[2];
  double fl = 0, by = 0;
  for (int i = 0; i < n; ++i) {
    svg_fwd_args(&L[i]->d, &a[i]);
    a[i].A = x[i];
    a[i].Wt = (char*)p->bp("warena") + L[i]->wf_off * p->esz();
    a[i].bias = params + p->params[L[i]->bparam].off;
    a[i].out = y[i];
    if (nll) {
      a[i].nll_img = nll->images6; a[i].nll_ch = 3 * i; a[i].nll_grad = nll->grad[i]; a[i].nll_part = nll->part[i];
      a[i].nll_gscale = nll->gscale; a[i].nll_noout = nll->noout;
    }
    fl += conv_flops(L[i]->d);
    by += conv_bytes(L[i]->d, 0, p->esz());
  }
  Scope sc(p, st, "fwd." + L[0]->name.substr(L[0]->name.find('.') + 1), fl, by);
  if (latent_gemm_on(p) && !nll && L[0]->d.H == 1 && L[0]->d.W == 1 && L[0]->d.KH == 1 && !L[0]->d.y_f32) {   // Dense (d1): latent_gemm.hip
    NtGemmProb q[2];
    for (int i = 0; i < n; ++i) {
      const sv_conv_desc& d = L[i]->d;
      NtGemmProb& g = q[i];
      memset(&g, 0, sizeof(g));
        g.f32 = p->d.dtype == SV_F32;
      g.A = x[i]; g.lda = d.ldx;
      g.W = a[i].Wt; g.ldw = svg_cin_pad(&d);
      g.out = y[i]; g.ldo = d.ldy;
      g.bias = a[i].bias; g.act = d.act;
      g.M = d.B; g.N = d.Cout; g.K = d.Cin; g.splitk = 1;
    }
    const int rc = svk_nt_gemm_multi(q, n, L[0]->d.B >= 256 ? 128 : 64, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  if (L[0]->polyc && !nll && n <= 2) {
    { double is = 0; for (int i = 0; i < n; ++i) is += poly_issued(L[i]->d, L[i]->d.KH == 6 ? 81 : 49, false); sc.issued(is); }
    // per-class polyphase (d4, d3): border kernel, then the four class problems of both networks in one launch
    const void* wf[2];
    const float* bs[2];
    void* fws[2] = {p->bp("polycfix_x"), p->bp("polycfix_xh")};
    for (int i = 0; i < n; ++i) { wf[i] = a[i].Wt; bs[i] = a[i].bias; }
    return svk_polyc_fwd_multi(&L[0]->d, n, x, wf, bs, y, fws, st);
  }
  if (svg_poly(&L[0]->d)) {
    { double is = 0; for (int i = 0; i < n; ++i) is += poly_issued(L[i]->d, 100, false); sc.issued(is); }
    // polyphase head: the out-of-image taps of the border rows / columns go to a workspace first; the conv's epilogue adds them
    const void* wfix[2];
    float* fixbuf[2];
    static const char* fb_name[2] = {"polyfix_x", "polyfix_xh"};
    const sv_conv_desc& d = L[0]->d;
    for (int i = 0; i < n; ++i) {
      wfix[i] = (const char*)a[i].Wt + (int64_t)32 * 25 * svg_cin_pad(&L[i]->d) * p->esz();
      fixbuf[i] = (float*)p->bp(fb_name[i]);
      a[i].fix = fixbuf[i];
    }
    SV_TRY(svk_poly_fix_multi(n, x, wfix, nullptr, fixbuf, d.B, d.H / 2, d.W / 2, d.ldx, d.Cout, st, d.dtype));
  }
  return svk_conv_dispatch_multi(a, n, L[0]->d.dtype, svg_pick_cfg(L[0]->d.Cout), st);
}
static int run_fwd_layer(sv_lgvae_plan* p, Layer& L, const void* x, const float* params, void* y, hipStream_t st) {
  Layer* Lp = &L;
  return run_fwd_layers(p, 1, &Lp, &x, params, &y, st);
}

// all parity classes of all n twin layers in one launch (stride 2: 4 classes x 2 networks = 8 problems)
static int run_dgrad_layers(sv_lgvae_plan* p, int n, Layer* const* L, const void* const* dy, const void* const* mask,
                            void* const* dx, bool f32_atomic, hipStream_t st, bool adj = false) {
  TapGemmArgs a[SV_MAX_MULTI];
  double fl = 0, by = 0;
  for (int i = 0; i < n; ++i) by += conv_bytes(L[i]->d, 1, p->esz());
  int m = 0, tap_cfg = svg_pick_cfg(L[0]->d.Cin);
  bool mixed = false;
  const int ncls = svg_dgrad_classes(&L[0]->d);
  if (!adj && !f32_atomic && ncls == 4) {              // classes that share one dY window (e2): one problem per network
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
      ok = svg_dgrad_merged_args(&L[i]->d, L[i]->wd_off, &a[i]);
      a[i].A = dy[i]; a[i].Wt = (char*)p->bp("warena") + L[i]->wd_off[0] * p->esz(); a[i].out = dx[i]; a[i].mask = mask[i];
      fl += conv_flops(L[i]->d);
    }
    if (ok) {
      Scope sc(p, st, "dgrad." + L[0]->name.substr(L[0]->name.find('.') + 1), fl, by);
      const int rc = svk_conv_dispatch_multi(a, n, L[0]->d.dtype, svg_pick_cfg(a[0].N), st);
      if (rc != SV_E_UNSUPPORTED) return rc;
    }
    fl = 0;
  }
  for (int i = 0; i < n; ++i) {
    fl += conv_flops(L[i]->d);
    for (int c = 0; c < ncls; ++c, ++m) {
      uint8_t srctap[SV_MAX_TAPS];
      svg_dgrad_args(&L[i]->d, c, &a[m], srctap);
      a[m].A = dy[i];
      a[m].Wt = (char*)p->bp("warena") + L[i]->wd_off[c] * p->esz();
      a[m].out = dx[i];
      a[m].mask = mask[i];
      a[m].adj = adj ? 1 : 0;           // dx = the LOW-RES gradient, mask = the low-res activation (row_conv.hip)
      if (f32_atomic) {
        a[m].out_f32 = 1;
        a[m].accum = 1;
        int c2 = tap_cfg;
        a[m].splitk = svg_choose_splitk(a[m].M, a[m].N, (a[m].P + 7) / 8, &c2);
        if (m == 0) tap_cfg = c2;
        else if (c2 != tap_cfg) mixed = true;
      }
    }
  }
  if (adj) {                            // no fused kernel for this geometry: nothing was launched, the caller falls back
    if (!svk_row_conv_supported(a, m, L[0]->d.dtype)) return SV_E_UNSUPPORTED;
  }
  Scope sc(p, st, "dgrad." + L[0]->name.substr(L[0]->name.find('.') + 1), fl, by);
  if (mixed) {   // split-K problems whose widths pick different tiles: one launch each
    for (int i = 0; i < m; ++i) {
      int c2 = svg_pick_cfg(L[0]->d.Cin);
      a[i].splitk = svg_choose_splitk(a[i].M, a[i].N, (a[i].P + 7) / 8, &c2);
      SV_TRY(svk_conv_dispatch(a[i], L[0]->d.dtype, c2, st));
    }
    return SV_OK;
  }
  // the classes of a stride-2 layer have different tap counts but plan to the same tile grid
  return svk_conv_dispatch_multi(a, m, L[0]->d.dtype, tap_cfg, st);
}
static int run_dgrad_layer(sv_lgvae_plan* p, Layer& L, const void* dy, const void* mask, void* dx, bool f32_atomic,
                           hipStream_t st) {
  Layer* Lp = &L;
  return run_dgrad_layers(p, 1, &Lp, &dy, &mask, &dx, f32_atomic, st);
}

static int run_wgrad_layers(sv_lgvae_plan* p, int n, Layer* const* L, const void* const* x, const void* const* dy,
                            float* grads, hipStream_t st) {
  WgradArgs a[SV_WGRAD_MAX_MULTI];
  double fl = 0, by = 0;
  for (int i = 0; i < n; ++i) by += conv_bytes(L[i]->d, 2, p->esz());
  const int64_t wsb = p->bbytes("wgrad_ws") / (SV_WGRAD_MAX_MULTI * sv_lgvae_plan::SIDE_MAX);   // one partial-sum slab region per stream and problem
  const std::string ln = L[0]->name.substr(L[0]->name.find('.') + 1), nm = "wgrad." + ln;
  // layers named in SV_WGRAD_MAIN keep their weight gradient on the main stream.  Measured (B = 512, 64x64): the side stream is
  // the critical path of the backward pass (it ends ~0.3 ms after the last input gradient), and d5's weight gradient slows the
  // concurrent d4 input gradient 2.5x; with d5 and the two tail layers e1, e2 on the main stream the step is 1.4 % shorter
  // ("" = everything on the side stream)
  // Batch dependence (re-measured per shard size): d5 on the main stream pays from ~768 images per launch (B = 512: -0.4 %)
  // and costs below that (B = 256: +0.6 %, 128: +4 %, 64: +2 %); e1 / e2 on the main stream pay at every size.
  // Round 3: d4's weight gradient on the rolling-window kernel (wgrad_roll.hip) is one persistent 8-wave workgroup per CU with the whole
  // register file: nothing co-resides with it, so beside the input-gradient chain it only gets the CUs that chain leaves (0.147 ms alone,
  // 0.27 live).  From 768 images per launch it stays on the main stream instead of d5's (re-measured, B = 512: "e1,e2,d5" 2.073 / 2.078 ms,
  // "e1,e2,d4" 2.056 / 2.063, "e1,e2,d5,d4" 2.072, "e1,e2,d4,d3" 2.076; B = 256 +-0, B = 128 / 64: +2.5 %, so not below the threshold)
  static const char* on_main_env = getenv("SV_WGRAD_MAIN");
  static const bool roll_off = getenv("SV_NO_WGRAD_ROLL") != nullptr;
  const bool big = n * L[0]->d.B >= 768;
  // With TWO side streams (whole bf16 steps at this size, see run_phases) only the two tail layers stay: "e1,e2" 2.004 / 2.006 ms,
  // "e1,e2,d1" 2.008, "e1" / "e2" 2.03, "" 2.039, "e1,e2,d4" 2.061, "e1,e2,d5" 2.119.
  // fp32 (round 4, with the tile weight gradients: every launch is matrix-pipe-bound, two of them side by side only slow each other): the three
  // encoder layers on the main stream, the decoders' on the side stream -- "e1,e2,e3" 11.65-11.72 ms against "e1,e2,d5" 12.05-12.08, no side stream
  // at all 11.87, "e1,e2" 11.74, "e1" 11.80, "e1,e2,e3,d2" 11.77 (profiles/r04_f32_streams.txt)
  // (two side streams, re-measured with round 4's kernels: "e1" 1.643-1.647 ms, "e1,e2" 1.651-1.653, "e1,d1" 1.650-1.655, "e2" 1.668, "e1,e3" 1.663:
  //  profiles/r04_b512_streams.txt;
  //  512 images per launch, one side stream: "e1" 1.067-1.070 against 1.071-1.074 (profiles/r04_b256_sweep.txt); 256 and 128: "e1,e2" stays)
  const char* on_main = on_main_env ? on_main_env : !big ? (n * L[0]->d.B >= 512 ? "e1" : "e1,e2") : p->side_use >= 2 ? "e1" : (L[0]->d.dtype == SV_BF16 && !roll_off) ? "e1,e2,d4" :
                        L[0]->d.dtype == SV_F32 ? "e1,e2,e3" : "e1,e2,d5";
  if (strstr(on_main, ln.c_str())) p->side_slot = sv_lgvae_plan::SIDE_MAX - 1;   // its own slab workspace: the side streams' slots are in use concurrently
  else st = p->wgrad_stream(st);
  if (latent_gemm_on(p) && L[0]->d.H == 1 && L[0]->d.W == 1 && L[0]->d.KH == 1 && n <= 4) {     // Dense (d1): latent_gemm.hip, whole batch per tile
    TnWgradProb q[4];
    for (int i = 0; i < n; ++i) {
      const sv_conv_desc& d = L[i]->d;
      q[i] = TnWgradProb{x[i], d.ldx, dy[i], d.ldy, grads + p->params[L[i]->kparam].off, grads + p->params[L[i]->bparam].off, d.B, svg_cin_pad(&d),
                         d.Cin, d.Cout, p->d.dtype == SV_F32};
      fl += conv_flops(d);
    }
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && svk_tn_wgrad_supported(q[i]);
    if (ok) {
      Scope sc(p, st, nm, fl, by);
      return svk_tn_wgrad_multi(q, n, st);
    }
    fl = 0;
  }
  // the decoder head's weight gradient in polyphase form (poly_wgrad.hip) from ~768 images per launch (its three small
  // kernels cost more than they save below that: 16 images 42 vs 20 us; 1024 images 132 vs 184 us)
  static const bool no_pw = getenv("SV_NO_POLY_WGRAD") != nullptr;
  // (round 2, tile-kernel main term: 512 images per launch +0.4 %, 1024 -1.0 % -> 768; round 4, rolling-window main term (wgrad_p5.hip): 512 images per
  //  launch -0.7 %, 256 +2.5 % -> 512: profiles/r04_poly_wgrad_min_sweep.txt)
  static const int pw_min = getenv("SV_POLY_WGRAD_MIN") ? atoi(getenv("SV_POLY_WGRAD_MIN")) : 512;
  if (!no_pw && L[0]->d.dtype == SV_BF16 && svg_poly(&L[0]->d) && n * L[0]->d.B >= pw_min && n <= 2 &&
      svk_poly_wgrad_supported(L[0]->d.H / 2, L[0]->d.W / 2, svg_cin_pad(&L[0]->d), L[0]->d.Cout)) {   // else: the direct form below
    static const char* pw_name[2] = {"polyw_x", "polyw_xh"};
    float* pw[2];
    float *dwv[2], *dbv[2];
    const int Cin = svg_cin_pad(&L[0]->d);
    for (int i = 0; i < n; ++i) {
      svg_poly_wgrad_args(&L[i]->d, &a[i]);
      pw[i] = (float*)p->bp(pw_name[i]);
      a[i].A = x[i]; a[i].dY = dy[i]; a[i].dW = pw[i]; a[i].dbias = pw[i] + 25 * Cin * 32;
      a[i].ws = (float*)((char*)p->bp("wgrad_ws") + (p->side_slot * SV_WGRAD_MAX_MULTI + i) * wsb); a[i].ws_bytes = wsb;
      dwv[i] = grads + p->params[L[i]->kparam].off; dbv[i] = grads + p->params[L[i]->bparam].off;
      fl += conv_flops(L[i]->d);
    }
    Scope sc(p, st, nm, fl, by);
    { double is = 0; for (int i = 0; i < n; ++i) is += poly_issued(L[i]->d, 100, false); sc.issued(is); }
    SV_TRY(svk_wgrad_tile_multi(a, n, st));
    const sv_conv_desc& d = L[0]->d;
    return svk_poly_wgrad_finish(n, x, dy, pw, dwv, dbv, d.B, d.H / 2, d.W / 2, d.ldx, Cin, d.Cout, SV_POLY_WGRAD_NWG, st);
  }
  // fp32: the polyphase weight gradient of the upsample -> conv layers (polyc_wgrad.hip: d4 per parity class, the head merged)
  static const int pcw_min = getenv("SV_POLYC_WGRAD_MIN") ? atoi(getenv("SV_POLYC_WGRAD_MIN")) : 0;
  if (L[0]->d.dtype == SV_F32 && n <= 2 && n * L[0]->d.B >= pcw_min && svg_polyc_wgrad_form(&L[0]->d) && !(p->polycw_bad >> (ln[1] - '0' - 1) & 1) &&
      p->bufidx.count("polycw" + std::to_string(ln[1] - '0' - 1) + "_x")) {
    const std::string base = "polycw" + std::to_string(ln[1] - '0' - 1);          // layer name d<k>: index k - 1 in dec[]
    float *dwv[2], *dbv[2], *slab[2], *pw[2];
    for (int i = 0; i < n; ++i) {
      dwv[i] = grads + p->params[L[i]->kparam].off; dbv[i] = grads + p->params[L[i]->bparam].off;
      slab[i] = (float*)((char*)p->bp("wgrad_ws") + (p->side_slot * SV_WGRAD_MAX_MULTI + i) * wsb);
      pw[i] = (float*)p->bp(base + (i == 0 ? "_x" : "_xh"));
      fl += conv_flops(L[i]->d);
    }
    Scope sc(p, st, nm, fl, by);
    { double is = 0; for (int i = 0; i < n; ++i) is += poly_issued(L[i]->d, svg_polyc_wgrad_form(&L[i]->d) == 2 ? 100 : 81, false); sc.issued(is); }
    const int rc = svk_polyc_wgrad_multi(&L[0]->d, n, x, dy, dwv, dbv, slab, wsb, pw, st);
    if (rc != SV_E_UNSUPPORTED) return rc;
    p->polycw_bad |= 1u << (ln[1] - '0' - 1);
    fl = 0;
  }
  // measured (profiles/r03_f_defer.txt): one launch instead of seven saves 20-26 us of SERIAL time, but the slabs (~250 MB a step) are then read
  // cold from HBM on the critical path after the join instead of warm from L2 / MALL on the side stream: B = 512 -0.5..1 %, B = 64 +5 %.  Opt-in.
  static const bool no_defer = getenv("SV_DEFER_REDUCE") == nullptr;
  static const char* slots[7] = {"e1", "e2", "e3", "d2", "d3", "d4", "d5"};
  int slot = -1;
  for (int k = 0; k < 7; ++k) if (ln == slots[k]) slot = k;
  // (not while gradient-bucket events are recorded: a deferred reduce writes dW / dbias AFTER the events the all-reduce waits on)
  const bool defer = !no_defer && !p->buckets_now && slot >= 0 && L[0]->d.dtype == SV_BF16 && n <= SV_WGRAD_MAX_MULTI;
  for (int i = 0; i < n; ++i) {
    svg_wgrad_args(&L[i]->d, &a[i]);
    a[i].A = x[i]; a[i].dY = dy[i];
    a[i].dW = grads + p->params[L[i]->kparam].off;
    a[i].dbias = grads + p->params[L[i]->bparam].off;
    a[i].ws = (float*)((char*)p->bp("wgrad_ws") + (p->side_slot * SV_WGRAD_MAX_MULTI + i) * wsb); a[i].ws_bytes = wsb;
    if (defer) {                        // its own slab region, reduced with every other layer's after the backward pass
      a[i].ws = (float*)((char*)p->bp("wslab") + (int64_t)(slot * SV_WGRAD_MAX_MULTI + i) * SV_WGRAD_WS_BYTES); a[i].ws_bytes = SV_WGRAD_WS_BYTES;
      a[i].defer = p->red_pending; a[i].n_defer = &p->n_pending;
    }
    fl += conv_flops(L[i]->d);
  }
  Scope sc(p, st, nm, fl, by);
  sc.issued(fl);                        // (a polyphase form above may have been refused after booking its count)
  if (!defer) sc.split(nm + ".reduce", 0, a[0].ev_mid);
  return svk_wgrad_dispatch_multi(a, n, L[0]->d.dtype, svg_pick_cfg(L[0]->d.Cout), st);
}
static int run_wgrad_layer(sv_lgvae_plan* p, Layer& L, const void* x, const void* dy, float* grads, hipStream_t st) {
  Layer* Lp = &L;
  return run_wgrad_layers(p, 1, &Lp, &x, &dy, grads, st);
}

// early: side stream 0 (sv_lgvae_plan::early_stream) or nullptr -- the decoders' half of the job table goes there, the encoders' half stays in front of their forward
static int phase_prep(sv_lgvae_plan* p, const sv_lgvae_step_args* s, hipStream_t st, hipStream_t early) {
  Scope sc(p, st, "prep_weights", 0, (double)p->nparams * (4 + 2.0 * p->esz()));
  const PrepJob* jobs = (const PrepJob*)p->bp("jobs");
  if (!early) return svk_prep_weights(s->params, p->bp("warena"), p->d.dtype, jobs, (int)p->jobs.size(), p->prep_blocks, st);
  SV_TRY(svk_prep_weights(s->params, p->bp("warena"), p->d.dtype, jobs, (int)p->jobs.size(), p->prep_blocks - p->dec_block0, early, p->dec_block0));
  return svk_prep_weights(s->params, p->bp("warena"), p->d.dtype, jobs, (int)p->jobs.size(), p->dec_block0, st);
}

static int phase_forward(sv_lgvae_plan* p, const sv_lgvae_step_args* s, bool do_enc, bool do_dec, hipStream_t st,
                         bool want_nll = false) {
  p->nll_fused = false;
  const sv_lgvae_desc& d = p->d;
  const int B = d.B, H = d.H, W = d.W, dt = d.dtype;
  const int Lg = d.global_latent, Ll = d.local_latent, Lc = Lg + Ll;
  const char* en[2] = {"x", "xh"};
  static const bool no_twin = getenv("SV_NO_TWIN_POINTWISE") != nullptr;   // A/B: one launch per network for the small pointwise kernels
  bool pre_slabs = false;
  int pre_S[2] = {0, 0};
  int64_t pre_stride[2] = {0, 0};
  // (not needed once both split-K launches of this plan have gone through latent_gemm.hip's slabs: nothing accumulates into these buffers)
  if (do_enc && !(latent_gemm_on(p) && p->lat_head_ok && p->lat_d1_ok && !d.external_global_encoder)) {
    // the head pre-activations (split-K partial sums) and dz (split-K dgrad of d1) accumulate with atomics
    char* z0 = (char*)p->bp("pre_x");
    char* z1 = (char*)p->bp("gz_xh") + p->bbytes("gz_xh");
    if (hipMemsetAsync(z0, 0, (size_t)(z1 - z0), st) != hipSuccess) return (int)hipGetLastError();
    p->gz_clean = true;
    p->gz_zero_skipped = false;
  } else if (do_enc) {
    p->gz_clean = true;                        // (nothing accumulates into dz either: the backward must not re-zero it)
    p->gz_zero_skipped = true;
  }
  if (do_enc) { p->dz_slabs = false; p->dz_valid = false; }
  if (do_enc && !(s->phases & SV_PHASE_INPUTS_STAGED)) {      // (staged: sv_scramble_gather_staged filled in8_x / in8_xh for this images6)
    Scope sc(p, st, "split_pad", 0, (double)B * H * W * (24 + 16.0 * p->esz()));
    SV_TRY(svk_split_pad(s->images6, p->bp("in8_x"), p->bp("in8_xh"), dt, (int64_t)B * H * W, st));
  }
  if (do_enc) {
    static const char* in_name[3] = {"in8_", "a1_", "a2_"};
    static const char* out_name[3] = {"a1_", "a2_", "a3_"};
    for (int l = 0; l < 3; ++l) {
      Layer* Ls[2] = {&p->enc[0][l], &p->enc[1][l]};
      const void* xs[2] = {p->bp(std::string(in_name[l]) + "x"), p->bp(std::string(in_name[l]) + "xh")};
      void* ys[2] = {p->bp(std::string(out_name[l]) + "x"), p->bp(std::string(out_name[l]) + "xh")};
      const int e0 = d.external_global_encoder ? 1 : 0;     // SPLIT-GMVAE: the caller runs its own encoder_x
      SV_TRY(run_fwd_layers(p, 2 - e0, Ls + e0, xs + e0, s->params, ys + e0, st));
    }
  }
  if (do_enc) {
    // heads: split-K GEMMs into the zeroed fp32 pre-activations (bias + softplus live in reparam_kl_fwd); the two
    // networks' heads share one launch (each alone is 16 output tiles)
    const int e0 = d.external_global_encoder ? 1 : 0;
    TapGemmArgs a[2];
    int cfg = 0, cfgs[2] = {0, 0};
    double fl = 0, by = 0;
    for (int e = e0; e < 2; ++e) {
      Layer& Lh = p->enc[e][3];
      TapGemmArgs& t = a[e - e0];
      svg_fwd_args(&Lh.d, &t);
      t.A = p->bp(std::string("a3_") + en[e]);
      t.Wt = (char*)p->bp("warena") + Lh.wf_off * p->esz();
      t.bias = nullptr; t.out = p->bp(std::string("pre_") + en[e]); t.out_f32 = 1;
      cfg = svg_pick_cfg(Lh.d.Cout);
      t.splitk = svg_choose_splitk(t.M, t.N, (t.P + 7) / 8, &cfg);
      cfgs[e - e0] = cfg;
      fl += conv_flops(Lh.d);
      by += conv_bytes(Lh.d, 0, p->esz());
    }
    Scope sc(p, st, "fwd.head", fl, by);
    bool done = false;
    if (latent_gemm_on(p)) {
      NtGemmProb q[2];
      float* outs[2];
      int nq = 0;
      for (int e = e0; e < 2; ++e, ++nq) {
        Layer& Lh = p->enc[e][3];
        NtGemmProb& g = q[nq];
        memset(&g, 0, sizeof(g));
        g.f32 = p->d.dtype == SV_F32;
        g.A = p->bp(std::string("a3_") + en[e]); g.lda = Lh.d.Cin;
        g.W = (char*)p->bp("warena") + Lh.wf_off * p->esz(); g.ldw = Lh.d.Cin;
        g.out = p->bp(std::string("lat_ws_") + en[e]); g.ldo = Lh.d.Cout;
        g.M = B; g.N = Lh.d.Cout; g.K = Lh.d.Cin; g.out_f32 = 1;
        g.splitk = svk_nt_gemm_pick_splitk(B, g.N, g.K, 2);     // (as the slab buffers were sized)
        g.slab_stride = (int64_t)B * g.ldo;
        outs[nq] = (float*)p->bp(std::string("pre_") + en[e]);
        if ((int64_t)g.splitk * g.slab_stride * 4 > p->bbytes(std::string("lat_ws_") + en[e])) return SV_E_WORKSPACE;   // (cannot happen: sized from the same calls)
      }
      const int rc = svk_nt_gemm_multi(q, nq, 64, st);
      if (rc == SV_OK) {
        done = true;
        p->lat_head_ok = true;
        // Sampling + KL sum the slabs themselves, in slice order: the same bits (more than 16 slices: one wave per row reads them too slowly, B = 64: +4 %)
        if (latent_fuse_on() && !e0 && !no_twin && q[0].splitk <= 16 && q[1].splitk <= 16) {
          pre_slabs = true;
          for (int e = 0; e < 2; ++e) { pre_S[e] = q[e].splitk; pre_stride[e] = q[e].slab_stride; }
        } else SV_TRY(svk_nt_slab_reduce(q, outs, nq, st));
      } else if (rc != SV_E_UNSUPPORTED) return rc;
    }
    if (done) {}
    else if (e0 || cfgs[0] == cfgs[1]) SV_TRY(svk_tap_gemm_multi(a, 2 - e0, dt, cfgs[0], st));
    else
      for (int e = 0; e < 2; ++e) SV_TRY(svk_tap_gemm(a[e], dt, cfgs[e], st));   // latent sizes with different tiles
  }
  if (do_enc && !d.external_global_encoder && !no_twin) {
    // both networks' Sampling + KL in one launch
    Scope sc(p, st, "reparam_kl_fwd", 0, (double)B * (Lg + Ll) * 16);
    const float *pre[2], *bm[2], *bs[2], *eps[2];
    float *eo[2], *zm[2], *zs[2], *zz[2], *kl[2];
    const int zc[2] = {0, Lg}, LL[2] = {Lg, Ll};
    for (int e = 0; e < 2; ++e) {
      const std::string sfx = en[e];
      const Layer& Lh = p->enc[e][3];
      pre[e] = (const float*)p->bp((pre_slabs ? "lat_ws_" : "pre_") + sfx);
      bm[e] = s->params + p->params[Lh.kparam + 1].off; bs[e] = s->params + p->params[Lh.kparam + 3].off;
      eps[e] = e == 0 ? s->eps_x : s->eps_x_hat;
      eo[e] = (float*)p->bp("eps_" + sfx); zm[e] = (float*)p->bp("z_mean_" + sfx); zs[e] = (float*)p->bp("z_sig_" + sfx);
      zz[e] = (float*)p->bp("z_" + sfx); kl[e] = (float*)p->bp("kl_" + sfx);
    }
    SV_TRY(svk_reparam_kl_fwd_twin(pre, bm, bs, eps, eo, zm, zs, zz, p->bp("zcat"), dt, Lc, zc, kl, B, LL, s->seed, s->step,
                                   s->sample_offset, st, p->dyn, pre_slabs ? pre_S : nullptr, pre_stride));
  } else
  for (int e = d.external_global_encoder ? 1 : 0; e < 2 && do_enc; ++e) {
    const std::string sfx = en[e];
    const int L = e == 0 ? Lg : Ll;
    {
      Scope sc(p, st, "reparam_kl_fwd", 0, (double)B * L * 16);
      const Layer& Lh = p->enc[e][3];
      // mean bias and sd bias are separate Keras tensors: two-pointer form of sv_reparam_kl_fwd
      const float* eps = e == 0 ? s->eps_x : s->eps_x_hat;
      SV_TRY(svk_reparam_kl_fwd2((const float*)p->bp("pre_" + sfx), s->params + p->params[Lh.kparam + 1].off,
                                 s->params + p->params[Lh.kparam + 3].off, eps, (float*)p->bp("eps_" + sfx),
                                 (float*)p->bp("z_mean_" + sfx), (float*)p->bp("z_sig_" + sfx),
                                 (float*)p->bp("z_" + sfx), p->bp("zcat"), dt, Lc, e == 0 ? 0 : Lg,
                                 (float*)p->bp("kl_" + sfx), B, L, s->seed, s->step, e, s->sample_offset, st, p->dyn));
    }
  }
  if (do_dec) {
    SV_TRY(p->early_wait(st));               // the decoders' weight images (and the zeroed gradient buffer) of this call's early side work
    const void* zin[2] = {p->bp("zcat"), (const char*)p->bp("zcat") + (size_t)Lg * p->esz()};
    {   // d1 differs between the twins (zcat vs local-only input): two shapes, one launch
      Layer* Ls[2] = {&p->dec[0][0], &p->dec[1][0]};
      void* ys[2] = {p->bp("h1_x"), p->bp("h1_xh")};
      SV_TRY(run_fwd_layers(p, 2, Ls, zin, s->params, ys, st));
    }
    {
      Layer* Ls[2] = {&p->dec[0][1], &p->dec[1][1]};
      const void* xs[2] = {p->bp("h1_x"), p->bp("h1_xh")};
      void* ys[2] = {p->bp("h2_x"), p->bp("h2_xh")};
      SV_TRY(run_fwd_layers(p, 2, Ls, xs, s->params, ys, st));
    }
    // d3..d5 consume the 2x bilinear upsample of the previous activation (vae/model.py:163-167).
    // Fused (bf16): the conv/wgrad tile staging interpolates from the low-res tensor, u2/u3/u4 are
    // never written.  Materialised (fp32 parity path, whose wgrad runs on the im2col kernel).
    static const char* lo_name[3] = {"h2_", "h3_", "h4_"};
    static const char* hi_name[3] = {"u2_", "u3_", "u4_"};
    static const char* out_name[3] = {"h3_", "h4_", "out6_"};
    for (int l = 0; l < 3; ++l) {
      Layer* Ls[2] = {&p->dec[0][2 + l], &p->dec[1][2 + l]};
      const void* xs[2];
      void* ys[2];
      for (int k = 0; k < 2; ++k) {
        const std::string sfx = en[k];
        xs[k] = p->bp(lo_name[l] + sfx);
        ys[k] = p->bp(out_name[l] + sfx);
        if (!Ls[k]->d.ups_in) {
          Scope sc(p, st, "upsample_fwd", 0, 0);
          SV_TRY(sv_upsample2x_fwd(xs[k], p->bp(hi_name[l] + sfx), dt, B, Ls[k]->d.H / 2, Ls[k]->d.W / 2, Ls[k]->d.Cin, st));
          xs[k] = p->bp(hi_name[l] + sfx);
        }
      }
      if (l == 2 && want_nll && svg_poly(&Ls[0]->d) && d.H * d.W >= 1024) {
        // training step: the head's epilogue evaluates the loss of its own pixels (NLL partial sums + the gradient record
        // g5), so dlogistic_kernel and its re-read of out6 drop out of the step (phase_loss then only sums the partials)
        FusedNll f;
        f.images6 = s->images6; f.gscale = 1.0f / (float)d.B; f.noout = (s->phases & SV_PHASE_NO_RECON) ? 1 : 0;
        f.grad[0] = p->bp("g5_x"); f.grad[1] = p->bp("g5_xh");
        f.part[0] = (float*)p->bp("nllpart_x"); f.part[1] = (float*)p->bp("nllpart_xh");
        const int rc = run_fwd_layers(p, 2, Ls, xs, s->params, ys, st, &f);
        if (rc == SV_OK) { p->nll_fused = true; continue; }
        if (rc != SV_E_UNSUPPORTED) return rc;
      }
      SV_TRY(run_fwd_layers(p, 2, Ls, xs, s->params, ys, st));
    }
  }
  return SV_OK;
}

static int phase_loss(sv_lgvae_plan* p, const sv_lgvae_step_args* s, bool with_grad, hipStream_t st) {
  const sv_lgvae_desc& d = p->d;
  const char* en[2] = {"x", "xh"};
  auto zs0 = [&](const char* kind, int64_t esz) {
    return (int64_t)((char*)p->bp(std::string(kind) + "xh") - (char*)p->bp(std::string(kind) + "x")) / esz;
  };
  const bool from_parts = p->nll_fused && with_grad;   // the head's epilogue left per-tile NLL sums and g5 (phase_forward): finalize_losses
                                                       // forms the per-image sums itself (fixed order), no separate kernel
  if (from_parts) {
  } else {
    // algorithmic bytes: read x, m, log_scale (12 B/element) + write dm, dls (2 * esz B/element); both networks
    Scope sc(p, st, "dlogistic_nll", 0, 2.0 * d.B * d.H * d.W * 3 * (12.0 + (with_grad ? 2.0 * p->esz() : 0.0)));
    auto zs = [&](const char* kind, int64_t esz) {
      return (int64_t)((char*)p->bp(std::string(kind) + "xh") - (char*)p->bp(std::string(kind) + "x")) / esz;
    };
    SV_TRY(svk_dlogistic_nll_multi(s->images6, 0, (const float*)p->bp("out6_x"), zs("out6_", 4), (float*)p->bp("nll_x"),
                                   zs("nll_", 4), with_grad ? p->bp("g5_x") : nullptr, zs("g5_", (int64_t)p->esz()), d.dtype,
                                   1.0f / (float)d.B, d.B, d.H, d.W, (float*)p->bp("nllpart_x"), zs("nllpart_", 4), 2, st));
  }
  const int HWp = d.H * d.W;
  SV_TRY(svk_finalize_losses((const float*)p->bp("nll_x"), (const float*)p->bp("nll_xh"), (const float*)p->bp("kl_x"),
                             (const float*)p->bp("kl_xh"), d.B, d.beta, (float*)p->bp("losses"),
                             (float*)p->bp("metric_acc"), s->accumulate_metrics, st,
                             from_parts ? (const float*)p->bp("nllpart_x") : nullptr,
                             from_parts ? (const float*)p->bp("nllpart_xh") : nullptr, HWp > 1024 ? HWp / 1024 : 1));
  return SV_OK;
}

static int phase_bwd_decoders(sv_lgvae_plan* p, const sv_lgvae_step_args* s, hipStream_t st) {
  if (!p->gz_clean) {   // a second backward over the same forward: re-zero the split-K dz accumulators
    char* z0 = (char*)p->bp("gz_x");
    char* z1 = (char*)p->bp("gz_xh") + p->bbytes("gz_xh");
    if (hipMemsetAsync(z0, 0, (size_t)(z1 - z0), st) != hipSuccess) return (int)hipGetLastError();
  }
  p->gz_clean = false;
  const sv_lgvae_desc& d = p->d;
  const int B = d.B, H = d.H, W = d.W, dt = d.dtype;
  const int Lg = d.global_latent;
  const char* en[2] = {"x", "xh"};
  auto both = [&](const char* n, const void** out) { out[0] = p->bp(std::string(n) + "x"); out[1] = p->bp(std::string(n) + "xh"); };
  const void* none[2] = {nullptr, nullptr};
  static const char* gy_name[3] = {"g5_", "g4_", "g3_"};       // gradient at the layer output
  static const char* gu_name[3] = {"gu4_", "gu3_", "gu2_"};    // gradient at the (virtual) upsampled input
  static const char* lo_name[3] = {"h4_", "h3_", "h2_"};       // low-res activation feeding the upsample
  static const char* hi_name[3] = {"u4_", "u3_", "u2_"};
  static const char* gl_name[3] = {"g4_", "g3_", "g2_"};       // gradient at the low-res activation
  for (int l = 0; l < 3; ++l) {                                // d5, d4, d3
    const int li = 4 - l;
    Layer* Ls[2] = {&p->dec[0][li], &p->dec[1][li]};
    const void *gy[2], *gu[2];
    both(gy_name[l], gy); both(gu_name[l], gu);
    const void* xin[2];
    for (int k = 0; k < 2; ++k) xin[k] = p->bp(std::string(Ls[k]->d.ups_in ? lo_name[l] : hi_name[l]) + en[k]);
    SV_TRY(run_wgrad_layers(p, 2, Ls, xin, gy, s->grads, st));
    const void *lo[2], *gl[2];
    both(lo_name[l], lo); both(gl_name[l], gl);
    // the input gradient lands at the LOW-RES activation in one launch where the fused kernel exists (ResizeBilinearGrad +
    // ReluGrad in the epilogue of Conv2DBackpropInput: the hi-res gradient gu never reaches HBM) ...
    // (it works on whole images -- its low-res rows straddle row bands -- so below ~one image per workgroup slot the banded
    // plain input gradient + upsample2x_bwd is faster: 64 images per network d5 43 vs 55 us, d4 40 vs 49; 128: 69 vs 56)
    static const bool no_adj = getenv("SV_NO_FUSED_ADJOINT") != nullptr;
    // (round 4, with the adjoint of d3 / d4 on the matrix pipe: 128 images per launch -- config 4's 64-image shard -- 0.641 -> 0.631 ms fused; the
    //  default moved from 256 to 128)
    static const int adj_min = getenv("SV_RC_ADJ_MIN") ? atoi(getenv("SV_RC_ADJ_MIN")) : 128;
    int frc = SV_E_UNSUPPORTED;
    if (Ls[0]->wdp_off >= 0 && Ls[1]->wdp_off >= 0) {
      // fp32: the polyphase form (polyd_dgrad.hip): conv-transpose + resize adjoint + ReLU gate as one stride-2 conv over dY, edge terms through a workspace
      const void* wp[2];
      void* ews[2] = {p->bp("polyd_x"), p->bp("polyd_xh")};
      double fl = 0, by = 0;
      for (int k = 0; k < 2; ++k) {
        wp[k] = (const char*)p->bp("warena") + Ls[k]->wdp_off * p->esz();
        fl += conv_flops(Ls[k]->d); by += conv_bytes(Ls[k]->d, 1, p->esz());
      }
      Scope sc(p, st, "dgrad." + Ls[0]->name.substr(Ls[0]->name.find('.') + 1), fl, by);
      { double is = 0; for (int k = 0; k < 2; ++k) is += poly_issued(Ls[k]->d, 81, true); sc.issued(is); }
      frc = svk_polyd_dgrad_multi(&Ls[0]->d, 2, gy, wp, lo, (void* const*)gl, ews, st);
    }
    if (frc == SV_E_UNSUPPORTED && Ls[0]->d.ups_in && Ls[1]->d.ups_in && !no_adj && 2 * B >= adj_min) frc = run_dgrad_layers(p, 2, Ls, gy, lo, (void* const*)gl, false, st, true);
    if (frc != SV_E_UNSUPPORTED) { SV_TRY(frc); continue; }
    // ... else the hi-res gradient goes through HBM and a stand-alone adjoint pass
    SV_TRY(run_dgrad_layers(p, 2, Ls, gy, none, (void* const*)gu, false, st));
    {
      const int h = (H / 2) >> l, w = (W / 2) >> l, c = 32 << l;
      const int64_t lo_bytes = (int64_t)B * h * w * c * p->esz();
      const bool contig = (const char*)gu[1] == (const char*)gu[0] + 4 * lo_bytes &&
                          (const char*)lo[1] == (const char*)lo[0] + lo_bytes && (const char*)gl[1] == (const char*)gl[0] + lo_bytes;
      Scope sc(p, st, std::string("upsample_bwd.") + lo_name[l][1], 0, 2.0 * 6 * lo_bytes);   // read g_hi (4x) + mask, write g_lo; both networks
      if (contig) SV_TRY(sv_upsample2x_bwd(gu[0], lo[0], (void*)gl[0], dt, 2 * B, h, w, c, st));
      else
        for (int k = 0; k < 2; ++k) SV_TRY(sv_upsample2x_bwd(gu[k], lo[k], (void*)gl[k], dt, B, h, w, c, st));
    }
  }
  {
    // d2 (input h1 = relu(d1): mask fused in the dgrad epilogue)
    Layer* Ls[2] = {&p->dec[0][1], &p->dec[1][1]};
    const void *g2[2], *h1[2], *g1[2];
    both("g2_", g2); both("h1_", h1); both("g1_", g1);
    SV_TRY(run_wgrad_layers(p, 2, Ls, h1, g2, s->grads, st));
    SV_TRY(run_dgrad_layers(p, 2, Ls, g2, h1, (void* const*)g1, false, st));
  }
  {
    // d1 (dense): dz accumulated in fp32 over split K; the twins (different input widths) share the launches
    Layer Ld[2] = {p->dec[0][0], p->dec[1][0]};
    Layer* Ls[2] = {&p->dec[0][0], &p->dec[1][0]};
    Layer* Lds[2] = {&Ld[0], &Ld[1]};
    const void* zin[2] = {p->bp("zcat"), (const char*)p->bp("zcat") + (size_t)Lg * p->esz()};
    const void *g1[2], *none2[2] = {nullptr, nullptr};
    void* gz[2] = {p->bp("gz_x"), p->bp("gz_xh")};
    both("g1_", g1);
    SV_TRY(run_wgrad_layers(p, 2, Ls, zin, g1, s->grads, st));
    for (int k = 0; k < 2; ++k) Ld[k].d.ldx = Ld[k].d.Cin;   // dz has its own row pitch (Lz), not the zcat pitch
    bool done = false;
    p->dz_slabs = false;
    p->dz_valid = true;
    if (latent_gemm_on(p)) {
      NtGemmProb q[2];
      float* outs[2];
      double fl = 0, by = 0;
      for (int k = 0; k < 2; ++k) {
        const sv_conv_desc& dd = Ld[k].d;
        NtGemmProb& g = q[k];
        memset(&g, 0, sizeof(g));
        g.f32 = p->d.dtype == SV_F32;
        g.A = g1[k]; g.lda = dd.ldy;
        g.W = (char*)p->bp("warena") + Ld[k].wd_off[0] * p->esz(); g.ldw = svg_gdy(&dd);
        g.out = p->bp(k == 0 ? "lat_ws_x" : "lat_ws_xh"); g.ldo = dd.Cin;
        g.M = B; g.N = dd.Cin; g.K = dd.Cout; g.out_f32 = 1;
        g.splitk = svk_nt_gemm_pick_splitk(B, g.N, g.K, 2);
        g.slab_stride = (int64_t)B * g.ldo;
        if ((int64_t)g.splitk * g.slab_stride * 4 > p->bbytes(k == 0 ? "lat_ws_x" : "lat_ws_xh")) return SV_E_WORKSPACE;           // (cannot happen: sized from the same calls)
        outs[k] = (float*)gz[k];
        fl += conv_flops(dd); by += conv_bytes(dd, 1, p->esz());
      }
      Scope sc(p, st, "dgrad.d1", fl, by);
      const int rc = svk_nt_gemm_multi(q, 2, 64, st);
      if (rc == SV_OK) {
        done = true;
        p->lat_d1_ok = true;
        static const bool no_twin = getenv("SV_NO_TWIN_POINTWISE") != nullptr;
        if (latent_fuse_on() && !d.external_global_encoder && !no_twin && q[0].splitk <= 16 && q[1].splitk <= 16) {     // reparam_kl_bwd sums the slabs (SPLIT-GMVAE reads gz_x: summed here)
          p->dz_slabs = true;
          for (int k = 0; k < 2; ++k) { p->dz_S[k] = q[k].splitk; p->dz_stride[k] = q[k].slab_stride; }
        } else SV_TRY(svk_nt_slab_reduce(q, outs, 2, st));
      } else if (rc != SV_E_UNSUPPORTED) return rc;
    }
    if (!done) SV_TRY(run_dgrad_layers(p, 2, Lds, g1, none2, gz, true, st));
  }
  return SV_OK;
}

static int phase_bwd_encoders(sv_lgvae_plan* p, const sv_lgvae_step_args* s, bool do_heads, bool do_convs, hipStream_t st, bool buckets = false) {
  const sv_lgvae_desc& d = p->d;
  const int B = d.B, dt = d.dtype;
  const int Lg = d.global_latent, Ll = d.local_latent, Lc = Lg + Ll;
  const char* en[2] = {"x", "xh"};
  const float kl_scale = d.beta / (float)B;
  const int e0 = d.external_global_encoder ? 1 : 0;
  if (do_heads && !p->dz_valid && p->gz_zero_skipped) {       // KL terms only (no decoder backward since the forward): dz = 0
    char* z0 = (char*)p->bp("gz_x");
    char* z1 = (char*)p->bp("gz_xh") + p->bbytes("gz_xh");
    if (hipMemsetAsync(z0, 0, (size_t)(z1 - z0), st) != hipSuccess) return (int)hipGetLastError();
    p->gz_zero_skipped = false;
  }
  if (do_heads) {
    Scope sc(p, st, "reparam_kl_bwd", 0, 0);
    static const bool no_twin = getenv("SV_NO_TWIN_POINTWISE") != nullptr;
    if (!e0 && !no_twin) {
      const float* dz[2] = {(const float*)p->bp("gz_x"), (const float*)p->bp("gz_x") + Lg};
      const float* dz2[2] = {nullptr, (const float*)p->bp("gz_xh")};
      const int ld[2] = {Lc, Lc}, ld2[2] = {0, Ll}, LL[2] = {Lg, Ll};
      const float* zm[2] = {(const float*)p->bp("z_mean_x"), (const float*)p->bp("z_mean_xh")};
      const float* zs[2] = {(const float*)p->bp("z_sig_x"), (const float*)p->bp("z_sig_xh")};
      const float* ep[2] = {(const float*)p->bp("eps_x"), (const float*)p->bp("eps_xh")};
      void* gp[2] = {p->bp("ghead_x"), p->bp("ghead_xh")};
      if (p->dz_slabs) {
        const float *sx = (const float*)p->bp("lat_ws_x"), *sxh = (const float*)p->bp("lat_ws_xh");
        const float* dzs[2] = {sx, sx + Lg};
        const float* dz2s[2] = {nullptr, sxh};
        const int S[2] = {p->dz_S[0], p->dz_S[0]}, S2[2] = {0, p->dz_S[1]};
        const int64_t sd[2] = {p->dz_stride[0], p->dz_stride[0]}, sd2[2] = {0, p->dz_stride[1]};
        SV_TRY(svk_reparam_kl_bwd_twin(dzs, ld, dz2s, ld2, zm, zs, ep, kl_scale, gp, dt, B, LL, st, S, sd, S2, sd2));
      } else
      SV_TRY(svk_reparam_kl_bwd_twin(dz, ld, dz2, ld2, zm, zs, ep, kl_scale, gp, dt, B, LL, st));
    } else {
    if (!e0)
      SV_TRY(sv_reparam_kl_bwd((const float*)p->bp("gz_x"), Lc, nullptr, 0, (const float*)p->bp("z_mean_x"),
                               (const float*)p->bp("z_sig_x"), (const float*)p->bp("eps_x"), kl_scale, p->bp("ghead_x"),
                               dt, B, Lg, st));
    SV_TRY(sv_reparam_kl_bwd((const float*)p->bp("gz_x") + Lg, Lc, (const float*)p->bp("gz_xh"), Ll,
                             (const float*)p->bp("z_mean_xh"), (const float*)p->bp("z_sig_xh"),
                             (const float*)p->bp("eps_xh"), kl_scale, p->bp("ghead_xh"), dt, B, Ll, st));
    }
  }
  if (do_heads) {
    // heads: two Keras kernels/biases per network -> wgrad problems on the column halves of ghead, all in one launch
    WgradArgs a[4];
    int n = 0;
    double fl = 0, by = 0;
    for (int e = e0; e < 2; ++e) by += conv_bytes(p->enc[e][3].d, 2, p->esz());
    for (int e = e0; e < 2; ++e) {
      Layer* L = p->enc[e];
      const int Lh = e == 0 ? Lg : Ll;
      for (int h = 0; h < 2; ++h, ++n) {
        svg_wgrad_args(&L[3].d, &a[n]);
        a[n].A = p->bp(std::string("a3_") + en[e]);
        a[n].dY = (const char*)p->bp(std::string("ghead_") + en[e]) + (size_t)h * Lh * p->esz();
        a[n].ycols = Lh; a[n].N = Lh;
        a[n].dW = s->grads + p->params[L[3].kparam + 2 * h].off;
        a[n].dbias = s->grads + p->params[L[3].kparam + 2 * h + 1].off;
        fl += conv_flops(L[3].d) / 2;
      }
    }
    {
      hipStream_t ws = p->wgrad_stream(st);
      Scope sc(p, ws, "wgrad.head", fl, by);
      bool done = false;
      if (latent_gemm_on(p) && n <= 4) {
        TnWgradProb q[4];
        bool ok = true;
        for (int i = 0; i < n; ++i) {
          q[i] = TnWgradProb{a[i].A, a[i].lda, a[i].dY, a[i].ldy, a[i].dW, a[i].dbias, B, a[i].Cin_pad, a[i].Cin_real, a[i].N, dt == SV_F32};
          ok = ok && svk_tn_wgrad_supported(q[i]);
        }
        if (ok) { SV_TRY(svk_tn_wgrad_multi(q, n, ws)); done = true; }
      }
      const int cg = svg_pick_cfg(Lg), cl = svg_pick_cfg(Ll);      // the narrower tile serves both widths
      if (!done) SV_TRY(svk_wgrad_dispatch_multi(a, n, dt, e0 ? cl : (cg > cl ? cg : cl), ws));
    }
    Layer* Ls[2] = {&p->enc[0][3], &p->enc[1][3]};
    const void *gh[2] = {p->bp("ghead_x"), p->bp("ghead_xh")}, *a3[2] = {p->bp("a3_x"), p->bp("a3_xh")};
    void* ga3[2] = {p->bp("ga3_x"), p->bp("ga3_xh")};
    bool done = false;
    if (latent_gemm_on(p)) {
      NtGemmProb q[2];
      int nq = 0;
      double fl = 0, by = 0;
      for (int e = e0; e < 2; ++e, ++nq) {
        Layer& Lh = p->enc[e][3];
        NtGemmProb& g = q[nq];
        memset(&g, 0, sizeof(g));
        g.f32 = p->d.dtype == SV_F32;
        g.A = gh[e]; g.lda = Lh.d.Cout;
        g.W = (char*)p->bp("warena") + Lh.wd_off[0] * p->esz(); g.ldw = Lh.d.Cout;
        g.out = ga3[e]; g.ldo = Lh.d.Cin; g.mask = a3[e];
        g.M = B; g.N = Lh.d.Cin; g.K = Lh.d.Cout; g.splitk = 1;
        fl += conv_flops(Lh.d); by += conv_bytes(Lh.d, 1, p->esz());
      }
      Scope sc(p, st, "dgrad.head", fl, by);
      const int rc = svk_nt_gemm_multi(q, nq, B >= 256 ? 128 : 64, st);
      if (rc == SV_OK) done = true;
      else if (rc != SV_E_UNSUPPORTED) return rc;
    }
    if (!done) SV_TRY(run_dgrad_layers(p, 2 - e0, Ls + e0, gh + e0, a3 + e0, ga3 + e0, false, st));
    if (buckets) SV_TRY(p->record_bucket(1, st));       // the heads' weight gradients are enqueued (main or side stream)
  }
  if (do_convs) {
    auto both = [&](const char* n, const void** out) { out[0] = p->bp(std::string(n) + "x"); out[1] = p->bp(std::string(n) + "xh"); };
    static const char* act_name[3] = {"in8_", "a1_", "a2_"};
    static const char* g_name[3] = {"ga1_", "ga2_", "ga3_"};
    for (int l = 2; l >= 0; --l) {
      Layer* Ls[2] = {&p->enc[0][l], &p->enc[1][l]};
      const void *x[2], *gy[2], *gx[2];
      both(act_name[l], x); both(g_name[l], gy);
      SV_TRY(run_wgrad_layers(p, 2 - e0, Ls + e0, x + e0, gy + e0, s->grads, st));
      if (l == 0) break;
      both(g_name[l - 1], gx);
      SV_TRY(run_dgrad_layers(p, 2 - e0, Ls + e0, gy + e0, x + e0, (void* const*)gx + e0, false, st));
    }
    if (buckets) SV_TRY(p->record_bucket(2, st));
  }
  return SV_OK;
}

}  // namespace

extern "C" int64_t sv_lgvae_param_count(const sv_lgvae_desc* d) {
  if (check_desc(d) != SV_OK) return -1;
  auto v = build_params(d);
  return v.back().off + (v.back().count + 3) / 4 * 4;
}

extern "C" int sv_lgvae_param_info(const sv_lgvae_desc* d, int32_t index, int64_t* offset, int32_t* ndim,
                                   int64_t shape[4], char name[96]) {
  int rc = check_desc(d);
  if (rc) return rc;
  auto v = build_params(d);
  if (index < 0 || index >= (int)v.size()) return SV_E_BADARG;
  const ParamInfo& p = v[index];
  if (offset) *offset = p.off;
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  if (name) snprintf(name, 96, "%s", p.name.c_str());
  return SV_OK;
}

extern "C" int sv_lgvae_plan_create(const sv_lgvae_desc* d, sv_lgvae_plan** out) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!out) return SV_E_BADARG;
  sv_lgvae_plan* p = new sv_lgvae_plan();
  p->d = *d;
  p->params = build_params(d);
  p->nparams = p->params.back().off + (p->params.back().count + 3) / 4 * 4;
  p->ws = nullptr; p->bound = false; p->prof_on = false;
  build_layers(p);
  for (int e = 0; e < 2; ++e)
    for (int l = 0; l < 4; ++l)
      if ((rc = svg_check(&p->enc[e][l].d))) { delete p; return rc; }
  for (int k = 0; k < 2; ++k)
    for (int l = 0; l < 5; ++l)
      if ((rc = svg_check(&p->dec[k][l].d))) { delete p; return rc; }
  build_prep_jobs(p);
  build_buffers(p);
  *out = p;
  return SV_OK;
}

extern "C" void sv_lgvae_plan_destroy(sv_lgvae_plan* p) {
  if (!p) return;
  for (auto& pe : p->pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
  for (auto e : p->event_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < p->nside; ++i) {
    (void)hipStreamSynchronize(p->side[i]);          // (shared with every other plan / tape of the process: not destroyed)
    (void)hipEventDestroy(p->ev_join[i]);
  }
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_early) (void)hipEventDestroy(p->ev_early);
  for (auto e : p->ev_cap)
    if (e) (void)hipEventDestroy(e);
  for (auto& row : p->ev_bucket)
    for (auto e : row)
      if (e) (void)hipEventDestroy(e);
  if (!p->graphs.empty()) (void)hipDeviceSynchronize();   // a replay may still be in flight
  for (auto& kv : p->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  delete p;
}

extern "C" int64_t sv_lgvae_workspace_bytes(const sv_lgvae_plan* p) { return p ? p->ws_bytes : -1; }

extern "C" int sv_lgvae_plan_bind(sv_lgvae_plan* p, void* workspace, int64_t bytes, void* stream) {
  if (!p || !workspace) return SV_E_BADARG;
  if (bytes < p->ws_bytes) return SV_E_WORKSPACE;
  if ((uintptr_t)workspace & 255) return SV_E_BADARG;
  p->ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  // The whole workspace starts from ZERO: pad channels / pad rows of the activation, gradient and weight-image buffers that no kernel ever writes are read as
  // zeros by the MFMA kernels, and a few accumulators (metric_acc, the polyphase head's dbias') count up from zero.  The Python mirror used to hand in a zeroed
  // tensor; a C caller's hipMalloc'd block is garbage (scripts/ws_poison_probe.py: with 0xFF bytes every loss was NaN), so the bind does it itself.
  hipError_t e = hipMemsetAsync(workspace, 0, (size_t)p->ws_bytes, st);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(p->bp("jobs"), p->jobs.data(), p->jobs.size() * sizeof(PrepJob), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize(st);   // the job table lives in plan-owned host memory: finish the copy now
  if (e != hipSuccess) return (int)e;
  p->bound = true;
  return SV_OK;
}

extern "C" int sv_lgvae_buffer(const sv_lgvae_plan* p, const char* name, int64_t* offset, int64_t* bytes) {
  if (!p || !name) return SV_E_BADARG;
  auto it = p->bufidx.find(name);
  if (it == p->bufidx.end()) return SV_E_BADARG;
  if (offset) *offset = p->bufs[it->second].off;
  if (bytes) *bytes = p->bufs[it->second].bytes;
  return SV_OK;
}

static int run_phases(sv_lgvae_plan* p, const sv_lgvae_step_args* s, hipStream_t st) {
  const int ph = s->phases;
  p->n_cap = 0;                                  // this call's fork / join events and the side streams it uses
  for (auto& u : p->side_used) u = false;
  {
    // Two weight-gradient side streams for a whole step at >= 768 images per launch, one otherwise.  Re-measured in round 3 (round 2: +-0):
    // B = 512 2.056 -> 2.005 ms (three launches in flight fill the CUs the rolling-window d4 kernel and the row-ring input gradients leave);
    // B = 256 +1.8 %, 128 +6 %, 64 +3 % (launches too small to share the chip three ways); a data-parallel step (phase-split calls: the
    // communication stream is a further active queue) 2.14 -> 2.15 ms over torch's nccl, 2.12 -> 3.31 ms over sv_comm (four active streams
    // on GPU_MAX_HW_QUEUES = 3): one.  SV_SIDE_STREAMS forces a count.
    static const int forced = getenv("SV_SIDE_STREAMS") ? atoi(getenv("SV_SIDE_STREAMS")) : 0;
    const bool whole = (ph & SV_PHASE_ALL) == SV_PHASE_ALL;
    // fp32 (round 5, polyphase decoder layers + 52-KB weight-gradient tiles): two side streams 9.60-9.62 -> 9.37-9.39 ms at B = 512 (profiles/r05_f32_streams.txt)
    // (fp32, 256 images per network: 5.14 -> 5.09 ms; 128: 2.84 -> 2.87: from 256)
    // (round 6, with the process's fixed side streams: bf16 256 images per network 1.068-1.080 -> 1.054-1.056 ms with two: the bf16 threshold moved from 768 to 512 too;
    //  profiles/r06_sweep_bf16.txt)
    p->side_use = forced > 0 ? forced : (whole && 2 * p->d.B >= 512) ? 2 : 1;
  }
  // early side work: the decoders' weight images and the gradient buffer's zero fill on side stream 0 beside the encoders' forward (see early_stream)
  const bool zero_here = (ph & SV_PHASE_LOSS) && s->grads;
  hipStream_t early = ((ph & SV_PHASE_PREP) && (ph & SV_PHASE_FWD_ENCODERS)) ? p->early_stream(st) : nullptr;
  if (early && zero_here) {
    if (hipMemsetAsync(s->grads, 0, (size_t)p->nparams * 4, early) != hipSuccess) return (int)hipGetLastError();
  }
  if (ph & SV_PHASE_PREP) SV_TRY(phase_prep(p, s, st, early));
  if (early) SV_TRY(p->early_done(early));
  static const bool no_fused_nll = getenv("SV_NO_FUSED_NLL") != nullptr;    // A/B: dlogistic_kernel after the forward
  static const bool no_fused_nll_f32 = getenv("SV_NO_FUSED_NLL_F32") != nullptr;    // (fp32 since round 6; A/B)
  const bool want_nll = !no_fused_nll && (ph & SV_PHASE_FWD_DECODERS) && (ph & SV_PHASE_LOSS) && s->grads && s->images6 &&
                        (p->d.dtype == SV_BF16 || !no_fused_nll_f32);
  if (!(ph & SV_PHASE_FORWARD)) p->nll_fused = false;
  if (ph & SV_PHASE_FORWARD) SV_TRY(phase_forward(p, s, ph & SV_PHASE_FWD_ENCODERS, ph & SV_PHASE_FWD_DECODERS, st, want_nll));
  if (ph & SV_PHASE_LOSS) {
    if (s->grads && !early) {
      Scope sc(p, st, "zero_grads", 0, (double)p->nparams * 4);
      if (hipMemsetAsync(s->grads, 0, (size_t)p->nparams * 4, st) != hipSuccess) return (int)hipGetLastError();
    }
    SV_TRY(phase_loss(p, s, s->grads != nullptr, st));
  }
  SV_TRY(p->early_wait(st));                 // (a call without the decoders' forward: before any gradient is written, at the latest here)
  const bool buckets = (ph & SV_PHASE_BUCKET_EVENTS) && !p->graph_on && !p->dyn;      // (not in captured steps: the events belong to eager launches)
  // a backward phase invalidates the events of every earlier step: sv_lgvae_bucket_wait must never succeed against events that belong to gradients
  // of a previous step (it returns SV_E_STATE instead, and the trainer falls back to ordering the collective behind the compute stream)
  if (ph & SV_PHASE_BACKWARD) p->clear_buckets();
  p->buckets_now = buckets;
  if (ph & SV_PHASE_BWD_DECODERS) {
    SV_TRY(phase_bwd_decoders(p, s, st));
    if (buckets) SV_TRY(p->record_bucket(0, st));
  }
  if (ph & (SV_PHASE_BWD_ENC_HEADS | SV_PHASE_BWD_ENC_CONVS))
    SV_TRY(phase_bwd_encoders(p, s, ph & SV_PHASE_BWD_ENC_HEADS, ph & SV_PHASE_BWD_ENC_CONVS, st, buckets));
  SV_TRY(p->join_side(st));
  if (p->n_pending) {                    // every layer's partial sums -> dW / dbias, one launch (fixed order: deterministic)
    Scope sc(p, st, "wgrad.all.reduce", 0, 0);
    const int np = p->n_pending;
    p->n_pending = 0;
    SV_TRY(svk_wgrad_reduce_all(p->red_pending, np, st));
  }
  if (ph & SV_PHASE_ADAM) {
    Scope sc(p, st, "adam_step", 0, (double)p->nparams * 28);
    SV_TRY(svk_adam_step(s->params, s->grads, s->adam_m, s->adam_v, p->nparams, s->lr, s->beta1, s->beta2,
                         s->adam_eps, s->t, s->grad_scale, p->dyn, st));
  }
  return SV_OK;
}

// everything of a step that is baked into a captured graph: the phase mask, every buffer, every scalar that reaches a
// kernel as a launch argument.  seed / step / sample_offset / (lr, t) do not belong here: they travel through SvDynArgs.
static std::vector<uint64_t> graph_key(const sv_lgvae_plan* p, const sv_lgvae_step_args* s, hipStream_t st) {
  auto f = [](float x) { uint32_t u; memcpy(&u, &x, 4); return (uint64_t)u; };
  const bool adam = s->phases & SV_PHASE_ADAM;
  return {(uint64_t)(uint32_t)s->phases, (uint64_t)s->params, (uint64_t)s->grads, (uint64_t)s->adam_m, (uint64_t)s->adam_v,
          (uint64_t)s->images6, (uint64_t)s->eps_x, (uint64_t)s->eps_x_hat, (uint64_t)st,
          adam ? f(s->beta1) : 0, adam ? f(s->beta2) : 0, adam ? f(s->adam_eps) : 0, adam ? f(s->grad_scale) : 0,
          (uint64_t)s->accumulate_metrics, p->host_state_bits(), (uint64_t)p->ws};
}

extern "C" int sv_lgvae_step(sv_lgvae_plan* p, const sv_lgvae_step_args* s, void* stream) {
  if (!p || !s) return SV_E_BADARG;
  if (!p->bound) return SV_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  const int ph = s->phases;
  const bool train = ph & SV_PHASE_BACKWARD;
  if ((ph & (SV_PHASE_PREP | SV_PHASE_FORWARD | SV_PHASE_ADAM | SV_PHASE_BACKWARD)) && !s->params) return SV_E_BADARG;
  if ((ph & (SV_PHASE_FWD_ENCODERS | SV_PHASE_LOSS)) && !s->images6) return SV_E_BADARG;
  if (train && !s->grads) return SV_E_BADARG;
  if ((ph & SV_PHASE_ADAM) && (!s->grads || !s->adam_m || !s->adam_v || s->t <= 0)) return SV_E_BADARG;
  // no capture on the legacy default stream (HIP forbids it): callers that want replay run on a created stream
  if (!p->graph_on || p->prof_on || !st) return run_phases(p, s, st);

  // hipGraph replay: the first step with a given key runs eagerly (it also raises the kernels' dynamic-LDS caps, which
  // must not happen inside a capture), the second is captured, every later one is one graph launch.
  const std::vector<uint64_t> key = graph_key(p, s, st);
  if (p->graphs.size() >= 16 && !p->graphs.count(key)) return run_phases(p, s, st);   // callers that rotate buffers: stay eager
  sv_lgvae_plan::GraphEntry& e = p->graphs[key];
  if (!e.exec && e.seen++ == 0) return run_phases(p, s, st);
  const float alpha = (ph & SV_PHASE_ADAM) ? (float)svk_adam_alpha(s->lr, s->beta1, s->beta2, s->t) : 0.f;
  SV_TRY(svk_set_dyn((SvDynArgs*)p->bp("dyn"), s->seed, s->step, s->sample_offset, alpha, st));
  if (!e.exec) {
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) return (int)hipGetLastError();
    p->dyn = (const SvDynArgs*)p->bp("dyn");
    const int rc = run_phases(p, s, st);
    p->dyn = nullptr;
    const hipError_t he = hipStreamEndCapture(st, &g);
    if (rc || he != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      return rc ? rc : (int)he;
    }
    const hipError_t hi = hipGraphInstantiate(&e.exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (hi != hipSuccess) { e.exec = nullptr; return (int)hi; }
    e.exit_state = p->host_state();              // what the captured phases left behind (the entry state is part of the key)
  } else {
    p->set_host_state(e.exit_state);             // the host-side state the replayed phases would have left behind
  }
  if (hipGraphLaunch(e.exec, st) != hipSuccess) return (int)hipGetLastError();
  return SV_OK;
}

extern "C" int sv_lgvae_graph_enable(sv_lgvae_plan* p, int32_t enable) {
  if (!p) return SV_E_BADARG;
  p->graph_on = enable != 0;
  if (!enable) {
    for (auto& kv : p->graphs)
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    p->graphs.clear();
  }
  return SV_OK;
}

extern "C" int sv_lgvae_bucket_wait(sv_lgvae_plan* p, int32_t bucket, void* stream) {
  if (!p || bucket < 0 || bucket > 3) return SV_E_BADARG;
  int n = 0;
  for (int k = (bucket == 3 ? 1 : bucket); k <= (bucket == 3 ? 2 : bucket); ++k)
    for (int i = 0; i <= sv_lgvae_plan::SIDE_MAX; ++i)
      if (p->bucket_rec[k][i]) {
        // test hook (negative control of tests/test_gpu_dist.py::test_buckets_wait_for_the_side_stream): drop the side streams' events
        if (p->dbg_bucket_skip_side && i > 0) continue;
        if (hipStreamWaitEvent((hipStream_t)stream, p->ev_bucket[k][i], 0) != hipSuccess) return (int)hipGetLastError();
        ++n;
      }
  return n ? SV_OK : SV_E_STATE;
}

// per-plan test hooks: "side_delay_us" (hold the first side-stream launch of every step back by that long), "bucket_skip_side" (sv_lgvae_bucket_wait
// drops the side streams' events: the negative control of the bucket-dependency test).  Deliberately an explicit call on one plan, not an environment
// variable: nothing a training job inherits can switch them on.
extern "C" int sv_lgvae_plan_debug(sv_lgvae_plan* p, const char* key, int64_t value) {
  if (!p || !key) return SV_E_BADARG;
  if (!strcmp(key, "side_delay_us")) { if (value < 0 || value > 1000000) return SV_E_BADARG; p->dbg_side_delay_us = (int)value; return SV_OK; }
  if (!strcmp(key, "bucket_skip_side")) { p->dbg_bucket_skip_side = value != 0; return SV_OK; }
  return SV_E_BADARG;
}

extern "C" int sv_lgvae_graph_count(const sv_lgvae_plan* p) {
  if (!p) return SV_E_BADARG;
  int n = 0;
  for (auto& kv : p->graphs) n += kv.second.exec != nullptr;
  return n;
}

extern "C" int sv_lgvae_profile_enable(sv_lgvae_plan* p, int32_t enable) {
  if (!p) return SV_E_BADARG;
  p->prof_on = enable != 0;
  if (enable) {
    for (auto& pe : p->pending) { p->event_pool.push_back(pe.a); p->event_pool.push_back(pe.b); }
    p->pending.clear();
    p->prof.clear();
    p->profidx.clear();
  }
  return SV_OK;
}

extern "C" int sv_lgvae_profile_filter(sv_lgvae_plan* p, const char* name) {
  if (!p) return SV_E_BADARG;
  p->prof_filter = name ? name : "";
  return SV_OK;
}

extern "C" int sv_lgvae_profile_read(sv_lgvae_plan* p, int32_t max_entries, char names[][64], double* total_ms,
                                     int32_t* launches, double* flops_per_launch, double* bytes_per_launch) {
  if (!p) return SV_E_BADARG;
  for (auto& pe : p->pending) {
    (void)hipEventSynchronize(pe.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
      p->prof[pe.entry].total_ms += ms;
      p->prof[pe.entry].launches += 1;
    }
    p->event_pool.push_back(pe.a);
    p->event_pool.push_back(pe.b);
  }
  p->pending.clear();
  int n = (int)p->prof.size();
  if (n > max_entries) n = max_entries;
  for (int i = 0; i < n; ++i) {
    if (names) snprintf(names[i], 64, "%s", p->prof[i].name.c_str());
    if (total_ms) total_ms[i] = p->prof[i].total_ms;
    if (launches) launches[i] = p->prof[i].launches;
    if (flops_per_launch) flops_per_launch[i] = p->prof[i].flops;
    if (bytes_per_launch) bytes_per_launch[i] = p->prof[i].bytes;
  }
  return n;
}

// FLOPs per launch the scope's chosen algorithm really multiplies (polyphase forms: fewer than the direct count flops_per_launch), same order as sv_lgvae_profile_read
extern "C" int sv_lgvae_profile_read_issued(sv_lgvae_plan* p, int32_t max_entries, double* issued_flops_per_launch) {
  if (!p || !issued_flops_per_launch) return SV_E_BADARG;
  int n = (int)p->prof.size();
  if (n > max_entries) n = max_entries;
  for (int i = 0; i < n; ++i) issued_flops_per_launch[i] = p->prof[i].issued;
  return n;
}

extern "C" const char* sv_version(void) { return "splitvae-hip 0.1 (gfx950)"; }

extern "C" int sv_set_deterministic(int32_t mode) {
  if (mode < -1 || mode > 1) return SV_E_BADARG;
  sv_deterministic_override().store(mode, std::memory_order_relaxed);
  return SV_OK;
}

extern "C" int sv_get_deterministic(void) { return sv_deterministic() ? 1 : 0; }
