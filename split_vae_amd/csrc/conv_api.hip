// Geometry + weight preparation + the public sv_conv2d_* entry points.
// TF 'SAME' padding [TF-2.0 semantics]: out = ceil(in/s), pad = max((out-1)s + k - in, 0),
// before = pad/2 (the extra row/column goes to the bottom/right: k4s1 -> 1/2, k6s1 -> 2/3).
#include "common.hip.h"
#include "kernels.h"
#include "conv_geom.h"
#include <stdlib.h>

// ============================================================================ weight preparation
// dst[(row*ntaps + t)*inner_ld + inner_off + c]:
//   forward  (transpose=0): row = co, c = ci : src[(srctap[t]*Cin + c)*Cout + row]
//   dgrad    (transpose=1): row = ci, c = co : src[(srctap[t]*Cin + row)*Cout + c]
// zero where row/c exceed the real channel counts (padding rows/channels of the MFMA tiles).
template <typename T>
__device__ __forceinline__ void prep_block(const PrepJob& j, const float* __restrict__ params,
                                           T* __restrict__ arena, int block_in_job) {
  if (j.poly >= 5) {
    // polyphase INPUT gradient (conv_geom.h: svg_polyd); source: the pk x pk HWIO master.  rows = ci (Cin padded), inner = dY channels (padded)
    const int K = j.pk, pad = (K - 1) / 2, R = svg_polyd_radius(K), NT = 2 * R + 1;
    const int total = j.poly == 5 ? j.rows * NT * NT * j.inner : j.poly == 6 ? 4 * 4 * NT * j.rows * j.inner : 4 * 4 * 4 * j.inner * j.rows;
    const int idx = block_in_job * 256 + threadIdx.x;
    if (idx >= total) return;
    const float* w = params + j.src_off;
    auto W = [&](int ky, int kx, int ci, int co) -> float {
      return (ky < 0 || kx < 0 || ci >= j.Cin || co >= j.Cout) ? 0.f : w[((int64_t)(ky * K + kx) * j.Cin + ci) * j.Cout + co];
    };
    float v = 0.f;
    if (j.poly == 5) {
      // [ci][t = dxi * NT + dyi][co] = W'_(py,px)[ty,tx][ci][co], (py, ty) from dyh = dyi - R, (px, tx) from dxh = dxi - R
      const int co = idx % j.inner, t = (idx / j.inner) % (NT * NT), ci = idx / (j.inner * NT * NT);
      int py, ty, px, tx;
      const bool oky = svg_polyd_hitap(K, t % NT - R, &py, &ty), okx = svg_polyd_hitap(K, t / NT - R, &px, &tx);
      if (oky && okx)
        for (int ky = 0; ky < K; ++ky) {
          const float cy = svg_pcoef(py, ky, ty, pad);
          if (cy == 0.f) continue;
          for (int kx = 0; kx < K; ++kx) {
            const float cx = svg_pcoef(px, kx, tx, pad);
            if (cx != 0.f) v += cy * cx * W(ky, kx, ci, co);
          }
        }
    } else if (j.poly == 6) {
      // [edge e: top, bottom, left, right][q][di][ci][co] = .25 sum_k c(p,k,t) (w[k+(q)][k] - w[k-(q)][k])   (rows; columns with the roles of ky / kx swapped)
      const int co = idx % j.inner, ci = (idx / j.inner) % j.rows, di = (idx / (j.inner * j.rows)) % NT, q = (idx / (j.inner * j.rows * NT)) & 3, e = idx / (j.inner * j.rows * NT * 4);
      int pp, tt, kp, km;
      const bool ok = svg_polyd_hitap(K, di - R, &pp, &tt);
      svg_polyd_kpm(K, e & 1, q, &kp, &km);
      if (ok && kp >= 0)                                 // (k+ < 0: q beyond this edge's rows: zero)
        for (int k = 0; k < K; ++k) {
          const float c = svg_pcoef(pp, k, tt, pad);
          if (c == 0.f) continue;
          v += 0.25f * c * (e < 2 ? W(kp, k, ci, co) - W(km, k, ci, co) : W(k, kp, ci, co) - W(k, km, ci, co));
        }
    } else {
      // [corner: TL, TR, BL, BR][qr][qs][ci][co] = .0625 (w[k+r][k+s] - w[k+r][k-s] - w[k-r][k+s] + w[k-r][k-s])
      const int co = idx % j.inner, ci = (idx / j.inner) % j.rows, qs = (idx / (j.rows * j.inner)) & 3, qr = (idx / (j.rows * j.inner * 4)) & 3, c = idx / (j.rows * j.inner * 16);
      int kpr, kmr, kps, kms;
      svg_polyd_kpm(K, c >> 1, qr, &kpr, &kmr);
      svg_polyd_kpm(K, c & 1, qs, &kps, &kms);
      if (kpr >= 0 && kps >= 0) v = 0.0625f * (W(kpr, kps, ci, co) - W(kpr, kms, ci, co) - W(kmr, kps, ci, co) + W(kmr, kms, ci, co));
    }
    arena[j.dst_off + idx] = from_f32<T>(v);
    return;
  }
  if (j.poly >= 3) {
    // per-class polyphase (conv_geom.h: svg_polyc); source: the pk x pk HWIO master
    const int total = j.poly == 3 ? j.rows * j.ntaps * j.inner : 2 * (j.pk - 1) * j.pk * j.rows * j.inner;
    const int idx = block_in_job * 256 + threadIdx.x;
    if (idx >= total) return;
    const float* w = params + j.src_off;
    const int K = j.pk, pad = (K - 1) / 2, ci = idx % j.inner;
    float v = 0.f;
    if (j.poly == 3) {
      // [co][t = txi * nty + tyi][ci]: W'_c[ty,tx] = sum_{ky,kx} cy(py,ky,ty) cx(px,kx,tx) w[ky,kx]
      const int py = j.pcls >> 1, px = j.pcls & 1;
      int ty0, tx0;
      const int nty = svg_polyc_taps(K, py, &ty0);
      (void)svg_polyc_taps(K, px, &tx0);
      const int t = (idx / j.inner) % j.ntaps, co = idx / (j.inner * j.ntaps);
      const int ty = ty0 + t % nty, tx = tx0 + t / nty;
      if (co < j.Cout && ci < j.Cin)
        for (int ky = 0; ky < K; ++ky) {
          const float cy = svg_pcoef(py, ky, ty, pad);
          if (cy == 0.f) continue;
          for (int kx = 0; kx < K; ++kx) {
            const float cx = svg_pcoef(px, kx, tx, pad);
            if (cx != 0.f) v += cy * cx * w[((int64_t)(ky * K + kx) * j.Cin + ci) * j.Cout + co];
          }
        }
    } else {
      // [cls][tap][co][ci] = -(sum over the taps k that leave the image at border class cls % (K-1)); classes 0 .. K-2: hi-res ROWS (tap = kx, the sum
      // runs over ky), K-1 .. 2K-3: COLUMNS (tap = ky, the sum runs over kx)
      const int co = (idx / j.inner) % j.rows, tap = (idx / (j.inner * j.rows)) % K, cls = idx / (j.inner * j.rows * K);
      const int c = cls % (K - 1);
      if (co < j.Cout && ci < j.Cin)
        for (int k = 0; k < K; ++k)
          if (svg_polyc_excl(K, c, k)) v -= w[((int64_t)(cls < K - 1 ? k * K + tap : tap * K + k) * j.Cin + ci) * j.Cout + co];
    }
    arena[j.dst_off + idx] = from_f32<T>(v);
    return;
  }
  if (j.poly) {
    // composite images of the polyphase head (conv_geom.h: svg_poly); source: the 6x6 HWIO master
    const int total = j.rows * j.ntaps * j.inner;
    const int idx = block_in_job * 256 + threadIdx.x;
    if (idx >= total) return;
    const float* w = params + j.src_off;
    const int ci = idx % j.inner;
    float v = 0.f;
    if (j.poly == 1) {
      // [n = (py*2+px)*8 + co][t = (tx+2)*5 + (ty+2)][ci]: blend coefficient of hi-res tap k of parity p on low-res offset t:
      // hi offset h = p + k - 2 from row 2i; even h = 2m: rows i+m-1 (.25), i+m (.75); odd h = 2m+1: i+m (.75), i+m+1 (.25)
      const int t = (idx / j.inner) % 25, n = idx / (j.inner * 25);
      const int co = n & 7, py = n >> 4, px = (n >> 3) & 1, tx = t / 5 - 2, ty = t % 5 - 2;
      auto coef = [](int p, int k, int t) -> float {
        const int h = p + k - 2, m = h >> 1;                      // arithmetic shift = floor
        if (h & 1) return t == m ? 0.75f : t == m + 1 ? 0.25f : 0.f;
        return t == m - 1 ? 0.25f : t == m ? 0.75f : 0.f;
      };
      if (co < j.Cout && ci < j.Cin)
        for (int ky = 0; ky < 6; ++ky) {
          const float cy = coef(py, ky, ty);
          if (cy == 0.f) continue;
          for (int kx = 0; kx < 6; ++kx) {
            const float cx = coef(px, kx, tx);
            if (cx != 0.f) v += cy * cx * w[((int64_t)(ky * 6 + kx) * j.Cin + ci) * j.Cout + co];
          }
        }
    } else {
      // border fix [cls][tap][n][ci] = -(sum over the excluded taps): classes 0..4 = hi-res rows 0, 1, 2H-3, 2H-2, 2H-1
      // (tap = kx, the sum runs over the ky that leave the image: {0,1}, {0}, {5}, {4,5}, {3,4,5}); 5..9 = the columns
      const int n = (idx / j.inner) & 15, tap = (idx / (j.inner * 16)) % 6, cls = idx / (j.inner * 96);
      const int c5 = cls % 5, lo = c5 == 0 ? 0 : c5 == 1 ? 0 : c5 == 2 ? 5 : c5 == 3 ? 4 : 3, hi = c5 == 0 ? 1 : c5 == 1 ? 0 : 5;
      if (n < j.Cout && ci < j.Cin)
        for (int k = lo; k <= hi; ++k) v -= w[((int64_t)(cls < 5 ? k * 6 + tap : tap * 6 + k) * j.Cin + ci) * j.Cout + n];
    }
    arena[j.dst_off + idx] = from_f32<T>(v);
    return;
  }
  if (j.s2d3) {
    // [co][t = tx * 3 + ty][ci16]: channel ci16 = (py*2+px)*3 + c of the space-to-depth view <- kernel position (2 ty + py, 2 tx + px), channel c (conv_geom.h: svg_s2d3)
    const int total = j.rows * 9 * 16;
    const int idx = block_in_job * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ci = idx & 15, t = (idx >> 4) % 9, co = idx / 144, tx = t / 3, ty = t - tx * 3, pq = ci / 3, c = ci - pq * 3;
    float v = 0.f;
    if (ci < 12 && co < j.Cout) v = params[j.src_off + ((int64_t)((2 * ty + (pq >> 1)) * 6 + 2 * tx + (pq & 1)) * 3 + c) * j.Cout + co];
    arena[j.dst_off + idx] = from_f32<T>(v);
    return;
  }
  if (j.packx_kw) {
    const int total = j.rows * j.ntaps * j.inner;
    const int idx = block_in_job * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % j.inner, t2 = idx / j.inner, t = t2 % j.ntaps, row = t2 / j.ntaps;
    const int KH = j.ntaps / (j.packx_kw + 1);        // taps are tx-major: t = tx * KH + ky (see svg_fwd_args)
    const int px = row >> 3, co = row & 7, tx = t / KH, ky = t - tx * KH;
    const int kx = tx - px;
    float v = 0.f;
    if (co < j.Cout && c < j.Cin && (unsigned)kx < (unsigned)j.packx_kw)
      v = params[j.src_off + ((int64_t)(ky * j.packx_kw + kx) * j.Cin + c) * j.Cout + co];
    arena[j.dst_off + ((int64_t)row * j.ntaps + t) * j.inner_ld + j.inner_off + c] = from_f32<T>(v);
    return;
  }
  if (!j.transpose) {
    // forward images dst[co][tap][ci] = src[tap][ci][co] (dense: one tap): per tap a [ci][co] -> [co][ci]
    // transpose in 64x64 tiles through LDS; fp32 reads are 16 B per thread along co (256-B row
    // segments), low-precision writes 16 B per thread along ci.  (One element per thread read the
    // HWIO master with a stride of Cout floats: the conv images took most of the kernel's time.)
    constexpr int EPP = ElemTraits<T>::EPP;
    __shared__ float tile[64][65];
    const int tr = (j.rows + 63) / 64, tc = (j.inner + 63) / 64;
    if (block_in_job >= j.ntaps * tr * tc) return;             // uniform per block
    const int t = block_in_job / (tr * tc), rem = block_in_job - t * (tr * tc);
    const int r0 = (rem % tr) * 64, c0 = (rem / tr) * 64;
    const int64_t soff = j.src_off + (int64_t)j.srctap[t] * j.Cin * j.Cout;
    const int64_t doff = j.dst_off + (int64_t)t * j.inner_ld + j.inner_off;
    const int64_t dpitch = (int64_t)j.ntaps * j.inner_ld;
    {
      const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
      const int row = r0 + tx * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cl = ty + 16 * k, c = c0 + cl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < j.Cin) {
          const float* sp = params + soff + (int64_t)c * j.Cout + row;
          if (row + 3 < j.Cout && !(j.Cout & 3) && !(soff & 3)) v = *(const float4*)sp;
          else {
            if (row < j.Cout) v.x = sp[0];
            if (row + 1 < j.Cout) v.y = sp[1];
            if (row + 2 < j.Cout) v.z = sp[2];
            if (row + 3 < j.Cout) v.w = sp[3];
          }
        }
        tile[cl][tx * 4] = v.x; tile[cl][tx * 4 + 1] = v.y; tile[cl][tx * 4 + 2] = v.z; tile[cl][tx * 4 + 3] = v.w;
      }
    }
    __syncthreads();
    {
      constexpr int TPR = 64 / EPP, RPP = 256 / TPR;            // threads per destination row, rows per pass
      const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
      for (int rl = ty; rl < 64; rl += RPP) {
        const int row = r0 + rl, c = c0 + tx * EPP;
        if (row >= j.rows || c >= j.inner) continue;
        T* dp = arena + doff + (int64_t)row * dpitch + c;
        T v[EPP];
#pragma unroll
        for (int e = 0; e < EPP; ++e) v[e] = from_f32<T>(tile[tx * EPP + e][rl]);
        if (c + EPP <= j.inner && !((doff + (int64_t)row * dpitch + c) % EPP)) *(uint4*)dp = *(uint4*)v;
        else
          for (int e = 0; e < EPP && c + e < j.inner; ++e) dp[e] = v[e];
      }
    }
    __syncthreads();                                           // the tile is reused by the block's next unit
    return;
  }
  if (j.transpose && !(j.inner & 7) && !(j.Cout & 7) && !(j.inner_off & 7) && !(j.inner_ld & 7)) {
    // dgrad images are contiguous in co on both sides: 8 channels per thread (two 16-B fp32 loads, one 16-B store
    // for bf16).  Source tensors are 16-B aligned (parameter table / allocator) and images 128-element aligned.
    const int q8 = j.inner >> 3;
    const int total8 = j.rows * j.ntaps * q8;
    const int idx = block_in_job * 256 + threadIdx.x;
    if (idx >= total8) return;
    const int c = (idx % q8) << 3, t2 = idx / q8, t = t2 % j.ntaps, row = t2 / j.ntaps;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (row < j.Cin && c < j.Cout) {
      const float* sp = params + j.src_off + ((int64_t)j.srctap[t] * j.Cin + row) * j.Cout + c;
      v0 = *(const float4*)sp; v1 = *(const float4*)(sp + 4);
    }
    T* d = arena + j.dst_off + ((int64_t)row * j.ntaps + t) * j.inner_ld + j.inner_off + c;
    T v[8] = {from_f32<T>(v0.x), from_f32<T>(v0.y), from_f32<T>(v0.z), from_f32<T>(v0.w),
              from_f32<T>(v1.x), from_f32<T>(v1.y), from_f32<T>(v1.z), from_f32<T>(v1.w)};
    if constexpr (sizeof(T) == 2) *(uint4*)d = *(uint4*)v;
    else { *(uint4*)d = *(uint4*)v; *(uint4*)(d + 4) = *(uint4*)(v + 4); }
    return;
  }
  const int total = j.rows * j.ntaps * j.inner;
  const int idx = block_in_job * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = idx % j.inner;
  const int t2 = idx / j.inner;
  const int t = t2 % j.ntaps;
  const int row = t2 / j.ntaps;
  float v = 0.f;
  const int st = j.srctap[t];
  if (!j.transpose) {
    if (row < j.Cout && c < j.Cin) v = params[j.src_off + ((int64_t)st * j.Cin + c) * j.Cout + row];
  } else {
    if (row < j.Cin && c < j.Cout) v = params[j.src_off + ((int64_t)st * j.Cin + row) * j.Cout + c];
  }
  arena[j.dst_off + ((int64_t)row * j.ntaps + t) * j.inner_ld + j.inner_off + c] = from_f32<T>(v);
}

template <typename T>
__global__ __launch_bounds__(256) void prep_table_kernel(const float* __restrict__ params, T* __restrict__ arena,
                                                         const PrepJob* __restrict__ jobs, int njobs, int block0) {
  // binary search the job owning this block (first_block is ascending); block0: the launch covers the table's blocks from there (a sub-range of the jobs)
  const int blk = (int)blockIdx.x + block0;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  const PrepJob& j = jobs[lo];   // by reference: uniform (scalar) loads, srctap[] indexed in memory
  // SV_PREP_UNITS units of 256 threads' work per block amortise the job lookup (dependent global loads)
  for (int u = 0; u < SV_PREP_UNITS; ++u) prep_block<T>(j, params, arena, (blk - j.first_block) * SV_PREP_UNITS + u);
}

template <typename T>
__global__ __launch_bounds__(256) void prep_single_kernel(const float* __restrict__ params, T* __restrict__ arena,
                                                          const PrepJob j) {
  for (int u = 0; u < SV_PREP_UNITS; ++u) prep_block<T>(j, params, arena, (int)blockIdx.x * SV_PREP_UNITS + u);
}

// blocks [block0, block0 + nblocks) of the job table (the whole table: 0, total)
int svk_prep_weights(const float* params, void* arena, int dtype, const PrepJob* jobs_dev, int njobs,
                     int nblocks, hipStream_t st, int block0) {
  if (nblocks < 1) return SV_OK;
  if (dtype == SV_BF16)
    hipLaunchKernelGGL((prep_table_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, params,
                       (bf16_t*)arena, jobs_dev, njobs, block0);
  else
    hipLaunchKernelGGL((prep_table_kernel<float>), dim3(nblocks), dim3(256), 0, st, params,
                       (float*)arena, jobs_dev, njobs, block0);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

static int prep_single(const float* params, void* arena, int dtype, const PrepJob& j, hipStream_t st) {
  if (dtype == SV_BF16)
    hipLaunchKernelGGL((prep_single_kernel<bf16_t>), dim3(j.nblocks), dim3(256), 0, st, params, (bf16_t*)arena, j);
  else
    hipLaunchKernelGGL((prep_single_kernel<float>), dim3(j.nblocks), dim3(256), 0, st, params, (float*)arena, j);
  SV_LAUNCH_CHECK();
  return SV_OK;
}

// ============================================================================ geometry
int svg_check(const sv_conv_desc* d) {
  if (!d) return SV_E_BADARG;
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0) return SV_E_BADARG;
  if (d->dtype != SV_BF16 && d->dtype != SV_F32) return SV_E_BADARG;
  if (d->stride < 1 || d->stride > 3) return SV_E_UNSUPPORTED;
  if (d->KH * d->KW > SV_MAX_TAPS) return SV_E_UNSUPPORTED;
  // Power-of-two extents run on the LDS-tile / row-ring kernels; any other extent (SPLIT-SPAIR's 48 / 24 / 12 backbone and
  // its stride-3 layer, spair/spair.py:382-384) on the im2col kernels, whose row decode divides.  Strided layers need the
  // stride to divide the extent (the input gradient iterates H/s x W/s parity classes).
  if (d->stride > 1 && ((d->H % d->stride) || (d->W % d->stride))) return SV_E_UNSUPPORTED;
  if (d->ldx < d->Cin || d->ldx % 8) return SV_E_BADARG;
  if (ilog2_exact(svg_cin_pad(d)) < 0) return SV_E_UNSUPPORTED;      // K pieces per tap: a power of two (the input gradient
                                                                   // contracts over Cout instead: checked there)
  if (d->ldy < d->Cout) return SV_E_BADARG;
  if (!d->y_f32 && d->ldy % 8) return SV_E_BADARG;
  if (d->ups_in && (d->stride != 1 || (d->H & 1) || (d->W & 1))) return SV_E_UNSUPPORTED;
  if ((int64_t)d->B * d->H * d->W * d->ldx >= (1LL << 31)) return SV_E_UNSUPPORTED;
  if ((int64_t)d->B * svg_oh(d) * svg_ow(d) * svg_gdy(d) >= (1LL << 31)) return SV_E_UNSUPPORTED;
  return SV_OK;
}

void svg_fwd_args(const sv_conv_desc* d, TapGemmArgs* a) {
  memset(a, 0, sizeof(*a));
  const int epp = svg_epp(d), cpad = svg_cin_pad(d);
  const int OH = svg_oh(d), OW = svg_ow(d);
  int pt, pl;
  svg_pads(d, &pt, &pl);
  a->M = d->B * OH * OW;
  a->lOY = ilog2_exact(OH); a->lOX = ilog2_exact(OW); a->OY = OH; a->OX = OW;
  a->IH = d->H; a->IW = d->W; a->lda = d->ldx;
  a->cl2 = ilog2_exact(cpad / epp);
  a->ntaps = d->KH * d->KW;
  a->Ktot = a->ntaps * cpad;
  a->P = a->Ktot / epp;
  a->S = d->stride; a->SX = d->stride;
  a->N = d->Cout;
  a->OHF = OH; a->OWF = OW; a->OS = 1; a->ooy = 0; a->oox = 0; a->ldo = d->ldy;
  a->act = d->act; a->out_f32 = d->y_f32; a->splitk = 1; a->ups = d->ups_in;
  if (svg_s2d3(d)) {
    // the 3 x 3 stride-1 conv over the space-to-depth view [B, H/2, W/2, 16] of the padded RGB tensor; taps x-major like every stride-1 forward
    a->IH = OH; a->IW = OW; a->lda = 16; a->cl2 = 2;
    a->ntaps = 9; a->Ktot = 144; a->P = 36; a->S = 1; a->SX = 1; a->s2d3 = 1;
    for (int tx = 0; tx < 3; ++tx)
      for (int ty = 0; ty < 3; ++ty) { a->dy[tx * 3 + ty] = (int8_t)(ty - 1); a->dx[tx * 3 + ty] = (int8_t)(tx - 1); }
    return;
  }
  if (svg_poly(d)) {
    // rows = LOW-RES pixels (i, j) of the [B, H/2, W/2, Cin] tensor; tap (ty, tx) in -2..2, x-major; 32 columns (py, px, co)
    const int h = d->H / 2, w = d->W / 2;
    a->M = d->B * h * w;
    a->lOY = ilog2_exact(h); a->lOX = ilog2_exact(w); a->OY = h; a->OX = w;
    a->IH = h; a->IW = w;
    a->ntaps = 25; a->Ktot = 25 * cpad; a->P = a->Ktot / epp;
    a->S = 1; a->SX = 1; a->N = 32; a->OS = 2; a->d2s = d->Cout; a->d2s_y = 1; a->clampin = 1; a->ups = 0;
    for (int tx = 0; tx < 5; ++tx)
      for (int ty = 0; ty < 5; ++ty) { a->dy[tx * 5 + ty] = (int8_t)(ty - 2); a->dx[tx * 5 + ty] = (int8_t)(tx - 2); }
    return;
  }
  if (svg_packx(d)) {
    // rows = output pixel pairs (y, 2X .. 2X+1); tap (ky, tx) reads input pixel (y + ky - pt, 2X + tx - pl)
    a->M = d->B * OH * (OW / 2);
    a->lOX = ilog2_exact(OW / 2); a->OX = OW / 2;
    a->ntaps = d->KH * (d->KW + 1);
    a->Ktot = a->ntaps * cpad;
    a->P = a->Ktot / epp;
    a->SX = 2; a->N = 16; a->d2s = d->Cout;
    for (int ky = 0; ky < d->KH; ++ky)
      for (int tx = 0; tx <= d->KW; ++tx) {
        a->dy[tx * d->KH + ky] = (int8_t)(ky - pt);
        a->dx[tx * d->KH + ky] = (int8_t)(tx - pl);
      }
    return;
  }
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) {
      const int t = svg_fwd_tap(d, kh, kw);
      a->dy[t] = (int8_t)(kh - pt);
      a->dx[t] = (int8_t)(kw - pl);
    }
}

int svg_dgrad_classes(const sv_conv_desc* d) { return d->stride * d->stride; }

// class cls = ph*stride + pw handles input pixels (ih, iw) = (stride*i2 + ph, stride*j2 + pw)
void svg_dgrad_args(const sv_conv_desc* d, int cls, TapGemmArgs* a, uint8_t srctap[SV_MAX_TAPS]) {
  memset(a, 0, sizeof(*a));
  const int epp = svg_epp(d), gdy = svg_gdy(d);
  const int OH = svg_oh(d), OW = svg_ow(d), s = d->stride;
  int pt, pl;
  svg_pads(d, &pt, &pl);
  const int ph = cls / s, pw = cls % s;
  const int gy = d->H / s, gx = d->W / s;     // iteration grid of this class
  a->M = d->B * gy * gx;
  a->lOY = ilog2_exact(gy); a->lOX = ilog2_exact(gx); a->OY = gy; a->OX = gx;
  a->IH = OH; a->IW = OW; a->lda = gdy;
  a->cl2 = ilog2_exact(gdy / epp);
  a->S = 1; a->SX = 1;
  a->N = d->Cin;
  a->OHF = d->H; a->OWF = d->W; a->OS = s; a->ooy = ph; a->oox = pw; a->ldo = d->ldx;
  a->act = SV_ACT_NONE; a->out_f32 = 0; a->splitk = 1;
  int nt = 0;
  if (s == 1) {
    // dx[ih] = sum_kh dy[ih - kh + pt] w[kh]; kh = KH-1-kh' -> offset kh' - (KH-1-pt)
    for (int kwp = 0; kwp < d->KW; ++kwp)        // x-major, y-minor (as the forward taps: svg_fwd_tap)
      for (int khp = 0; khp < d->KH; ++khp) {
        a->dy[nt] = (int8_t)(khp - (d->KH - 1 - pt));
        a->dx[nt] = (int8_t)(kwp - (d->KW - 1 - pl));
        srctap[nt] = (uint8_t)((d->KH - 1 - khp) * d->KW + (d->KW - 1 - kwp));
        ++nt;
      }
  } else {
    // input row ih = s*i2 + ph receives dy[oh] w[kh] with s*oh = ih + pt - kh: kh = kh0 + s*a, kh0 = (ph + pt) mod s,
    // oh = i2 + (ph + pt - kh0)/s - a.  (s = 2, even kernels: KH/2 taps in every class; s = 3, k = 4: 2 or 1.)
    const int kh0 = (ph + pt) % s, kw0 = (pw + pl) % s;
    for (int ay = 0; kh0 + s * ay < d->KH; ++ay)
      for (int ax = 0; kw0 + s * ax < d->KW; ++ax) {
        a->dy[nt] = (int8_t)((ph + pt - kh0) / s - ay);
        a->dx[nt] = (int8_t)((pw + pl - kw0) / s - ax);
        srctap[nt] = (uint8_t)((kh0 + s * ay) * d->KW + (kw0 + s * ax));
        ++nt;
      }
  }
  a->ntaps = nt;
  a->Ktot = nt * gdy;
  a->P = a->Ktot / epp;
}

// Stride-2 layers whose four parity classes read the SAME dY window (k = 6, pad 2: rows / columns -1..1 of the class
// grid in every class) run as ONE problem: the class weight images, laid back to back, are a [4*Cin][ntaps*gdy] image
// of a stride-1 conv dY -> 4*Cin columns on the class grid, and column n = (class, channel) is stored to the class's
// sub-pixel (TapGemmArgs::cls_n).  One staging of the dY tile and one A fragment then serve four classes.
// -> false when the classes differ (k = 4: windows {0,-1} and {1,0}), the widths do not fit a 128-column tile, or the
// caller's images are not contiguous (class_stride = elements between consecutive class images).
bool svg_dgrad_merged_args(const sv_conv_desc* d, const int64_t* class_off, TapGemmArgs* a) {
  static const bool off = getenv("SV_NO_CLS_MERGE") != nullptr;        // A/B: one problem per parity class
  if (off || d->stride != 2 || (d->Cin & 7) || 4 * d->Cin > 128 || (4 * d->Cin) % 32) return false;
  uint8_t srctap[SV_MAX_TAPS];
  svg_dgrad_args(d, 0, a, srctap);
  for (int c = 1; c < 4; ++c) {
    TapGemmArgs b;
    svg_dgrad_args(d, c, &b, srctap);
    if (b.ntaps != a->ntaps || memcmp(b.dy, a->dy, a->ntaps) || memcmp(b.dx, a->dx, a->ntaps)) return false;
    if (class_off[c] - class_off[c - 1] != svg_wprep_elems_class(d, 1, c - 1)) return false;
  }
  a->N = 4 * d->Cin; a->cls_n = d->Cin; a->ooy = 0; a->oox = 0;
  return true;
}

void svg_wgrad_args(const sv_conv_desc* d, WgradArgs* a) {
  memset(a, 0, sizeof(*a));
  const int epp = svg_epp(d), cpad = svg_cin_pad(d);
  const int OH = svg_oh(d), OW = svg_ow(d);
  int pt, pl;
  svg_pads(d, &pt, &pl);
  a->M = d->B * OH * OW;
  a->lOY = ilog2_exact(OH); a->lOX = ilog2_exact(OW); a->OY = OH; a->OX = OW;
  a->IH = d->H; a->IW = d->W; a->lda = d->ldx; a->S = d->stride; a->SX = d->stride;
  a->ldy = svg_gdy(d);
  a->ycols = svg_gdy(d);
  a->cl2 = ilog2_exact(cpad / epp);
  a->Cin_pad = cpad; a->Cin_real = d->Cin; a->N = d->Cout; a->ups = d->ups_in;
  a->ntaps = d->KH * d->KW;
  a->Nrows = a->ntaps * cpad;
  if (svg_s2d3(d)) {
    // space-to-depth form: 3 x 3 taps (y-major) over the 16-channel view; the reduce maps (tap, (py,px,c)) back to the [6][6][3][N] gradient (dw_index)
    a->IH = OH; a->IW = OW; a->lda = 16; a->S = 1; a->SX = 1; a->cl2 = 2;
    a->Cin_pad = 16; a->Cin_real = 12; a->ntaps = 9; a->Nrows = 144; a->s2d3 = 1;
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx) { a->dy[ty * 3 + tx] = (int8_t)(ty - 1); a->dx[ty * 3 + tx] = (int8_t)(tx - 1); }
    a->msplit = a->M;
    return;
  }
  if (svg_packx(d)) {
    // the x-packed conv's weight gradient: rows = pixel pairs, dY = the [B,H,W,8] gradient viewed as
    // [B,H,W/2,16]; the tile kernel folds dW' back into the HWIO gradient (the im2col kernel cannot)
    a->M = d->B * OH * (OW / 2);
    a->lOX = ilog2_exact(OW / 2); a->OX = OW / 2;
    a->SX = 2; a->ldy = 16; a->ycols = 16; a->N = 16;
    a->fold_kw = d->KW; a->fold_c = d->Cout;
    a->ntaps = d->KH * (d->KW + 1);
    a->Nrows = a->ntaps * cpad;
    for (int ky = 0; ky < d->KH; ++ky)
      for (int tx = 0; tx <= d->KW; ++tx) {
        a->dy[ky * (d->KW + 1) + tx] = (int8_t)(ky - pt);
        a->dx[ky * (d->KW + 1) + tx] = (int8_t)(tx - pl);
      }
    a->msplit = a->M;
    return;
  }
  for (int kh = 0; kh < d->KH; ++kh)
    for (int kw = 0; kw < d->KW; ++kw) {
      a->dy[kh * d->KW + kw] = (int8_t)(kh - pt);
      a->dx[kh * d->KW + kw] = (int8_t)(kw - pl);
    }
  static const int target_wgs = getenv("SV_WGRAD_WGS") ? atoi(getenv("SV_WGRAD_WGS")) : 512;
  svg_wgrad_set_msplit(a, svg_pick_cfg(d->Cout), d->dtype, target_wgs);
}

// m-splits of the im2col wgrad: every split adds one atomic pass over dW (~1.3 TB/s chip-wide), so use only as
// many as it takes to put `target_wgs` workgroups (~2 per CU; SV_WGRAD_WGS overrides for tuning) on the chip
void svg_wgrad_set_msplit(WgradArgs* a, int cfg, int dtype, int target_wgs) {
  static const int BRt[4] = {64, 128, 256, 256}, BNt[4] = {128, 64, 32, 16};
  const int ms = dtype == SV_BF16 ? 64 : 32;
  const int tiles = ((a->Nrows + BRt[cfg] - 1) / BRt[cfg]) * ((a->N + BNt[cfg] - 1) / BNt[cfg]);
  int z = target_wgs / (tiles > 0 ? tiles : 1);
  const int maxz = (a->M + ms - 1) / ms;
  if (z < 1 || sv_deterministic()) z = 1;      // deterministic: no m-split, the kernel's plain (non-atomic) epilogue
  if (z > maxz) z = maxz;
  a->msplit = round_up((a->M + z - 1) / z, ms);
}

void svg_prep_job_fwd(const sv_conv_desc* d, PrepJob* j) {
  memset(j, 0, sizeof(*j));
  static const int BNt[4] = {128, 64, 32, 16};
  j->ntaps = d->KH * d->KW;
  j->Cin = d->Cin; j->Cout = d->Cout;
  j->rows = round_up(d->Cout, BNt[svg_pick_cfg(d->Cout)]);
  j->inner = svg_cin_pad(d);
  j->inner_ld = j->inner; j->inner_off = 0;
  j->transpose = 0;
  if (svg_s2d3(d)) {
    j->ntaps = 9; j->inner = 16; j->inner_ld = 16; j->s2d3 = 1;
    j->nblocks = (j->rows * 144 + 256 * SV_PREP_UNITS - 1) / (256 * SV_PREP_UNITS);
    return;
  }
  if (svg_poly(d)) {
    j->ntaps = 25; j->rows = 32; j->poly = 1;
    j->nblocks = (j->rows * j->ntaps * j->inner + 256 * SV_PREP_UNITS - 1) / (256 * SV_PREP_UNITS);
    return;
  }
  if (svg_packx(d)) {
    j->ntaps = d->KH * (d->KW + 1);
    j->rows = 16;
    j->packx_kw = d->KW;
  }
  for (int t = 0; t < j->ntaps; ++t) j->srctap[t] = (uint8_t)t;
  if (!j->packx_kw)
    for (int kh = 0; kh < d->KH; ++kh)
      for (int kw = 0; kw < d->KW; ++kw) j->srctap[svg_fwd_tap(d, kh, kw)] = (uint8_t)(kh * d->KW + kw);
  j->nblocks = svg_prep_nblocks(j);
}

void svg_prep_job_polyfix(const sv_conv_desc* d, PrepJob* j) {
  memset(j, 0, sizeof(*j));
  j->Cin = d->Cin; j->Cout = d->Cout;
  j->rows = 160; j->ntaps = 6; j->inner = svg_cin_pad(d); j->inner_ld = j->inner;     // [10][6][16][Cin] (SV_POLY_FIX_ELEMS)
  j->poly = 2;
  j->nblocks = (j->rows * j->ntaps * j->inner + 256 * SV_PREP_UNITS - 1) / (256 * SV_PREP_UNITS);
}

// ---- per-class polyphase (conv_geom.h: svg_polyc)
void svg_polyc_fwd_args(const sv_conv_desc* d, int cls, TapGemmArgs* a) {
  memset(a, 0, sizeof(*a));
  const int epp = svg_epp(d), cpad = svg_cin_pad(d), h = d->H / 2, w = d->W / 2, K = d->KH;
  const int py = cls >> 1, px = cls & 1;
  int ty0, tx0;
  const int nty = svg_polyc_taps(K, py, &ty0), ntx = svg_polyc_taps(K, px, &tx0);
  a->M = d->B * h * w;
  a->lOY = ilog2_exact(h); a->lOX = ilog2_exact(w); a->OY = h; a->OX = w;
  a->IH = h; a->IW = w; a->lda = d->ldx;
  a->cl2 = ilog2_exact(cpad / epp);
  a->ntaps = nty * ntx; a->Ktot = a->ntaps * cpad; a->P = a->Ktot / epp;
  a->S = 1; a->SX = 1; a->N = d->Cout;
  a->OHF = d->H; a->OWF = d->W; a->OS = 2; a->ooy = py; a->oox = px; a->ldo = d->ldy;
  a->act = d->act; a->out_f32 = 0; a->splitk = 1; a->ups = 0; a->clampin = 1;
  a->fix_nc = svg_polyc_nclass(K); a->fix_pad = (K - 1) / 2;
  for (int txi = 0; txi < ntx; ++txi)
    for (int tyi = 0; tyi < nty; ++tyi) { a->dy[txi * nty + tyi] = (int8_t)(ty0 + tyi); a->dx[txi * nty + tyi] = (int8_t)(tx0 + txi); }
}

void svg_prep_job_polyc(const sv_conv_desc* d, int cls, PrepJob* j) {
  memset(j, 0, sizeof(*j));
  static const int BNt[4] = {128, 64, 32, 16};
  int t0;
  j->Cin = d->Cin; j->Cout = d->Cout;
  j->rows = round_up(d->Cout, BNt[svg_pick_cfg(d->Cout)]);
  j->inner = svg_cin_pad(d); j->inner_ld = j->inner;
  j->ntaps = svg_polyc_taps(d->KH, cls >> 1, &t0) * svg_polyc_taps(d->KH, cls & 1, &t0);
  j->poly = 3; j->pk = d->KH; j->pcls = cls;
  j->nblocks = (j->rows * j->ntaps * j->inner + 256 * SV_PREP_UNITS - 1) / (256 * SV_PREP_UNITS);
}

void svg_prep_job_polyc_fix(const sv_conv_desc* d, PrepJob* j) {
  memset(j, 0, sizeof(*j));
  j->Cin = d->Cin; j->Cout = d->Cout;
  j->rows = d->Cout; j->inner = svg_cin_pad(d); j->inner_ld = j->inner; j->ntaps = d->KH;
  j->poly = 4; j->pk = d->KH;
  j->nblocks = (int)((svg_polyc_fix_elems(d) + 256 * SV_PREP_UNITS - 1) / (256 * SV_PREP_UNITS));
}

int64_t svg_polyc_class_elems(const sv_conv_desc* d, int cls) {
  PrepJob j;
  svg_prep_job_polyc(d, cls, &j);
  return ((int64_t)j.rows * j.ntaps * j.inner + 127) / 128 * 128;      // every class image 128-element aligned
}
int64_t svg_polyc_fix_elems(const sv_conv_desc* d) { return (int64_t)2 * (d->KH - 1) * d->KH * d->Cout * svg_cin_pad(d); }
// row-class terms [B][K-1][W][Cout] + column-class terms [B][H][K-1][Cout], fp32 (H, W: the hi-res extent)
int64_t svg_polyc_fix_ws_bytes(const sv_conv_desc* d) { return (int64_t)d->B * (d->KH - 1) * (d->H + d->W) * d->Cout * 4; }

// ---- polyphase input gradient (conv_geom.h: svg_polyd)
void svg_polyd_args(const sv_conv_desc* d, TapGemmArgs* a) {
  memset(a, 0, sizeof(*a));
  const int epp = svg_epp(d), gdy = svg_gdy(d), h = d->H / 2, w = d->W / 2, R = svg_polyd_radius(d->KH), NT = 2 * R + 1;
  a->M = d->B * h * w;
  a->lOY = ilog2_exact(h); a->lOX = ilog2_exact(w); a->OY = h; a->OX = w;
  a->IH = d->H; a->IW = d->W; a->lda = gdy;             // the input is the hi-res dY
  a->cl2 = ilog2_exact(gdy / epp);
  a->ntaps = NT * NT; a->Ktot = a->ntaps * gdy; a->P = a->Ktot / epp;
  a->S = 2; a->SX = 2; a->N = d->Cin;
  a->OHF = h; a->OWF = w; a->OS = 1; a->ooy = 0; a->oox = 0; a->ldo = d->ldx;
  a->act = SV_ACT_NONE; a->out_f32 = 0; a->splitk = 1;
  a->fix_nc = 2; a->fix_pad = 1;                        // edge terms: first / last low-res row and column
  for (int dxi = 0; dxi < NT; ++dxi)
    for (int dyi = 0; dyi < NT; ++dyi) { a->dy[dxi * NT + dyi] = (int8_t)(dyi - R); a->dx[dxi * NT + dyi] = (int8_t)(dxi - R); }
}

void svg_prep_job_polyd(const sv_conv_desc* d, int which, PrepJob* j) {
  memset(j, 0, sizeof(*j));
  static const int BNt[4] = {128, 64, 32, 16};
  const int NT = 2 * svg_polyd_radius(d->KH) + 1;
  j->Cin = d->Cin; j->Cout = d->Cout; j->pk = d->KH; j->poly = 5 + which;
  if (which == 0) { j->rows = round_up(d->Cin, BNt[svg_pick_cfg(d->Cin)]); j->inner = svg_gdy(d); j->ntaps = NT * NT; }
  else { j->rows = svg_cin_pad(d); j->inner = svg_polyd_cop(d); j->ntaps = which == 1 ? 16 * NT : 64; }
  j->inner_ld = j->inner;
  j->nblocks = (int)(((int64_t)j->rows * j->ntaps * j->inner + 256 * SV_PREP_UNITS - 1) / (256 * SV_PREP_UNITS));
}
int64_t svg_polyd_elems(const sv_conv_desc* d, int which) {
  PrepJob j;
  svg_prep_job_polyd(d, which, &j);
  return ((int64_t)j.rows * j.ntaps * j.inner + 127) / 128 * 128;
}
int64_t svg_polyd_ws_bytes(const sv_conv_desc* d) { return (int64_t)d->B * 2 * (d->H / 2 + d->W / 2) * svg_cin_pad(d) * 4; }

void svg_prep_job_dgrad(const sv_conv_desc* d, int cls, PrepJob* j) {
  memset(j, 0, sizeof(*j));
  static const int BNt[4] = {128, 64, 32, 16};
  TapGemmArgs a;
  svg_dgrad_args(d, cls, &a, j->srctap);
  j->ntaps = a.ntaps;
  j->Cin = d->Cin; j->Cout = d->Cout;
  j->rows = round_up(d->Cin, BNt[svg_pick_cfg(d->Cin)]);
  j->inner = svg_gdy(d);
  j->inner_ld = j->inner; j->inner_off = 0;
  j->transpose = 1;
  j->nblocks = svg_prep_nblocks(j);
}

// stand-alone forward image of a per-class polyphase layer: the four class images, the border-class image, and -- 128-element aligned behind them -- the DIRECT
// (fused-resize) image, which the forward falls back to when the caller brings no workspace for the border terms (the plan's arena keeps the class images only)
static int64_t svg_polyc_direct_off(const sv_conv_desc* d) {
  int64_t n = svg_polyc_fix_elems(d);
  for (int c = 0; c < 4; ++c) n += svg_polyc_class_elems(d, c);
  return (n + 127) / 128 * 128;
}
int64_t svg_wprep_elems_class(const sv_conv_desc* d, int for_dgrad, int cls) {
  PrepJob j;
  if (!for_dgrad && svg_polyc(d)) {
    svg_prep_job_fwd(d, &j);
    return svg_polyc_direct_off(d) + (int64_t)j.rows * j.ntaps * j.inner;
  }
  if (for_dgrad) svg_prep_job_dgrad(d, cls, &j); else svg_prep_job_fwd(d, &j);
  return (int64_t)j.rows * j.ntaps * j.inner + (!for_dgrad && svg_poly(d) ? SV_POLY_FIX_ELEMS(svg_cin_pad(d)) : 0);
}

// ============================================================================ public API
// (for_dgrad: the direct-form class images; a layer with a polyphase input gradient -- svg_polyd -- keeps its main / edge / corner images behind them, 128-element aligned)
static int64_t svg_polyd_base(const sv_conv_desc* d) {
  int64_t n = 0;
  for (int c = 0; c < svg_dgrad_classes(d); ++c) n += svg_wprep_elems_class(d, 1, c);
  return (n + 127) / 128 * 128;
}
extern "C" int64_t sv_conv2d_wprep_elems(const sv_conv_desc* d, int32_t for_dgrad) {
  if (svg_check(d) != SV_OK) return -1;
  if (!for_dgrad) return svg_wprep_elems_class(d, 0, 0);
  int64_t n = 0;
  for (int c = 0; c < svg_dgrad_classes(d); ++c) n += svg_wprep_elems_class(d, 1, c);
  if (svg_polyd(d)) n = svg_polyd_base(d) + svg_polyd_elems(d, 0) + svg_polyd_elems(d, 1) + svg_polyd_elems(d, 2);
  return n;
}

extern "C" int sv_conv2d_prep_weights(const sv_conv_desc* d, const float* w_hwio, void* w_fwd, void* w_dgrad,
                                      void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!w_hwio) return SV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (w_fwd && svg_polyc(d)) {
    PrepJob j;
    int64_t off = 0;
    for (int c = 0; c < 4; ++c) {
      svg_prep_job_polyc(d, c, &j);
      j.dst_off = off;
      rc = prep_single(w_hwio, w_fwd, d->dtype, j, st);
      if (rc) return rc;
      off += svg_polyc_class_elems(d, c);
    }
    svg_prep_job_polyc_fix(d, &j);
    j.dst_off = off;
    rc = prep_single(w_hwio, w_fwd, d->dtype, j, st);
    if (rc) return rc;
    svg_prep_job_fwd(d, &j);                             // the direct image (no-workspace forward)
    j.dst_off = svg_polyc_direct_off(d);
    rc = prep_single(w_hwio, w_fwd, d->dtype, j, st);
    if (rc) return rc;
  } else if (w_fwd) {
    PrepJob j;
    svg_prep_job_fwd(d, &j);
    rc = prep_single(w_hwio, w_fwd, d->dtype, j, st);
    if (rc) return rc;
    if (svg_poly(d)) {                                   // the border-fix image follows the composite image
      const int64_t main_elems = (int64_t)j.rows * j.ntaps * j.inner;
      svg_prep_job_polyfix(d, &j);
      j.dst_off = main_elems;
      rc = prep_single(w_hwio, w_fwd, d->dtype, j, st);
      if (rc) return rc;
    }
  }
  if (w_dgrad) {
    int64_t off = 0;
    for (int c = 0; c < svg_dgrad_classes(d); ++c) {
      PrepJob j;
      svg_prep_job_dgrad(d, c, &j);
      j.dst_off = off;
      rc = prep_single(w_hwio, w_dgrad, d->dtype, j, st);
      if (rc) return rc;
      off += (int64_t)j.rows * j.ntaps * j.inner;
    }
    if (svg_polyd(d)) {
      off = svg_polyd_base(d);
      for (int which = 0; which < 3; ++which) {
        PrepJob j;
        svg_prep_job_polyd(d, which, &j);
        j.dst_off = off;
        rc = prep_single(w_hwio, w_dgrad, d->dtype, j, st);
        if (rc) return rc;
        off += svg_polyd_elems(d, which);
      }
    }
  }
  return SV_OK;
}

// Forward with an optional workspace.  The polyphase head (svg_poly) delivers its border terms through the workspace
// (fix kernel first, added by the conv's epilogue); without one they are added to y with atomics after the conv.
extern "C" int64_t sv_conv2d_fwd_workspace_bytes(const sv_conv_desc* d) {
  if (svg_check(d) != SV_OK) return -1;
  if (svg_polyc(d)) return svg_polyc_fix_ws_bytes(d);
  return svg_poly(d) ? svk_poly_fix_ws_bytes(d->B, d->H / 2, d->W / 2) : 0;
}

// the four class problems + the border kernel of n <= 2 per-class polyphase layers (svg_polyc) of one geometry; w_fwd[i] = the image
// sv_conv2d_prep_weights / the plan's jobs laid out (classes 0..3, then the border classes), fixws[i] = svg_polyc_fix_ws_bytes(d) bytes
int svk_polyc_fwd_multi(const sv_conv_desc* d, int n, const void* const* x, const void* const* w_fwd, const float* const* bias, void* const* y,
                        void* const* fixws, hipStream_t st) {
  if (n < 1 || n > 2) return SV_E_BADARG;
  const size_t esz = d->dtype == SV_BF16 ? 2 : 4;
  const int K = d->KH, nc = svg_polyc_nclass(K);
  int64_t coff[5] = {0, 0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) coff[c + 1] = coff[c] + svg_polyc_class_elems(d, c);
  const void* wfix[2];
  float *frow[2], *fcol[2];
  for (int i = 0; i < n; ++i) {
    wfix[i] = (const char*)w_fwd[i] + coff[4] * esz;
    frow[i] = (float*)fixws[i];
    fcol[i] = frow[i] + (int64_t)d->B * nc * d->W * d->Cout;
  }
  int rc = svk_polyc_fix_multi(n, x, wfix, frow, fcol, d->B, d->H / 2, d->W / 2, svg_cin_pad(d), d->Cout, K, d->dtype, st);
  if (rc) return rc;
  TapGemmArgs a[SV_MAX_MULTI];
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 4; ++c) {
      TapGemmArgs& t = a[i * 4 + c];
      svg_polyc_fwd_args(d, c, &t);
      t.A = x[i]; t.Wt = (const char*)w_fwd[i] + coff[c] * esz; t.bias = bias ? bias[i] : nullptr; t.out = y[i];
      t.fix = frow[i]; t.fix2 = fcol[i];
    }
  return svk_conv_dispatch_multi(a, 4 * n, d->dtype, svg_pick_cfg(d->Cout), st);
}

// do the four class problems of a per-class polyphase layer plan on the tile kernel?  Asked when a plan chooses the layer's form (lgvae_plan.hip), so that a
// refusal (LDS, shape) selects the direct form at creation instead of failing the step after the border kernel has run
bool svk_polyc_fwd_plannable(const sv_conv_desc* d) {
  static float dummy[4];
  for (int c = 0; c < 4; ++c) {
    TapGemmArgs t;
    svg_polyc_fwd_args(d, c, &t);
    t.A = dummy; t.Wt = dummy; t.out = dummy; t.fix = dummy; t.fix2 = dummy;     // (the planner only tests them for presence)
    TileConvArgs a;
    int cfg = 0;
    if (!svk_tile_conv_plan(t, d->dtype, d->B, &a, &cfg)) return false;
  }
  return true;
}

// im2col tile for a layer the tile kernel does not plan (non-power-of-two grids, stride 3: SPAIR's backbone): when the
// 128-row tiles of svg_pick_cfg leave most of the 256 CUs idle, 64 x 32 tiles (tap-GEMM cfg 4) give 8x the workgroups.
static int svg_im2col_cfg(const TapGemmArgs& a, int cfg) {
  static const bool off = getenv("SV_NO_SMALL_IM2COL") != nullptr;
  static const int BMt[4] = {128, 128, 256, 256}, BNt[4] = {128, 64, 32, 16};
  if (off || cfg > 2 || (a.N % 32) || a.splitk > 1) return cfg;
  const int64_t tiles = (int64_t)((a.M + BMt[cfg] - 1) / BMt[cfg]) * ((a.N + BNt[cfg] - 1) / BNt[cfg]);
  return tiles < 256 ? 4 : cfg;
}

extern "C" int sv_conv2d_nhwc_fwd_ws(const sv_conv_desc* d, const void* x, const void* w_fwd, const float* bias, void* y,
                                     void* ws, int64_t ws_bytes, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!x || !w_fwd || !y) return SV_E_BADARG;
  const bool polyc = svg_polyc(d);
  if (polyc && ws && ws_bytes >= svg_polyc_fix_ws_bytes(d))                    // the border terms travel through the workspace
    return svk_polyc_fwd_multi(d, 1, &x, &w_fwd, &bias, &y, &ws, (hipStream_t)stream);
  TapGemmArgs a;
  svg_fwd_args(d, &a);
  a.A = x; a.Wt = w_fwd; a.bias = bias; a.out = y;
  // (per-class polyphase layer without a workspace: the direct fused-resize form on the direct image sv_conv2d_prep_weights keeps behind the class images)
  if (polyc) a.Wt = (const char*)w_fwd + svg_polyc_direct_off(d) * (int64_t)(d->dtype == SV_BF16 ? 2 : 4);
  if (!svg_poly(d)) {
    // dense layers (1x1 on a 1x1 grid) with an fp32 pre-activation output: a [B, Cin] x [Cin, Cout] GEMM whose tile grid is a
    // handful of workgroups (SPLIT-GMVAE's y_block / prior / posterior layers, vae/model.py:54-75) -- split K into the zeroed output
    static const bool no_sk = getenv("SV_NO_DENSE_SPLITK") != nullptr;
    if (!no_sk && d->y_f32 && d->act == SV_ACT_NONE && d->H == 1 && d->W == 1 && d->KH == 1 && d->KW == 1 && !d->ups_in) {
      int cfg = svg_pick_cfg(d->Cout);
      const int sk = svg_choose_splitk(a.M, a.N, (a.P + 7) / 8, &cfg);
      if (sk > 1) {
        if (hipMemsetAsync(y, 0, (size_t)d->B * d->ldy * sizeof(float), (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
        a.splitk = sk;
        return svk_tap_gemm(a, d->dtype, cfg, (hipStream_t)stream);
      }
    }
    // fp32 conv layers whose output is a handful of tiles but whose K is deep (SPLIT-GMVAE's 128 -> 128 stride-2 layers at 64 images: 8 x 8 -> 4 x 4 pixels, K = 2048:
    // the tile kernel ran 32 workgroups for 77 us): K split over workgroups on the im2col GEMM, fp32 partial sums added into the zeroed output (the bias rides on slice 0).
    // Only without an activation (atomics cannot apply one), never under SV_DETERMINISTIC (svg_choose_splitk returns 1).  SV_CONV_SPLITK_TILES: the largest 128 x 128-tile
    // count that takes the form (0: off).  Measured: profiles/r06_gm_streams.txt
    // (SPLIT-GMVAE fp32, 64 images: off 1.708 ms per step, 8 tiles -- the 4 x 4 layer -- 1.656, 32 tiles -- the 8 x 8 layer too -- 1.633)
    static const int sk_tiles = getenv("SV_CONV_SPLITK_TILES") ? atoi(getenv("SV_CONV_SPLITK_TILES")) : 32;
    if (sk_tiles > 0 && d->dtype == SV_F32 && d->act == SV_ACT_NONE && !d->ups_in && !polyc && !svg_s2d3(d) && !svg_packx(d) && d->KH > 1 && d->ldy == d->Cout &&
        (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128) <= sk_tiles && a.P >= 256) {
      int cfg = svg_pick_cfg(d->Cout);
      const int sk = svg_choose_splitk(a.M, a.N, (a.P + 7) / 8, &cfg);
      if (sk > 1) {
        if (hipMemsetAsync(y, 0, (size_t)a.M * d->ldy * sizeof(float), (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
        a.splitk = sk; a.out_f32 = 1;
        return svk_tap_gemm(a, d->dtype, cfg, (hipStream_t)stream);
      }
    }
    return svk_conv_dispatch(a, d->dtype, svg_im2col_cfg(a, svg_pick_cfg(d->Cout)), (hipStream_t)stream);
  }
  const void* wfix = (const char*)w_fwd + (int64_t)32 * 25 * svg_cin_pad(d) * (d->dtype == SV_BF16 ? 2 : 4);
  float* yf = (float*)y;
  float* fixbuf = ws && ws_bytes >= svk_poly_fix_ws_bytes(d->B, d->H / 2, d->W / 2) ? (float*)ws : nullptr;
  if (fixbuf) {
    rc = svk_poly_fix_multi(1, &x, &wfix, nullptr, &fixbuf, d->B, d->H / 2, d->W / 2, d->ldx, d->Cout, (hipStream_t)stream, d->dtype);
    if (rc) return rc;
    a.fix = fixbuf;
  }
  rc = svk_conv_dispatch(a, d->dtype, svg_pick_cfg(d->Cout), (hipStream_t)stream);
  if (rc == SV_OK && !fixbuf)
    rc = svk_poly_fix_multi(1, &x, &wfix, &yf, nullptr, d->B, d->H / 2, d->W / 2, d->ldx, d->Cout, (hipStream_t)stream, d->dtype);
  return rc;
}

extern "C" int sv_conv2d_nhwc_fwd(const sv_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                                  void* y, void* stream) {
  return sv_conv2d_nhwc_fwd_ws(d, x, w_fwd, bias, y, nullptr, 0, stream);
}

extern "C" int sv_conv2d_nhwc_dgrad(const sv_conv_desc* d, const void* dy, const void* w_dgrad,
                                    const void* relu_mask, void* dx, int32_t dx_f32_atomic, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!dy || !w_dgrad || !dx) return SV_E_BADARG;
  if (dx_f32_atomic && relu_mask) return SV_E_BADARG;
  if (ilog2_exact(svg_gdy(d)) < 0) return SV_E_UNSUPPORTED;
  const size_t esz = d->dtype == SV_BF16 ? 2 : 4;
  int64_t off = 0;
  if (!dx_f32_atomic && svg_dgrad_classes(d) == 4) {      // classes with one shared window: one launch (cls_n)
    int64_t coff[4] = {0, 0, 0, 0};
    for (int c = 1; c < 4; ++c) coff[c] = coff[c - 1] + svg_wprep_elems_class(d, 1, c - 1);
    TapGemmArgs a;
    if (svg_dgrad_merged_args(d, coff, &a)) {
      a.A = dy; a.Wt = w_dgrad; a.out = dx; a.mask = relu_mask;
      rc = svk_conv_dispatch(a, d->dtype, svg_pick_cfg(a.N), (hipStream_t)stream);
      if (rc != SV_E_UNSUPPORTED) return rc;
    }
  }
  if (!dx_f32_atomic && svg_dgrad_classes(d) == 4) {       // (stride 2; SPLIT-SPAIR's stride-3 backbone layer has nine classes and stays on the per-class loop)
    // the parity classes of a stride-2 layer as ONE launch (they plan to the same tile grid; lgvae_plan.hip: run_dgrad_layers does the same): SPLIT-GMVAE's 128-channel
    // encoder layers -- too wide for the merged form above -- went out as four launches of 32 / 128 workgroups each (4 x 21 + 4 x 36 us of its 64-image fp32 step)
    TapGemmArgs a[4];
    int64_t o2 = 0;
    const int ncls = svg_dgrad_classes(d);
    for (int c = 0; c < ncls; ++c) {
      uint8_t srctap[SV_MAX_TAPS];
      svg_dgrad_args(d, c, &a[c], srctap);
      a[c].A = dy; a[c].Wt = (const char*)w_dgrad + o2 * esz; a[c].out = dx; a[c].mask = relu_mask;
      o2 += svg_wprep_elems_class(d, 1, c);
    }
    rc = svk_conv_dispatch_multi(a, ncls, d->dtype, svg_im2col_cfg(a[0], svg_pick_cfg(d->Cin)), (hipStream_t)stream);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  for (int c = 0; c < svg_dgrad_classes(d); ++c) {
    TapGemmArgs a;
    uint8_t srctap[SV_MAX_TAPS];
    svg_dgrad_args(d, c, &a, srctap);
    a.A = dy; a.Wt = (const char*)w_dgrad + off * esz; a.out = dx; a.mask = relu_mask;
    int cfg = svg_pick_cfg(d->Cin);
    if (dx_f32_atomic) {
      a.out_f32 = 1;
      a.accum = 1;                         // dx is an accumulator whatever the split (kernels.h)
      a.splitk = svg_choose_splitk(a.M, a.N, (a.P + 7) / 8, &cfg);
    } else cfg = svg_im2col_cfg(a, cfg);
    rc = svk_conv_dispatch(a, d->dtype, cfg, (hipStream_t)stream);
    if (rc) return rc;
    off += svg_wprep_elems_class(d, 1, c);
  }
  return SV_OK;
}

// Input gradient of a layer whose input is the 2x bilinear upsample of a low-res tensor (ups_in), delivered at the
// LOW-RES tensor: dx_lo = mask(resize_adjoint(conv_transpose(dy, w))) in one launch (row_conv.hip).
extern "C" int sv_conv2d_nhwc_dgrad_lowres(const sv_conv_desc* d, const void* dy, const void* w_dgrad,
                                           const void* relu_mask_lo, void* dx_lo, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!dy || !w_dgrad || !dx_lo) return SV_E_BADARG;
  if (!d->ups_in || d->stride != 1) return SV_E_BADARG;
  if (ilog2_exact(svg_gdy(d)) < 0) return SV_E_UNSUPPORTED;
  TapGemmArgs a;
  uint8_t srctap[SV_MAX_TAPS];
  svg_dgrad_args(d, 0, &a, srctap);
  a.A = dy; a.Wt = w_dgrad; a.out = dx_lo; a.mask = relu_mask_lo; a.adj = 1;
  return svk_conv_dispatch(a, d->dtype, svg_pick_cfg(d->Cin), (hipStream_t)stream);
}

// The same with a workspace: layers with a polyphase input gradient (svg_polyd: the fp32 step's d4 / d5) deliver their edge terms through it.
extern "C" int64_t sv_conv2d_dgrad_lowres_workspace_bytes(const sv_conv_desc* d) {
  if (svg_check(d) != SV_OK) return -1;
  return svg_polyd(d) ? svg_polyd_ws_bytes(d) : 0;
}
extern "C" int sv_conv2d_nhwc_dgrad_lowres_ws(const sv_conv_desc* d, const void* dy, const void* w_dgrad, const void* relu_mask_lo, void* dx_lo,
                                              void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!dy || !w_dgrad || !dx_lo) return SV_E_BADARG;
  if (svg_polyd(d) && workspace && workspace_bytes >= svg_polyd_ws_bytes(d)) {
    const void* wp = (const char*)w_dgrad + svg_polyd_base(d) * (d->dtype == SV_BF16 ? 2 : 4);
    return svk_polyd_dgrad_multi(d, 1, &dy, &wp, &relu_mask_lo, &dx_lo, &workspace, (hipStream_t)stream);
  }
  return sv_conv2d_nhwc_dgrad_lowres(d, dy, w_dgrad, relu_mask_lo, dx_lo, stream);
}

extern "C" int sv_conv2d_nhwc_wgrad(const sv_conv_desc* d, const void* x, const void* dy, float* dw,
                                    float* dbias, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!x || !dy || !dw) return SV_E_BADARG;
  WgradArgs a;
  svg_wgrad_args(d, &a);
  a.A = x; a.dY = dy; a.dW = dw; a.dbias = dbias;
  return svk_wgrad_dispatch(a, d->dtype, svg_pick_cfg(d->Cout), (hipStream_t)stream);
}

// Main term of the polyphase weight gradient of a svg_poly layer (poly_wgrad.hip, tests/test_polyphase_math.py):
// dWp[t = (tx+2)*5 + (ty+2)][ci][(py*2+px)*8 + co] = sum x~[i+ty, j+tx, ci] dy[2i+py, 2j+px, co] on the LOW-RES grid (tile kernel id 8)
void svg_poly_wgrad_args(const sv_conv_desc* d, WgradArgs* a) {
  memset(a, 0, sizeof(*a));
  const int h = d->H / 2, w = d->W / 2, cpad = svg_cin_pad(d);
  a->M = d->B * h * w; a->lOY = ilog2_exact(h); a->lOX = ilog2_exact(w); a->OY = h; a->OX = w;
  a->IH = h; a->IW = w; a->lda = d->ldx; a->S = 1; a->SX = 1;
  a->ldy = 32; a->ycols = 32; a->cl2 = ilog2_exact(cpad / 8);
  a->Cin_pad = cpad; a->Cin_real = d->Cin; a->N = 32; a->ntaps = 25; a->Nrows = 25 * cpad;
  a->clampin = 1; a->dy_s2d = 8; a->assign = 1; a->msplit = a->M;
  for (int tx = 0; tx < 5; ++tx)
    for (int ty = 0; ty < 5; ++ty) { a->dy[tx * 5 + ty] = (int8_t)(ty - 2); a->dx[tx * 5 + ty] = (int8_t)(tx - 2); }
}


// The polyphase weight gradient as one call (tests, micro-benchmarks; the training plan sequences the same kernels): dw / dbias
// accumulated like sv_conv2d_nhwc_wgrad; workspace of sv_conv2d_wgrad_poly_workspace_bytes, whose first use must find it zeroed.
extern "C" int64_t sv_conv2d_wgrad_poly_workspace_bytes(const sv_conv_desc* d) {
  if (svg_check(d) != SV_OK) return -1;
  if (!svg_poly(d) || d->dtype != SV_BF16) return 0;         // (fp32: the x-packed weight gradient, sv_conv2d_nhwc_wgrad_ws)
  return SV_WGRAD_WS_BYTES + svk_poly_wgrad_ws_floats(svg_cin_pad(d), SV_POLY_WGRAD_NWG) * 4;
}
extern "C" int sv_conv2d_nhwc_wgrad_poly(const sv_conv_desc* d, const void* x_lo, const void* dy, float* dw, float* dbias, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!svg_poly(d) || d->dtype != SV_BF16) return SV_E_UNSUPPORTED;
  if (!x_lo || !dy || !dw || !workspace || workspace_bytes < sv_conv2d_wgrad_poly_workspace_bytes(d)) return SV_E_BADARG;
  float* pw = (float*)((char*)workspace + SV_WGRAD_WS_BYTES);
  const int Cin = svg_cin_pad(d);
  if (!svk_poly_wgrad_supported(d->H / 2, d->W / 2, Cin, d->Cout)) return SV_E_UNSUPPORTED;   // before anything is enqueued
  WgradArgs a;
  svg_poly_wgrad_args(d, &a);
  a.A = x_lo; a.dY = dy; a.dW = pw; a.dbias = pw + 25 * Cin * 32; a.ws = (float*)workspace; a.ws_bytes = SV_WGRAD_WS_BYTES;
  rc = svk_wgrad_tile(a, (hipStream_t)stream);
  if (rc) return rc;
  return svk_poly_wgrad_finish(1, &x_lo, &dy, &pw, &dw, &dbias, d->B, d->H / 2, d->W / 2, d->ldx, Cin, d->Cout, SV_POLY_WGRAD_NWG,
                               (hipStream_t)stream);
}

// (layers with a polyphase weight gradient at fp32 -- polyc_wgrad.hip -- keep dW' and the frame slabs behind the partial-sum slabs)
extern "C" int64_t sv_conv2d_wgrad_workspace_bytes(const sv_conv_desc* d) {
  if (svg_check(d) != SV_OK) return -1;
  return SV_WGRAD_WS_BYTES + svk_polyc_wgrad_ws_floats(d) * 4;
}

extern "C" int sv_conv2d_nhwc_wgrad_ws(const sv_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = svg_check(d);
  if (rc != SV_OK) return rc;
  if (!x || !dy || !dw) return SV_E_BADARG;
  if (svg_polyc_wgrad_form(d) && workspace && workspace_bytes >= sv_conv2d_wgrad_workspace_bytes(d)) {
    float* slab = (float*)workspace;
    float* pw = (float*)((char*)workspace + SV_WGRAD_WS_BYTES);
    rc = svk_polyc_wgrad_multi(d, 1, &x, &dy, &dw, &dbias, &slab, SV_WGRAD_WS_BYTES, &pw, (hipStream_t)stream);
    if (rc != SV_E_UNSUPPORTED) return rc;
  }
  WgradArgs a;
  svg_wgrad_args(d, &a);
  a.A = x; a.dY = dy; a.dW = dw; a.dbias = dbias;
  a.ws = (float*)workspace; a.ws_bytes = workspace ? workspace_bytes : 0;
  return svk_wgrad_dispatch(a, d->dtype, svg_pick_cfg(d->Cout), (hipStream_t)stream);
}
