// Internal (non-ABI) launch helpers shared by the translation units of libsplitvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SV_MAX_TAPS 81      // 9 x 9: the stride-2 hi-res tap form of the polyphase input gradient (conv_geom.h: svg_polyd); 6 x 7 = 42: the x-packed 6x6 conv (svg_packx)

// ---- generic "tap GEMM": out[m, n] = sum_{t, c} A[pix(m) + tap t, c] * Wt[n][t][c]
// rows m enumerate (b, oy, ox) over a power-of-two OY x OX grid; taps are (dy, dx) offsets on an
// input grid sampled at stride S.  Forward convs, every dgrad (per parity class for stride 2)
// and the dense layers are all instances.
struct TapGemmArgs {
  const void* A;        // [B, IH, IW, lda] activations (dtype T)
  const void* Wt;       // [Npad][ntaps*Cin] prepared weights (dtype T), Npad multiple of the N tile
  const float* bias;    // [N] or null
  void* out;            // T, or float when out_f32
  const void* mask;     // optional ReLU mask (T), indexed like out
  int M;                // B*OY*OX
  int lOY, lOX;         // log2 of the per-image iteration grid, or -1 when OY / OX is not a power of two (SPLIT-SPAIR's
                        // 48 -> 24 -> 12 -> 4 backbone, spair/spair.py:382-384: im2col kernels only, divide-based row decode)
  int OY, OX;           // the per-image iteration grid itself
  int IH, IW, lda;
  int cl2;              // log2(16-byte pieces per tap) ; Cin = (1<<cl2)*EPP
  int P;                // total 16-byte pieces along K
  int Ktot;             // ntaps*Cin  (row length of Wt in elements)
  int S, SX;            // input coordinate = oy*S + dy, ox*SX + dx  (SX = S except for the x-packed conv)
  int N;                // real output channels
  int OHF, OWF, OS, ooy, oox, ldo;   // out pixel = ((b*OHF + oy*OS+ooy)*OWF + ox*OS+oox), ldo channels
  int act, out_f32, splitk, ntaps;
  int ups;              // A is the LOW-RES tensor [B, IH/2, IW/2, lda]; the conv sees its 2x bilinear upsample
  int d2s;              // > 0: x-pixel-packed conv: column n = px*8 + co holds channel co < d2s of output pixel
                        // (oy, 2*ox + px); out is the unpacked [B, OHF, OWF, ldo] tensor
  int d2s_y;            // with d2s: N = 32 columns n = (py*2 + px)*8 + co of output pixel (2*oy + py, 2*ox + px): the POLYPHASE
                        // form of (2x bilinear upsample -> conv): a plain conv over the LOW-RES tensor whose four output
                        // parities are four column classes (conv_api.hip: svg_poly)
  int clampin;          // input coordinates outside the image clamp to the edge (replicate) instead of reading zero
  int s2d3;             // A is the 8-channel padded RGB tensor [B, 2 IH, 2 IW, 8]; the conv sees its space-to-depth view [B, IH, IW, 16] (channel (py*2+px)*3 + c,
                        // 12 real + 4 zero): the 6 x 6 stride-2 first encoder layer as a 3 x 3 stride-1 conv with K = 144 instead of 288 (conv_geom.h: svg_s2d3)
  // FUSED LOSS (with d2s_y, training step): the epilogue evaluates the discretised-logistic NLL of its pixels against
  // nll_img (images6 [B,OHF,OWF,6], channels nll_ch..nll_ch+2), writes the gradient nll_gscale * d nll / d out6 to nll_grad
  // ([B,OHF,OWF,8] bf16, what dlogistic_kernel would write) and one partial sum per (image, tile) to nll_part[b * tiles + tile];
  // out is still written.  Needs one image per tile (OY * OX >= 256).
  const float* nll_img; void* nll_grad; float* nll_part; int nll_ch; float nll_gscale;
  int nll_noout;              // fused loss: the reconstruction (out6) is not stored (SV_PHASE_NO_RECON: dead after the loss in a training step)
  const float* fix;     // with d2s_y: border terms [B][10][max(OHF, OWF)][8] added by the epilogue (poly_fix.hip), or null
  // without d2s_y (per-class polyphase, conv_geom.h: svg_polyc): fix = row-class terms [B][fix_nc][OWF][N], fix2 = column-class terms [B][OHF][fix_nc][N],
  // added before the activation to the pixels of hi-res rows / columns 0 .. fix_pad-1 and OHF-nb .. OHF-1 (nb = fix_nc - fix_pad)
  const float* fix2; int fix_nc, fix_pad;
  int cls_n;            // > 0: MERGED PARITY CLASSES of a stride-2 input gradient whose classes share one tap window (k = 6, pad 2:
                        // every class reads dy rows / columns -1..1): ONE problem with N = 4 * cls_n columns, column n = class
                        // (n / cls_n) channel (n % cls_n), class (ph, pw) = (c >> 1, c & 1) lands on output pixel
                        // (OS * oy + ph, OS * ox + pw): the dY tile is staged once for all four classes and an A fragment feeds
                        // 4x the MFMAs (tile kernel only)
  int accum;            // out_f32 target is an ACCUMULATOR (sv_conv2d_nhwc_dgrad's dx_f32_atomic contract: several producers add into one zeroed
                        // fp32 buffer -- the fan-in of SPLIT-GMVAE's encoder, vae/model.py:127-133): a launch with splitk == 1 (no atomics:
                        // small problems, and every problem under SV_DETERMINISTIC) adds with a plain read-modify-write instead of assigning
  int adj;              // input gradient of a layer whose input is a 2x bilinear upsample, FUSED with the resize adjoint: `out`
                        // is the LOW-RES gradient [B, OHF/2, OWF/2, ldo], `mask` the low-res activation (ReLU gate, may be
                        // null).  Only the row-ring kernel implements it (SV_E_UNSUPPORTED elsewhere)
  int8_t dy[SV_MAX_TAPS];
  int8_t dx[SV_MAX_TAPS];
};
// cfg: 0 = 128x128 tile, 1 = 128x64, 2 = 256x32, 3 = 256x16
int svk_tap_gemm(const TapGemmArgs& a, int dtype, int cfg, hipStream_t st);
// n <= SV_TAP_MAX_MULTI independent problems (any shapes, one tile configuration) in one launch
#define SV_TAP_MAX_MULTI 4
struct TapGemmMulti {
  TapGemmArgs a[SV_TAP_MAX_MULTI];
  int zbase[SV_TAP_MAX_MULTI];   // first blockIdx.z of each problem (its split-K slices follow)
  int n;
};
int svk_tap_gemm_multi(const TapGemmArgs* a, int n, int dtype, int cfg, hipStream_t st);

// ---- direct conv with the input tile resident in LDS (tile_conv.hip); planned from a TapGemmArgs
struct TileConvArgs {
  const void* A; const void* Wt; const float* bias; void* out; const void* mask;
  int B, IH, IW, lda;
  int cl2, P, Ktot, S, SX;
  int lTW, lTH, lNB;          // log2 tile width / height (iteration-grid pixels) / images per tile
  int OY, OX, tilesX, tilesY, ntiles;
  int TIW, TIH, y_lo, x_lo;   // LDS input tile extent (pixels) and the tap-offset origin
  int PS;                     // bytes per pixel in the LDS tile
  int plane_bytes;            // > 0: planar tile (tile_stage.hip.h), PS = 32
  int nph, lnph;              // channel phases of the tile (1 << lnph); 1 = the whole pixel at once
  int xcd_chunk;              // > 0: XCD-aware tile order: workgroup id w runs tile (w & 7) * xcd_chunk + (w >> 3)
  int off_bytes, in_bytes;    // LDS carve: piece-offset table, input tile
  int buf_bytes;              // persistent kernel: bytes of each of its two input-tile buffers
  int wslots;                 // weight-tile slots in LDS: 2 (ring, streamed per K step) or the number of K steps (whole K resident)
  int N, OHF, OWF, OS, ooy, oox, ldo, act, out_f32, ntaps;
  int dbg;                    // profiling ablation bits (SV_TC_DBG): 1 skip staging, 2 skip MFMA loop, 4 skip stores
  int ups;                    // input tile staged through the fused 2x bilinear upsample
  int d2s;                    // depth-to-space (x) epilogue of the pixel-packed conv: real channels per sub-pixel
  int cls_n;                  // merged parity classes (TapGemmArgs::cls_n): channels per class
  int d2s_y, clampin;         // TapGemmArgs::d2s_y / clampin
  int s2d3;                   // TapGemmArgs::s2d3
  int dma;                    // the input tile is staged by LDS-DMA (tile_stage.hip.h: stage_tile_plain_dma; fp32, plain / clamped input, planar or unpadded tiles)
  const float* fix;           // TapGemmArgs::fix
  const float* fix2; int fix_nc, fix_pad;   // TapGemmArgs::fix2 / fix_nc / fix_pad (per-class polyphase)
  const float* nll_img; void* nll_grad; float* nll_part; int nll_ch; float nll_gscale; int nll_noout;   // TapGemmArgs: fused loss
  int8_t dy[SV_MAX_TAPS];
  int8_t dx[SV_MAX_TAPS];
};
#define SV_MAX_MULTI 8
// process-wide side streams (streams.hip): index 0 / 1 = the plan's weight-gradient streams; the tape's lanes 1.. and the SPLIT-GMVAE step's second stream share them
#define SV_SHARED_STREAMS 3
hipStream_t sv_shared_stream(int k);
struct TileConvMulti { TileConvArgs a[SV_MAX_MULTI]; };   // kernel argument: blockIdx.z selects the problem
bool svk_tile_conv_plan(const TapGemmArgs& t, int dtype, int B, TileConvArgs* a, int* cfg_out);
int svk_tile_conv_multi(const TileConvArgs* a, int n, int dtype, int cfg, hipStream_t st);
int svk_conv_dispatch_multi(const TapGemmArgs* t, int n, int dtype, int tap_cfg, hipStream_t st);
int svk_tile_conv(const TileConvArgs& a, int dtype, int cfg, hipStream_t st);
// weight-stationary row-ring kernel (row_conv.hip): SV_E_UNSUPPORTED when the shape has no instantiation
int svk_row_conv_try(const TapGemmArgs* t, int n, int dtype, hipStream_t st);
bool svk_row_conv_supported(const TapGemmArgs* t, int n, int dtype);   // would svk_row_conv_try launch? (nothing is launched)
// picks the direct kernel when the problem fits it, the im2col tap GEMM otherwise
int svk_conv_dispatch(const TapGemmArgs& t, int dtype, int tap_cfg, hipStream_t st);

// ---- weight gradient: dW[(t,ci)][co] += sum_m A[pix(m)+tap t, ci] * dY[m, co]; dbias += colsum(dY)
struct WgradArgs {
  const void* A;      // [B, IH, IW, lda]
  const void* dY;     // [M, ldy]
  float* dW;          // [ntaps*Cin_real][N] fp32, atomically accumulated
  float* dbias;       // [N] or null
  int M, lOY, lOX, OY, OX, IH, IW, lda, S, SX;   // SX: x stride (= S except for the x-packed conv); lOY / lOX = -1: not a power of two
  int fold_kw, fold_c;   // x-packed conv (svg_packx): dW'[(ky,tx)][ci][px*8+co] folds into dW[ky][tx-px][ci][co], co < fold_c
  int ldy, ycols;     // dY row stride and number of valid columns from the dY pointer (multiple of 8)
  int cl2;            // log2(pieces per tap)
  int Cin_pad, Cin_real, N, Nrows;   // Nrows = ntaps*Cin_pad (padded wrow count)
  int ntaps, msplit;  // msplit = rows of m per blockIdx.z slice (multiple of the m-step)
  int ups;            // A is the low-res tensor, the layer input is its 2x bilinear upsample (tile kernel only)
  int clampin;        // input coordinates outside the image clamp to the edge (polyphase weight gradient; tile kernel only)
  int dy_s2d;         // dY is the hi-res gradient [B, 2*OY, 2*OX, 8] read as its space-to-depth view [B, OY, OX, 32]
  int dy_os, dy_oy, dy_ox;   // dy_os = 2: dY is the hi-res gradient [B, 2*OY, 2*OX, ldy] and iteration pixel (oy, ox) reads its pixel (2 oy + dy_oy, 2 ox + dy_ox):
                             // one parity class of the per-class polyphase weight gradient (conv_geom.h: svg_polyc; wgrad_tile_f32.hip only)
  int assign;         // slab path: the reduce WRITES dW (every element has one owner) instead of adding to it
  int s2d3;           // TapGemmArgs::s2d3 (A = the padded RGB tensor read through its space-to-depth view; dW lands in the [6][6][3][N] layout)
  float* ws;          // optional partial-sum workspace for the two-stage (deterministic) flush of the tile kernel
  int64_t ws_bytes;
  hipEvent_t ev_mid[2];   // profiling: when set, both are recorded after the main kernel, before the slab reduce
  struct WgradReduceDesc* defer; int* n_defer;   // slab path: do NOT launch the reduce, append its descriptor here (svk_wgrad_reduce_all later;
                                                 // the slab workspace must stay untouched until then)
  int8_t dy[SV_MAX_TAPS];
  int8_t dx[SV_MAX_TAPS];
};
// cfg: 0 = 64 wrows x 128 cols, 1 = 128 x 64, 2 = 256 x 32, 3 = 256 x 16
int svk_wgrad(const WgradArgs& a, int dtype, int cfg, hipStream_t st);
// n <= SV_WGRAD_IM2COL_MAX_MULTI independent problems (any shapes, one tile configuration) in one launch
#define SV_WGRAD_IM2COL_MAX_MULTI 4
struct WgradMulti {
  WgradArgs a[SV_WGRAD_IM2COL_MAX_MULTI];
  int zbase[SV_WGRAD_IM2COL_MAX_MULTI];   // first blockIdx.z of each problem (its m-splits follow)
  int plain[SV_WGRAD_IM2COL_MAX_MULTI];   // the problem has one m-split: dW += acc without atomics (each element has one owner)
  int n;
};
int svk_wgrad_multi(const WgradArgs* a, int n, int dtype, int cfg, hipStream_t st);

// ---- weight gradient with LDS-resident input/dY tiles (wgrad_tile.hip, bf16); planned from WgradArgs
struct WgradTileArgs {
  const void* A; const void* dY; float* dW; float* dbias;
  float* slab; float* ws; int64_t ws_bytes;   // slab = ws when the two-stage flush is used
  float* bslab;                               // with slab: [msplit][128] bias partials behind the dW slabs (summed by the reduce kernel)
  struct WgradReduceDesc* defer; int* n_defer;   // see WgradArgs
  int B, IH, IW, lda, cl2, S, SX, fold_kw, fold_c;
  int layer_id;             // instantiation id (tuning table of svk_wgrad_tile_multi)
  int contig;               // tiles of a workgroup: contiguous run (1) or strided by the grid (0)
  int pairx;                // 8-channel pixels (e1): rows 8..15 of a fragment are the NEXT pixel = the next x tap;
                            // the tap list holds every other x tap and accumulator row r means (tap 2u + (r>>3), channel r&7)
  int lTW, lTH, lNB, OY, OX, tilesX, tilesY, ntiles;
  int TIW, TIH, y_lo, x_lo, PS;
  int ldy, YS, lycp;        // dY channels per pixel; bytes per dY pixel in LDS; log2(16-B pieces per dY pixel)
  int in_bytes, dy_bytes;
  int dbg;                  // ablation: 1 = skip the atomic flush
  int ups;                  // fused 2x bilinear upsample of the input
  int clampin, dy_s2d, assign;   // WgradArgs::clampin / dy_s2d / assign
  int dy_os, dy_oy, dy_ox;       // WgradArgs::dy_os / dy_oy / dy_ox
  int s2d3;                      // WgradArgs::s2d3
  int dma;                       // fp32 kernel: tiles staged by LDS-DMA (wgrad_tile_f32.hip)
  int db;                        // ... into two LDS buffers: tile t + 1 in flight beside the MFMAs of tile t
  int CW, ncg;              // input-channel slice width per workgroup and number of slices (cl2 = log2(CW/8))
  int Cin_real, N, ntaps;
  int8_t dy[SV_MAX_TAPS];
  int8_t dx[SV_MAX_TAPS];
};
// one deferred slab reduce (one problem of one layer): everything wgrad_reduce_kernel needs, as run-time values
struct WgradReduceDesc {
  const float* slab; float* dW; const float* bslab; float* dbias;
  int msplit, groups, ncg, CW, Cin_real, N, ntaps, fold_kw, fold_c, pairx, assign, TPW, CIF, COF;
  int s2d3;           // dw_index's space-to-depth map (last member: the positional initialisers of the other kernels leave it 0)
};
#define SV_WGRAD_DEFER_MAX 16
struct WgradReduceAll { WgradReduceDesc d[SV_WGRAD_DEFER_MAX]; int first[SV_WGRAD_DEFER_MAX + 1]; int n; };   // first: block ranges of the flat grid
int svk_wgrad_reduce_all(const WgradReduceDesc* d, int n, hipStream_t st);   // every pending reduce in ONE launch (blockIdx.z = descriptor)
#define SV_WGRAD_MAX_MULTI 2
// rolling-window form for the 6x6 conv over an upsampled 64-channel input (d4; wgrad_roll.hip): needs the slab workspace
bool svk_wgrad_roll_supported(const WgradArgs* w, int n);
int svk_wgrad_roll_multi(const WgradArgs* w, int n, hipStream_t st);
// rolling-window form of the polyphase main term of the head's weight gradient (d5; wgrad_p5.hip): needs the slab workspace
bool svk_wgrad_p5_supported(const WgradArgs* w, int n);
int svk_wgrad_p5_multi(const WgradArgs* w, int n, hipStream_t st);
// e1's weight gradient, one pipeline per wave (wgrad_e1.hip): needs the slab workspace
bool svk_wgrad_e1_supported(const WgradArgs* w, int n);
int svk_wgrad_e1_multi(const WgradArgs* w, int n, hipStream_t st);
// e2's weight gradient, rolling window over the space-to-depth input (wgrad_e2.hip): needs the slab workspace
bool svk_wgrad_e2_supported(const WgradArgs* w, int n);
int svk_wgrad_e2_multi(const WgradArgs* w, int n, hipStream_t st);
struct WgradTileMulti { WgradTileArgs a[SV_WGRAD_MAX_MULTI]; };     // blockIdx.z selects the problem
struct WgradReduceMulti { const float* slab[SV_WGRAD_MAX_MULTI]; float* dW[SV_WGRAD_MAX_MULTI]; const float* bslab[SV_WGRAD_MAX_MULTI]; float* dbias[SV_WGRAD_MAX_MULTI]; };
// fp32 (the reference's precision): LDS tiles + v_mfma_f32_16x16x4_f32 + the same slabs (wgrad_tile_f32.hip); SV_E_UNSUPPORTED -> im2col kernel
int svk_wgrad_tile_f32_multi(const WgradArgs* w, int n, hipStream_t st);
// the four parity classes of n <= 2 per-class polyphase layers in one launch (cls[c * n + i]); appends the 4 n slab-reduce descriptors
int svk_wgrad_polyc_f32_multi(const WgradArgs* cls, int n, int mask, WgradReduceDesc* rd, int* nrd, hipStream_t st);
int svk_wgrad_tile(const WgradArgs& w, hipStream_t st);   // SV_E_UNSUPPORTED -> use svk_wgrad
int svk_wgrad_tile_multi(const WgradArgs* w, int n, hipStream_t st);   // n twin layers, one launch (own ws each)
int svk_wgrad_dispatch_multi(const WgradArgs* w, int n, int dtype, int cfg, hipStream_t st);
// tile kernel when the layer has an instantiation (bf16), im2col kernel otherwise
int svk_wgrad_dispatch(const WgradArgs& w, int dtype, int cfg, hipStream_t st);
#define SV_WGRAD_WS_BYTES (512LL * 36 * 4 * 256 * 4 + 512LL * 128 * 4)   // 512 workgroups x 36 fragments x 4 waves x 256 floats + their bias partials

// ---- the latent block's GEMMs (latent_gemm.hip): out [M, N] = A [M, K] . W^T, A and the prepared image W [N, K] K-contiguous
struct NtGemmProb {
  const void* A; int lda;         // bf16 [M][lda]
  const void* W; int ldw;         // bf16 [N][ldw] (a prepared forward / input-gradient image)
  void* out; int ldo;             // bf16 [M][ldo], or (out_f32) the fp32 slabs [splitk][M][ldo] of the K slices
  const float* bias;              // [N] or null (bf16 output only)
  const void* mask;               // bf16 [M][ldo] or null: out = mask > 0 ? out : 0
  int M, N, K, act, splitk, out_f32;
  int f32;                        // operands / output / mask are float instead of bf16
  int64_t slab_stride;            // floats between K slices
  int zbase;                      // first blockIdx.z of this problem (filled by svk_nt_gemm_multi)
};
struct NtGemmMulti { NtGemmProb p[2]; int n; };
struct NtReduceMulti { const float* slab[2]; float* out[2]; int S[2]; int64_t stride[2]; int64_t count[2]; };
bool svk_nt_gemm_supported(const NtGemmProb& p);
int svk_nt_gemm_pick_splitk(int M, int N, int K, int nprob);
int svk_nt_gemm_multi(NtGemmProb* p, int n, int bm, hipStream_t st);
int svk_nt_slab_reduce(const NtGemmProb* p, float* const* out, int n, hipStream_t st);

// dW [Kw, N] = X^T . dY, dbias [N] = colsum(dY): X [M, ldx] (columns [0, Kw) used; Kw_real <= Kw rows of dW are stored), dY [M, ldy]
struct TnWgradProb { const void* X; int ldx; const void* dY; int ldy; float* dW; float* dbias; int M, Kw, Kw_real, N; int f32; };
struct TnWgradMulti { TnWgradProb p[4]; };
bool svk_tn_wgrad_supported(const TnWgradProb& p);
int svk_tn_wgrad_multi(const TnWgradProb* p, int n, hipStream_t st);

// ---- batched weight preparation (fp32 HWIO master -> MFMA-ready images), job table in device memory
struct PrepJob {
  int64_t src_off;     // element offset into the flat fp32 parameter buffer
  int64_t dst_off;     // element offset (of dtype) into the prepared-weight arena
  int32_t ntaps;       // taps in the destination image
  int32_t Cin, Cout;   // real channel counts of the HWIO master
  int32_t rows, inner; // destination [rows][ntaps][inner] ...
  int32_t inner_ld, inner_off; // ... stored with row pitch inner_ld at column offset inner_off
  int32_t transpose;   // 0: rows=co, inner=ci (forward); 1: rows=ci, inner=co (dgrad)
  int32_t packx_kw;    // > 0: forward image of the x-packed conv of a KH x packx_kw kernel: row n = px*8 + co,
                       // tap (ky, tx) of KH x (KW+1) <- source tap (ky, tx - px), zero outside the kernel
  int32_t poly;        // 1: polyphase forward image of (2x bilinear upsample -> 6x6 conv) [32][25][Cin]; 2: its border-fix image
                       // [10 classes][6 taps][16][Cin] (conv_api.hip: prep_poly)
                       // 3: per-class polyphase image [rows][nty*ntx (x-major)][Cin] of class pcls of a pk x pk kernel (conv_geom.h: svg_polyc)
                       // 4: its border-class image [2*(pk-1) classes][pk taps][rows][Cin] = -(sum over the taps that leave the image)
                       // 5: main image of the polyphase INPUT gradient [rows = ci][(2R+1)^2 hi-res taps (x-major)][co]; 6: its edge images
                       // [4 edges: top, bottom, left, right][4 rows from the edge][2R+1][ci][co]; 7: its corner images [4 corners][4][4][ci][co] (conv_geom.h: svg_polyd)
  int32_t pk, pcls;    // poly 3 .. 7: kernel size; poly 3: parity class py*2 + px
  int32_t s2d3;        // forward image of the space-to-depth form (svg_s2d3): [co][t = tx*3 + ty][16] <- the 6 x 6 x 3 master
  int32_t first_block; // first block of this job in the launch
  int32_t nblocks;
  uint8_t srctap[SV_MAX_TAPS];  // destination tap -> source (kh*KW+kw) tap
};
int svk_prep_weights(const float* params, void* arena, int dtype, const PrepJob* jobs_dev, int njobs,
                     int nblocks, hipStream_t st, int block0 = 0);

// border correction of the polyphase forward (poly_fix.hip): the out-of-image taps of the 5 hi-res border rows / columns.
// fixbuf[i] != null: written to the workspace BEFORE the conv, whose epilogue adds it (TapGemmArgs::fix); else added to out6[i]
// with atomics AFTER the conv.  n <= 2 twin problems per launch.
int64_t svk_poly_fix_ws_bytes(int B, int h, int w);
int svk_poly_fix_multi(int n, const void* const* x_lo, const void* const* wfix, float* const* out6, float* const* fixbuf, int B,
                       int h, int w, int lda, int Cout, hipStream_t st, int dtype = 1 /* SV_BF16 */);
// the same for the per-class form (svg_polyc): any K in {4, 6}, Cout a multiple of 16; fixrow [B][K-1][2w][Cout], fixcol [B][2h][K-1][Cout]
int svk_polyc_fix_multi(int n, const void* const* x_lo, const void* const* wfix, float* const* fixrow, float* const* fixcol, int B, int h, int w,
                        int Cin, int Cout, int K, int dtype, hipStream_t st);

// polyphase weight gradient of the decoder head, the small terms (poly_wgrad.hip)
int64_t svk_poly_wgrad_ws_floats(int Cin, int nwg);
bool svk_poly_wgrad_supported(int h_lo, int w_lo, int Cin, int Cout);   // checked before the main term is launched
int svk_poly_wgrad_finish(int n, const void* const* x_lo, const void* const* dy, float* const* ws, float* const* dW, float* const* dbias,
                          int B, int h, int w, int lda, int Cin, int Cout, int nwg, hipStream_t st);

int svk_split_pad(const float* images6, void* x8, void* xh8, int dtype, int64_t npix, hipStream_t st);
int svk_finalize_losses(const float* nll_x, const float* nll_xh, const float* kl_x, const float* kl_xh,
                        int B, float beta, float* losses, float* metric_acc, int accumulate, hipStream_t st,
                        const float* part_x = nullptr, const float* part_xh = nullptr, int P = 0);   // part_*: per-tile NLL partials (fused loss)

// Step-varying scalars of a captured step (hipGraph replay, lgvae_plan.hip): the kernels that consume them read this
// device record instead of their launch arguments when `dyn` is non-null; svk_set_dyn writes it before each replay.
struct SvDynArgs {
  uint64_t seed, step;
  int64_t sample_offset;
  float adam_alpha;       // lr * sqrt(1 - beta2^t) / (1 - beta1^t)
  float pad;
};
// both networks' heads in one launch (index 0 = x, 1 = x-hat; Philox stream id = the index)
int svk_reparam_kl_fwd_twin(const float* const* pre, const float* const* bias_mean, const float* const* bias_sd,
                            const float* const* eps, float* const* eps_out, float* const* z_mean, float* const* z_sig,
                            float* const* z, void* z_lp, int z_dtype, int ldz, const int* z_col, float* const* kl, int B,
                            const int* L, uint64_t seed, uint64_t step, int64_t sample_offset, hipStream_t st,
                            const SvDynArgs* dyn = nullptr, const int* S = nullptr, const int64_t* slab_stride = nullptr);   // S: `pre` = K-slice slabs, summed in the kernel
int svk_reparam_kl_bwd_twin(const float* const* dz, const int* ld_dz, const float* const* dz2, const int* ld_dz2,
                            const float* const* z_mean, const float* const* z_sig, const float* const* eps, float kl_scale,
                            void* const* g_pre, int g_dtype, int B, const int* L, hipStream_t st,
                            const int* S = nullptr, const int64_t* stride = nullptr, const int* S2 = nullptr, const int64_t* stride2 = nullptr);   // S: dz / dz2 = K-slice slabs
int svk_set_dyn(SvDynArgs* dyn, uint64_t seed, uint64_t step, int64_t sample_offset, float adam_alpha, hipStream_t st);
double svk_adam_alpha(float lr, float beta1, float beta2, int64_t t);
int svk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int64_t t, float grad_scale, const SvDynArgs* dyn, hipStream_t st);
int svk_reparam_kl_fwd2(const float* pre, const float* bias_mean, const float* bias_sd, const float* eps,
                        float* eps_out, float* z_mean, float* z_sig, float* z, void* z_lp, int z_dtype, int ldz,
                        int z_col, float* kl, int B, int L, uint64_t seed, uint64_t step, int stream_id,
                        int64_t sample_offset, hipStream_t st, const SvDynArgs* dyn = nullptr);
// per-image sums of P partials (fixed order): the tail of svk_dlogistic_nll_multi, also used after the fused loss epilogue
int svk_nll_rowsum(const float* partial_ws, float* nll, int B, int P, int64_t zs_part, int64_t zs_nll, int nets, hipStream_t st);
int svk_dlogistic_nll_multi(const float* images6, int ch_off, const float* out6, int64_t zs_out, float* nll,
                            int64_t zs_nll, void* grad, int64_t zs_grad, int grad_dtype, float grad_scale, int B,
                            int H, int W, float* partial_ws, int64_t zs_part, int nets, hipStream_t st);
