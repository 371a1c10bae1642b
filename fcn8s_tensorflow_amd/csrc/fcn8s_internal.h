// Internal launcher interface shared by the HIP translation units.
// gfx950 only.  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fcn8s_hip.h"      // the status codes (defer_error)

// gfx950 (MI355X, CDNA4) only: the kernels rely on 160 KB of LDS per workgroup (conv_bf16_256_kernel: five 32 KB stages; wino_out_in_kernel: 96 KB
// static), on global_load_lds_dwordx4, ds_read_b64_tr_b16 and the gfx950 MFMA shapes.  A device pass for any other target stops here with a
// readable message instead of "local memory limit exceeded" deep inside a template; fcn8s_create checks the same thing at run time.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libfcn8s_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
#define FCN8S_LDS_BYTES_NEEDED (160u * 1024u)

#include <string>
namespace fcn8s {

// symbol of the MFMA kernel the last launch_* call used (for the per-kernel profile view)
extern thread_local const char* g_last_kernel;
// A launcher that cannot run -- a shape no kernel of the chosen arithmetic takes, a scratch allocation that failed, a broken internal promise -- records why and
// returns WITHOUT launching; the C entry point that called it (fcn8s_forward_loss, fcn8s_backward_bucket, fcn8s_predict, ..., every fcn8s_op_*) then returns `code`
// with the text in fcn8s_last_error instead of the process dying in abort() (in a data-parallel run an aborted rank also cost its peers a full watchdog timeout).
// The first error wins; the slot belongs to the calling thread.
void defer_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int take_deferred_error(std::string* text);       // the recorded code (0: none) and its text; clears the slot

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution on the f32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   Y[m, j] = epilogue( alpha * sum_{tap, c} X[src(m, tap), c] * W[tap*Cin + c, j] )
//
// m runs over an (N, Ma, Mb) grid.  src(m,tap) = (n, in_scale*a + dy, in_scale*b + dx),
// (dy,dx) = (ty,tx)*tap_step + tap_off, zero outside [0,Hi)x[0,Wi).
// The output row of m is pixel (n, out_scale*a + out_offy, out_scale*b + out_offx)
// of an (N,Ho,Wo,ldy) tensor; rows falling outside are not stored.
// gridDim.z enumerates "phases" (transposed-conv sub-pixel phases): phase z uses
// weights w + z*w_phase_stride and adds (z / phases_x, z % phases_x) to the output offsets.
// One parameterisation therefore covers: 3x3 / 7x7 / 1x1 SAME convs, their data
// gradients (flipped+transposed weights), the strided conv that is the data
// gradient of a transposed conv, and the transposed conv itself (s*s phases of 2x2 convs).
// ---------------------------------------------------------------------------
struct IgemmArgs {
    const float* x; const float* w; const float* bias; const float* addend; const float* mask;
    float* y;
    int N, Ma, Mb; long long M;
    int Hi, Wi, Cin, ldx;
    int KW, in_scale, tap_step, tap_off, Ktot;
    int Ho, Wo, Cout, ldy;
    int out_scale, out_offy, out_offx;
    int phases_x; long long w_phase_stride;
    float alpha; int relu; float mask_scale;
    int dropout; float keep_prob; unsigned long long seed; unsigned int stream_id;
    int batched; long long x_batch_stride, y_batch_stride;   // gridDim.z independent GEMMs (Winograd positions)
    int m_fastest;                                           // tile order, set by the launcher (see igemm.hip)
    int bt, ldw;                                             // bt: the B operand is stored transposed, w[z][n][k] with row stride ldw (plain batched GEMMs on the LDS-DMA kernel only)
    int split;                                               // 3: the LDS-DMA kernels take their products as six bf16 MFMAs of the operands split in three (FCN8S_PREC_F32X3); 2: three MFMAs of two pieces (FCN8S_PREC_F32X2); 0: f32 MFMA
};
void launch_igemm(const IgemmArgs& a, int phases, hipStream_t s);

// ---------------------------------------------------------------------------
// bf16-operand / fp32-accumulate SAME conv (stride 1, odd K) on v_mfma_f32_32x32x16_bf16 -- the optional fc6 / fc7
// precision mode (gemm_bf16.hip).  x: fp32 NHWC; wt: bf16 K-tile-major blocks written by launch_w_to_bf16_tiles from the
// fp32 HWIO kernel; y: fp32.  Needs Cin % 32 == 0, Cout % 128 == 0 (launch returns false otherwise).
// ---------------------------------------------------------------------------
struct Bf16ConvArgs {
    const float* x; const unsigned short* xh;   // activations: fp32, or (xh != nullptr) already converted by launch_f32_to_bf16
    const unsigned short* wt; const float* bias; float* y;
    int N, H, W, Cin, Cout, K;
    int relu, dropout; float keep_prob; unsigned long long seed; unsigned int stream_id;
    long long M; int m_fastest;        // filled in by the launcher
};
void launch_w_to_bf16_tiles(const float* w, unsigned short* wt, int K, int Cout, hipStream_t s);   // w[K][Cout] -> wt[K/32][Cout][32]
bool launch_conv_bf16(const Bf16ConvArgs& a, hipStream_t s);
void launch_f32_to_bf16(const float* x, unsigned short* y, long long n, hipStream_t s);                // n % 8 == 0
// 256 x 256 tile variant (gemm_bf16.hip): zero-padded bf16 activations [N][H + K - 1][W + K - 1][Cin], kernel transposed to wt[Cout][K*K*Cin] bf16
struct Bf16Conv256Args {
    const unsigned short* xp; const unsigned short* wt; const float* bias; float* y;
    int N, H, W, Cin, Cout, K;
    int relu, dropout; float keep_prob; unsigned long long seed; unsigned int stream_id;
    long long M; int m_fastest;        // filled in by the launcher
    const float* addend;               // optional, [M][Cout]: added before ReLU / mask (a data gradient's skip-path term)
    const float* mask; float mask_scale;   // optional, [M][Cout]: y = mask > 0 ? y * mask_scale : 0 (the ReLU / dropout of the layer whose input gradient this is)
    int any_shape;                     // 1: Cout % 64 == 0 and any M are taken (64- / 128-column tiles, partial last row tile); 0: the round-3 rule
    const unsigned short* mask16;      // optional (flat-position kernel only): the padded bf16 copy [N][H + 2][W + 2][Cout] whose sign is the mask, instead of `mask`
    float* colpart;                    // optional (flat-position kernel only): [row tiles][Cout] column sums of the tile's stored values, one row per row tile
    int ksplit; float* part;           // set by the launcher: K split over blockIdx.y into `ksplit` slabs of raw accumulators at `part`
    int rows_bn;                       // (A/B) 128: the flat-position kernel takes its 128-column form where Cout % 128 == 0 (default: 64 columns everywhere)
    int guarded;                       // 1: xp has zeroed guard rows in front and behind (>= W + 3 + 16 rows of Cin): the flat-position kernel may be taken
    unsigned short* yb; int yb_pad;    // optional: also write the output as bf16 into the interior of a padded copy [N][H + 2 yb_pad][W + 2 yb_pad][Cout] (points at its pixel 0)
    // Channel-chunk PLANES (bf16_train's per-layer copies): a padded copy may be stored as [C / 32][rows][32] instead of [rows][C] -- the 64-byte slice (one
    // position, 32 channels) that is one LDS row of a K-tile then lies next to its neighbour positions' slices and an LDS-DMA instruction reads 1 KB of contiguous
    // memory, whole 128-byte lines, instead of sixteen half lines.  *_ps = elements between two planes ((rows + guard rows) * 32); 0 = [rows][C].  The pointers
    // point at padded pixel 0 of plane 0.
    long long xp_ps, yb_ps, mask16_ps;
};
bool conv_bf16_256_ok(long long M, int Cin, int Cout, int mode);      // mode 0 never, 1 when it fills the chip, 2 whenever the shapes allow
void launch_w_to_bf16_t(const float* w, unsigned short* wt, int K, int Cout, hipStream_t s);
void launch_f32_to_bf16_padded(const float* x, unsigned short* xp, int N, int H, int W, int C, int pad, hipStream_t s, long long ps = 0);      // ps: plane stride of xp in elements, 0 = [rows][C]
int conv_bf16_rows_bm(int Cout, int rows_bn);      // positions per row tile (= per partial row of `colpart`) of the form launch_conv_bf16_256 picks
bool launch_conv_bf16_256(const Bf16Conv256Args& a, hipStream_t s);
// the padded bf16 copy of an output gradient (interior only: the border of xp is zero already) with db[c] += column sums of x on the way
bool launch_f32_to_bf16_padded_colsum(const float* x, unsigned short* xp, float* db, int N, int H, int W, int C, int pad, hipStream_t s, long long ps = 0);
// the kernel of a SAME convolution's data gradient as conv_bf16_256_kernel wants it: wt[Cin][(flipped taps, Cout)] bf16 (Cout % 8 == 0)
void launch_w_to_bf16_flip_t(const float* w, unsigned short* wt, int K, int Cin, int Cout, hipStream_t s);
// weight gradient with bf16-rounded operands: dW[tap][ci][co] = sum_q A[q + off(tap)][ci] * B[q][co] over R flat padded-pixel rows (gemm_bf16.hip)
struct Bf16WgradArgs {
    const unsigned short* A; const unsigned short* B; float* C;      // A = padded bf16 input [R][Ci], B = padded bf16 output gradient [R][Cj], both with guard rows
    long long R; int Ci, Cj, K, Wp;                                   // R = N * Hp * Wp, Wp = W + K - 1
    long long chunk, split_stride; int plain_store;                   // set by the launcher
    long long a_ps, b_ps;                                             // channel-chunk planes (see Bf16Conv256Args): elements between two 32-channel planes of A / B, 0 = [R][C]
};
bool launch_wgrad_bf16(const Bf16WgradArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------
// Weight-gradient GEMM on the f32 MFMA:
//   C[tap][i][j] += alpha * sum_p A[srcA(p, tap), i] * B[p, j]
// p runs over the (N, Pa, Pb) pixel grid of B; srcA = (n, a_scale*a + dy, a_scale*b + dx).
// Split over p across gridDim.y with float atomics into a zero-initialised C.
// ---------------------------------------------------------------------------
struct WgradArgs {
    const float* A; const float* B; float* C;
    int N, Pa, Pb; long long P;
    int Ha, Wa, Adim, lda, Areal;      // A tensor (N,Ha,Wa,lda); Adim channels loadable (mult of 4); Areal rows stored
    int Bdim, ldb;                     // B tensor (N,Pa,Pb,ldb); Bdim channels (mult of 4)
    int KW, a_scale, tap_off, ntaps;   // dy = ty + tap_off
    int ldc;                           // C[tap] is [Areal][ldc]
    float alpha;
    float* colsum;                     // optional: colsum[j] += sum_p B[p,j]  (bias gradient), or nullptr
    int batched; long long a_batch_stride, b_batch_stride;   // gridDim.z = independent GEMMs (C stride = Areal*ldc)
    int c_uninitialized;               // 0: C was zeroed by the caller (accumulate with atomics); 1: the launcher decides (plain store or memset)
    int plain_store;                   // set by the launcher
    int split;                         // as IgemmArgs::split
    long long split_stride;            // set by the launcher (deterministic mode): != 0 = reduction split k stores its partial tile at C + k * split_stride
};
void launch_wgrad(const WgradArgs& a, hipStream_t s);

// ---- deterministic mode (model option "deterministic") ---------------------------------------------------------------
// Every reduction that the default path splits over blocks and joins with fp32 atomics (weight gradients, bias gradients: the order of the
// additions then depends on block scheduling, so no two runs give the same bits) instead stores one partial result per split into a scratch slab
// and a second kernel adds the slabs in split order.  Set per thread by the model's entry points; launchers read it.
extern thread_local int t_deterministic;
float* det_scratch(hipStream_t s, size_t floats);              // per-(device, stream) scratch, grown on demand; nullptr if the allocation failed
float* scratch2(hipStream_t s, size_t floats);                 // a second such buffer (the column sums' partials, which run beside a det_scratch user)
void scratch_release(hipStream_t s);                           // frees both buffers of the calling device's stream s (fcn8s_destroy; the stream is idle)
// C[r][c] (= or +=) sum_{k < nsplit, in order} ws[k * slab + r * ldc + c]   for r < rows, c < cols
void launch_det_reduce(float* C, const float* ws, long long rows, int cols, int ldc, long long slab, int nsplit, bool accumulate, hipStream_t s);
// 3x3 / 7x7 SAME conv weight (+bias) gradient, several taps per block (3x3: all nine, 7x7: one filter row);
// returns false if the shape is not covered (K, Cin % 64, Cout % 64, W % 16): the caller then uses launch_wgrad.
bool launch_wgrad_taps(const float* X, const float* dZ, float* dW, float* db, int N, int H, int W, int Cin, int Cout,
                       int K, hipStream_t s);

// conv1_1 (3 -> 64 channels, 4-channel padded input) weight + bias gradient, VALU, HBM-bound;
// returns false if the shape is not covered (Cout != 64 or W % 64).
bool launch_conv1_wgrad(const float* X4, const float* dZ, float* dW, float* db, int N, int H, int W, int Cout, int mfma, hipStream_t s);

// conv1_1 forward with bias + ReLU on the LDS-DMA gather kernel (Cout = 64 only; returns false otherwise): x4 [N,H,W,4], w48 [48][64]
// (taps x 4 channels, rows 36..47 zero), zero16 = 16 zero bytes in device memory (what taps outside the image read)
bool launch_conv1_fwd(const float* x4, const float* w48, const float* bias, float* y, const float* zero16, int N, int H, int W, int Cout, int tiled, hipStream_t s, unsigned short* yb16 = nullptr, long long yb16_ps = 0);

// weight gradient of the 16x16 / stride-8 transposed conv with 20 channels (VALU; dW zero-initialised, accumulated with atomics);
// returns false for any other shape: the caller then uses launch_wgrad.
bool launch_tconv_wgrad(const float* X, const float* dY, float* dW, int N, int Hi, int Wi, int C, int K, int S, hipStream_t s);

// ---------------------------------------------------------------------------
// Element-wise / reduction kernels (HBM-bound)
// ---------------------------------------------------------------------------
void launch_preprocess(const void* img, int dtype, float* out4, long long npix, hipStream_t s);
void launch_maxpool_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t s);
void launch_maxpool_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C,
                        int relu_mask, hipStream_t s);
void launch_maxpool_route(const float* x, unsigned char* route, int N, int H, int W, int C, hipStream_t s);
void launch_maxpool_fwd_route(const float* x, float* y, unsigned char* r, int N, int H, int W, int C, hipStream_t s, unsigned short* yb16 = nullptr, int pad = 0, long long yb16_ps = 0, int round16 = 0);
// ... the same over the block's last activation kept only as its bf16 copy (channel-chunk planes of the zero-bordered map, plane stride xb_ps); yb16 likewise
void launch_maxpool_fwd_route16(const unsigned short* xb, long long xb_ps, unsigned char* r, int N, int H, int W, int C, hipStream_t s, unsigned short* yb16, int pad, long long yb16_ps);   // forward pool + those bytes in one pass
// Where the logits of pixel slot p live.  blocked == 0: slot = pixel, NHWC.  blocked != 0: the layout the last transposed conv
// (k = 2s, stride s, pad s/2) produces when it runs as ONE GEMM (model.hip: tconv_gemm_*): rows = (n, q, qx) over an (H/s + 1) x (W/s + 1)
// grid of s x s output blocks that start at pixel (s q - s/2, s qx - s/2), columns = (r, rx, class); slots of the half blocks that
// stick out of the image exist but are invalid (their gradient is written as 0).
struct PixMap { int blocked, H, W, QH, QW, S; };
static __host__ __device__ inline long long pixmap_slots(const PixMap& m, long long npix, int N) { return m.blocked ? (long long)N * m.QH * m.QW * m.S * m.S : npix; }
// partials: double[>= softmax_xent_blocks(npix)]
int  softmax_xent_blocks(long long npix);
void launch_softmax_xent(const float* logits, const uint8_t* labels, float* dlogits, double* partials,
                         long long npix, int C, float grad_scale, hipStream_t s, float* colsum = nullptr,   /* colsum[c] += sum_p dlogits[p,c] (zero-initialised by the caller) */
                         const PixMap* map = nullptr, int N = 0);
// loss_out[0] = sum(partials)/npix + 0.5*rate*regsum[0]
void launch_finalize_loss(const double* partials, int nparts, long long npix, const float* regsum,
                          float rate, float* loss_out, hipStream_t s);
void launch_softmax_argmax(const float* logits, float* softmax_out, long long* argmax_out,
                           long long npix, int C, hipStream_t s, const PixMap* map = nullptr, int N = 0);
// ---- the k = 2s transposed conv as one GEMM (see PixMap): operand / result re-layouts, all tiny next to the GEMMs -------------
// A[(n,q,qx)][(a,b,ci)] = x[n, q-1+a, qx-1+b, ci] (0 outside), row stride KP >= 4C (columns >= 4C zero)
void launch_tconv_im2col(const float* x, float* A, int N, int Hi, int Wi, int C, int KP, hipStream_t s);
// dx[n,i,j,ci] = sum_{a,b} dA[(n, i+1-a, j+1-b)][(a,b,ci)]
void launch_tconv_col2im(const float* dA, float* dx, int N, int Hi, int Wi, int C, int KP, hipStream_t s);
// w[K,K,Cout,Cin] (K = 2S) -> b2[4C][S*S*C] (forward B operand), b2t[S*S*C][KP] (data-gradient B operand, columns >= 4C zero), bias tiled to S*S*C
void launch_tconv_pack_gemm(const float* w, const float* bias, float* b2, float* b2t, float* bias_tiled, int C, int S, int KP, hipStream_t s);
// dw[K,K,Cout,Cin] += from db2[KP][S*S*C] (rows >= 4C ignored)
void launch_tconv_unpack_dw(const float* db2, float* dw, int C, int S, hipStream_t s);
void launch_unblock_logits(const float* blocked, float* nhwc, const PixMap& map, int N, int C, hipStream_t s);
void launch_onehot_to_ids(const void* oh, int elem_bytes, long long npix, int C, uint8_t* ids, int* bad, hipStream_t s);
void launch_confusion(const uint8_t* labels, const long long* pred, long long npix,
                      unsigned long long* conf, int C, hipStream_t s);
// skinny.hip: 1x1 convs with <= 32 output (forward, weight gradient) or input (data gradient) channels; false = shape not covered
bool launch_head_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int K, int C, float alpha, hipStream_t s,
                     float* scratch = nullptr, size_t scratch_floats = 0);      // scratch: lets a launch with few row blocks split K over more blocks (two-pass sum)
bool launch_head_dgrad(const float* dy, const float* wt, const float* mask, float mask_scale, float* dx, long long M, int K, int C, float alpha,
                       hipStream_t s);
bool launch_head_wgrad(const float* x, const float* dy, float* dw, long long M, int K, int C, float alpha, hipStream_t s);
// bf16_train: max-pool backward (ReLU fused) whose output is the padded bf16 copy of dZ (interior; border zero already) plus db[c] += column sums of dZ
bool launch_maxpool_bwd_bf16(const float* x, const float* dy, unsigned short* dzb, float* db, int N, int H, int W, int C, hipStream_t s, const unsigned char* route = nullptr, long long dzb_ps = 0);
void launch_colsum(const float* x, float* out, long long rows, int C, hipStream_t s);      // out[c] += sum_r x[r,c]
void launch_sumsq(const float* x, float* out, long long n, hipStream_t s);                  // out[0] += sum x^2
void launch_axpy(float* y, const float* x, float a, long long n, hipStream_t s);           // y += a*x
void launch_tf_adam(float* theta, const float* g, float* m, float* v, long long n,
                    float lr_t, float b1, float b2, float eps, float gscale, hipStream_t s);
void launch_sgd_momentum(float* theta, const float* g, float* buf, long long n,
                         float lr, float mom, float gscale, hipStream_t s);
// weight re-layouts (run once per step, tiny next to the convs)
void launch_flip_transpose(const float* w, float* wt, int taps, int Cin, int Cout, hipStream_t s); // wt[T-1-t][co][ci] = w[t][ci][co]
void launch_pad_cin(const float* w, float* w4, int taps, int Cin, int Cinp, int Cout, hipStream_t s);
void launch_tconv_phase_pack(const float* w, float* wp, int K, int S, int C, hipStream_t s);
void launch_dropout_mask(float* mask, long long n, float keep_prob, unsigned long long seed,
                         unsigned int stream_id, hipStream_t s);
// Winograd F(tile x tile, 3x3) transforms (tile = 2, 4 or 6) around P = (tile+2)^2 batched GEMMs; C % 4 == 0; H, W % tile == 0 for
// tile 2 (tiles 4 and 6 handle partial edge tiles; T = N * ceil(H/tile) * ceil(W/tile)).
// KS = 3: plain 3x3 conv.  KS = 7: the filter is cut into a 3x3 grid of 3x3 sub-filters whose products add up in the
// Winograd domain (GEMM depth 9*C); u / v rows are then [sub][channel].
int wino_r(int KS);                 // sub-filter size: 3 for 3x3 kernels; 7x7 (fc6): 4
int wino_nsub(int KS);              // sub-filters per dimension: ceil(KS / r)
int wino_alpha(int tile, int KS);   // tile + r - 1; the number of Winograd positions is alpha^2
long long wino_slab(long long T, int C);   // floats between the slabs of consecutive Winograd positions of a [P][T][C] tensor (T*C + skew)
void launch_wino_filter(int tile, const float* w, float* u, int Cin, int Cout, int KS, hipStream_t s, int transpose_out = 0);   // w[KS*KS][Cin][Cout] -> u[P][nsub*Cin][Cout]  (transpose_out: u[P][Cout][Cin], KS = 3 only)
void launch_wino_input(int tile, const float* x, float* v, int N, int H, int W, int C, int KS, hipStream_t s, unsigned* rbits_out = nullptr);   // x[N,H,W,C] -> v[P][T][nsub*C]
void launch_wino_input_conv1(const float* x0, const float* w4, const float* bias, float* v, int N, int H, int W, hipStream_t s, unsigned* rbits_out = nullptr);   // conv1_1 + conv1_2's F(6x6,3x3) input transform
void launch_wino_input_xb(const float* x, float* v, unsigned short* xb, int N, int H, int W, int C, hipStream_t s, unsigned* rbits_out = nullptr);   // F(6x6,3x3), + padded bf16 copy of x
// F(4x4,3x3) only: V = B^T dy B (input transform of the data-gradient conv) AND dM = A dy A^T (weight-gradient transform) from
// one read of dy; returns false if the shape is not covered.
// pidx != nullptr: `dy` is instead the gradient of the 2x2/2 max-pool output, [N,H/2,W/2,C], and pidx the per-window argmax bytes
// written by launch_wino_output -- the max-pool backward (with the ReLU mask) is applied on the fly and dy never exists in HBM.
bool launch_wino_input_dout(int tile, const float* dy, float* v, float* dm, int N, int H, int W, int C, hipStream_t s, const unsigned char* pidx = nullptr);
void launch_wino_output(int tile, const float* m, const float* bias, const float* addend, const float* mask, float mask_scale,
                        int relu, float* y, int N, int H, int W, int C, int dropout, float keep, unsigned long long seed,
                        unsigned int stream_id, hipStream_t s, float* pool = nullptr, unsigned char* pidx = nullptr, int KS = 3,
                        unsigned* rbits_out = nullptr, const unsigned* rbits_in = nullptr);                      // m[P][T][C] -> y[N,H,W,C]
// rbits_out: also record (y > 0) as one bit per element (wino_rbits_words() words); rbits_in: use such a record instead of `mask`
size_t wino_rbits_words(int tile, int N, int H, int W, int C);
// (pool != nullptr: also writes the 2x2/2 max-pool of y, [N,H/2,W/2,C] -- the tiles are aligned with the pool windows -- and,
//  if pidx != nullptr, one byte per pooled element: index 0..3 of the window's first maximum, 4 if that maximum is not > 0)
void launch_wino_dout(int tile, const float* dy, float* dm, int N, int H, int W, int C, hipStream_t s, int KS = 3, const unsigned char* pidx = nullptr);   // dy -> dm[P][T][C] = A dY A^T  (tile 6 with pidx: dy is d(pool), routed through the argmax bytes)
// F(6x6,3x3) data gradient as the adjoint of the forward algorithm: dv[P][T][C] = dM U^T -> dx = overlap-added B dv B^T (+ skip addend, ReLU mask)
void launch_wino_dgrad_output_dout(const float* dv, const unsigned* rbits_in, float* dm, int N, int H, int W, int C, hipStream_t s);   // ... followed in registers by the previous layer's dM = A dZ A^T (dZ is not written)
void launch_wino_dgrad_output_sub44(const float* dv, float* dx, int N, int H, int W, int C, hipStream_t s);   // fc6: see winograd.hip
void launch_wino_dgrad_output(const float* dv, const float* addend, const float* mask, float mask_scale, const unsigned* rbits_in, float* y,
                              int N, int H, int W, int C, hipStream_t s);
bool launch_wino_out_in(const float* m, const float* bias, float* v, unsigned* rbits_out, int N, int H, int W, int C, hipStream_t s, bool always = false);   // conv L's output transform + conv L+1's input transform in one kernel (Y never written); false = shape not covered
void launch_wino_dfilter(int tile, const float* du, float* dw, int Cin, int Cout, int KS, hipStream_t s);        // du[P][nsub*Cin][Cout] -> dw[KS*KS][Cin][Cout]
// params: int[4] per image = {y offset, x offset, flip, brightness on/off}; vlut: [N][256] new V per old V (may be nullptr)
void launch_augment_u8(const unsigned char* img, const unsigned char* lab, unsigned char* oimg, unsigned char* olab, const int* params,
                       const unsigned char* vlut, int N, int H, int W, int Ho, int Wo, int void_id, hipStream_t s);
// params: int[4] per image = {resized height, resized width, y offset, x offset}
void launch_resample_u8(const unsigned char* img, const unsigned char* lab, unsigned char* oimg, unsigned char* olab, const int* params,
                        int N, int H, int W, int Ho, int Wo, int void_id, hipStream_t s);
// wrapping sum over every 61st element's bit pattern (weighted by position): changes whenever an optimizer step or a bulk copy touches the buffer
void launch_fingerprint(const float* x, long long n, unsigned long long* out, hipStream_t s);
void launch_init_normal(float* w, long long n, float stddev, int truncated, unsigned long long seed,
                        unsigned int stream_id, hipStream_t s);

// counter-based RNG shared by the fused dropout epilogue and the mask dump
__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__host__ __device__ inline uint32_t philox_u32(unsigned long long idx, unsigned long long seed, uint32_t stream)
{
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream, c3 = 0x9E3779B9u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__host__ __device__ inline float philox_uniform(unsigned long long idx, unsigned long long seed, uint32_t stream)
{
    return (float)(philox_u32(idx, seed, stream) >> 8) * (1.0f / 16777216.0f);   // [0,1)
}

}  // namespace fcn8s
