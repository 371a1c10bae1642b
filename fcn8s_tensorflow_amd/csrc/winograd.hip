// Winograd F(m x m, 3x3) transforms (Lavin & Gray 2015, correlation form), m = 2 or 4, around alpha^2 batched
// GEMMs (alpha = m + 2) that run on igemm_fwd_kernel / wgrad_kernel (MFMA).
//
//   forward / data gradient : V = B^T d B,  U = G g G^T,  M[xi] = V[xi] U[xi] (GEMM over channels),  Y = A^T M A
//   weight gradient         : dM = A dY A^T,  dU[xi] = V[xi]^T dM[xi] (GEMM over tiles),  dg = G^T dU G
//
// Multiplies per output pixel and (ci, co) pair: direct 9, F(2x2) 4, F(4x4) 2.25.  Bytes per transformed pixel:
// F(2x2) 4x the tensor, F(4x4) 2.25x.  fp32 error against an fp64 direct conv (measured, ReLU data, K = 2304):
// direct 2e-7, F(2x2) 5e-7, F(4x4) 7e-6 of the output range -- far inside the 1e-3 logit tolerance.
// The transforms are HBM-bound element-wise kernels (16-byte accesses along the channel axis).
//
// Large kernels (fc6, 7x7): the filter is zero-extended and cut into a grid of r x r sub-filters; sub-filter (a, b) is an
// r x r conv of the input shifted by (r a, r b).  All of them share the OUTPUT tiling, so their products add up in the
// Winograd domain: per position xi one GEMM with K = nsub^2 * Cin ([V_00 | V_01 | ...] x [U_00; U_01; ...]).
//   r = 3: 9x9 extension, 3x3 grid, F(4x4,3x3): 36 positions, 20.25 multiplies per output instead of 49;
//   r = 4: 8x8 extension, 2x2 grid, F(4x4,4x4): 49 positions, 12.25 multiplies per output (default for fc6).
// F(4x4,4x4) uses the points {0, 1, -1, 1/2, -1/2, -2, inf} (Cook-Toom; matrices derived and checked symbolically): fp32
// error 1e-5 of the output range at K = 2048 against 5e-6 for F(4x4,3x3).
#include "fcn8s_internal.h"
#include <cstdlib>

namespace fcn8s {

static inline int wcap(long long work)
{
    long long b = (work + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

template <int M, int R = 3> struct WinoMat;
template <> struct WinoMat<2, 3> {
    static constexpr int A = 4;
    static __device__ __forceinline__ float bt(int i, int j) { constexpr float m[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}}; return m[i][j]; }
    static __device__ __forceinline__ float g(int i, int j) { constexpr float m[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}}; return m[i][j]; }
    static __device__ __forceinline__ float at(int i, int j) { constexpr float m[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}}; return m[i][j]; }
};
template <> struct WinoMat<4, 3> {
    static constexpr int A = 6;
    static __device__ __forceinline__ float bt(int i, int j)
    {
        constexpr float m[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                   {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
        return m[i][j];
    }
    static __device__ __forceinline__ float g(int i, int j)
    {
        constexpr float m[6][3] = {{1.f / 4, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                   {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
        return m[i][j];
    }
    static __device__ __forceinline__ float at(int i, int j)
    {
        constexpr float m[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
        return m[i][j];
    }
};

template <> struct WinoMat<6, 3> {                  // F(6,3), points 0, 1, -1, 2, -2, 1/2, -1/2, inf
    static constexpr int A = 8;
    static __device__ __forceinline__ float bt(int i, int j)
    {
        constexpr float m[8][8] = {{4, 0, -21, 0, 21, 0, -4, 0}, {0, -4, -4, 17, 17, -4, -4, 0}, {0, 4, -4, -17, 17, 4, -4, 0}, {0, 2, 1, -10, -5, 8, 4, 0},
                                   {0, -2, 1, 10, -5, -8, 4, 0}, {0, 4, 8, -5, -10, 1, 2, 0}, {0, -4, 8, 5, -10, -1, 2, 0}, {0, -4, 0, 21, 0, -21, 0, 4}};
        return m[i][j];
    }
    static __device__ __forceinline__ float g(int i, int j)
    {
        constexpr float m[8][3] = {{1.f / 4, 0, 0}, {1.f / 18, 1.f / 18, 1.f / 18}, {1.f / 18, -1.f / 18, 1.f / 18}, {1.f / 360, 1.f / 180, 1.f / 90},
                                   {1.f / 360, -1.f / 180, 1.f / 90}, {16.f / 45, 8.f / 45, 4.f / 45}, {16.f / 45, -8.f / 45, 4.f / 45}, {0, 0, 1.f / 4}};
        return m[i][j];
    }
    static __device__ __forceinline__ float at(int i, int j)
    {
        constexpr float m[6][8] = {{1, 1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 1.f / 2, -1.f / 2, 0}, {0, 1, 1, 4, 4, 1.f / 4, 1.f / 4, 0},
                                   {0, 1, -1, 8, -8, 1.f / 8, -1.f / 8, 0}, {0, 1, 1, 16, 16, 1.f / 16, 1.f / 16, 0}, {0, 1, -1, 32, -32, 1.f / 32, -1.f / 32, 1}};
        return m[i][j];
    }
};
template <> struct WinoMat<4, 4> {                  // F(4,4), points 0, 1, -1, 1/2, -1/2, -2, inf
    static constexpr int A = 7;
    static __device__ __forceinline__ float bt(int i, int j)
    {
        constexpr float m[7][7] = {{2, 1, -10, -5, 8, 4, 0}, {0, -2, -3, 7, 12, 4, 0}, {0, 2, -1, -9, 4, 4, 0}, {0, 2, 5, 0, -5, -2, 0},
                                   {0, -2, 3, 4, -3, -2, 0}, {0, -1, 0, 5, 0, -4, 0}, {0, 2, 1, -10, -5, 8, 4}};
        return m[i][j];
    }
    static __device__ __forceinline__ float g(int i, int j)
    {
        constexpr float m[7][4] = {{1.f / 2, 0, 0, 0}, {1.f / 18, 1.f / 18, 1.f / 18, 1.f / 18}, {1.f / 6, -1.f / 6, 1.f / 6, -1.f / 6},
                                   {8.f / 15, 4.f / 15, 2.f / 15, 1.f / 15}, {8.f / 9, -4.f / 9, 2.f / 9, -1.f / 9},
                                   {1.f / 90, -1.f / 45, 2.f / 45, -4.f / 45}, {0, 0, 0, 1.f / 4}};
        return m[i][j];
    }
    static __device__ __forceinline__ float at(int i, int j)
    {
        constexpr float m[4][7] = {{1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 1.f / 2, -1.f / 2, -2, 0}, {0, 1, 1, 1.f / 4, 1.f / 4, 4, 0},
                                   {0, 1, -1, 1.f / 8, -1.f / 8, -8, 1}};
        return m[i][j];
    }
};

// VEC floats per lane along the channel axis.  F(2x2): 16 bytes (four floats).  F(4x4): 8 bytes (two floats) -- the 6x6 tile
// of four floats per lane needs 170-200 VGPRs (2 waves/SIMD); with float2 the same kernels fit 4+ waves/SIMD, which these
// HBM-bound kernels need to keep enough loads in flight.
template <int VEC> struct alignas(VEC * 4) VecF {
    float d[VEC];
};
template <int VEC> static __device__ __forceinline__ VecF<VEC> vzero() { VecF<VEC> r; _Pragma("unroll") for (int i = 0; i < VEC; ++i) r.d[i] = 0.f; return r; }
template <int VEC> static __device__ __forceinline__ VecF<VEC> vfma(float s, const VecF<VEC>& a, VecF<VEC> acc)
{
    _Pragma("unroll") for (int i = 0; i < VEC; ++i) acc.d[i] = fmaf(s, a.d[i], acc.d[i]);
    return acc;
}
template <int VEC> static __device__ __forceinline__ VecF<VEC> vload_nt(const VecF<VEC>* p)
{
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const vt v = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
    VecF<VEC> r; _Pragma("unroll") for (int i = 0; i < VEC; ++i) r.d[i] = v[i];
    return r;
}
template <int VEC> static __device__ __forceinline__ void vstore_nt(VecF<VEC>* p, const VecF<VEC>& v)
{
    typedef float vt __attribute__((ext_vector_type(VEC)));
    vt t; _Pragma("unroll") for (int i = 0; i < VEC; ++i) t[i] = v.d[i];
    __builtin_nontemporal_store(t, reinterpret_cast<vt*>(p));
}
// 16-byte slab stores for the 8-byte-lane transforms (tools/labs/xform_lab.hip had measured 2-12 % on the stand-alone input transform):
// lanes 2j and 2j + 1 hold the channel pairs c and c + 1 of the same tile; for two neighbouring Winograd positions b and b + 1 they swap halves, so that
// the even lane writes 16 bytes (four channels) of position b and the odd lane 16 bytes of position b + 1 -- half as many store instructions.
// `p` = this lane's own 8-byte address of position b (slab stride `slab` in 8-byte units to position b + 1); needs C / 2 even (every width here is).
// BUILT INTO wino_input_kernel, wino_input_conv1_kernel, wino_dout_kernel and wino_dgrad_output_dout_kernel in round 5 and measured in the training step
// (profiles/r05_st16_ab.txt, three alternating rounds of library builds on one box): transforms 12.97 -> 13.08 ms per step, the step 59.01 -> 59.19 ms --
// SLOWER.  Keeping two positions' values live until the exchange costs registers (wino_dout_kernel 90 -> 190 VGPRs, wino_input_kernel 148 -> 217: two
// waves per SIMD instead of five / three), and these kernels live on occupancy.  Bit-identical results (the parity suite passes on that build).  Off.
#ifndef FCN8S_ST16
#define FCN8S_ST16 0
#endif
#ifndef FCN8S_ST16_GATHER
#define FCN8S_ST16_GATHER FCN8S_ST16
#endif
template <bool NT> static __device__ __forceinline__ void store_pair16(VecF<2>* p, long long slab, const VecF<2>& s0, const VecF<2>& s1)
{
    const bool odd = threadIdx.x & 1;
    const VecF<2> send = odd ? s0 : s1;
    VecF<2> got; got.d[0] = __shfl_xor(send.d[0], 1); got.d[1] = __shfl_xor(send.d[1], 1);
    VecF<4> o;
    if (!odd) { o.d[0] = s0.d[0]; o.d[1] = s0.d[1]; o.d[2] = got.d[0]; o.d[3] = got.d[1]; }
    else { o.d[0] = got.d[0]; o.d[1] = got.d[1]; o.d[2] = s1.d[0]; o.d[3] = s1.d[1]; }
    VecF<4>* dst = reinterpret_cast<VecF<4>*>(odd ? p - 1 + slab : p);
    if constexpr (NT) vstore_nt<4>(dst, o); else *dst = o;
    __builtin_amdgcn_sched_barrier(0);          // (without it hipcc computes every position first and stores afterwards: 90 -> 192 VGPRs in wino_dout_kernel)
}
#define f4fma vfma<VEC>
#define f4zero vzero<VEC>
#define VF VecF<VEC>

// ---- filters: u[xi][sub*Cin + ci][co] = (G g_sub G^T)[xi];  g_sub = taps (3a..3a+2, 3b..3b+2) of the KS x KS filter -----
// transpose_out (nsub == 1 only): u[xi][co][ci] instead of u[xi][ci][co] -- the B operand of the data gradient computed as the
// adjoint of the forward algorithm (launch_wino_dgrad_output), dV[xi] = dM[xi] U[xi]^T
template <int M, int R>
__global__ void wino_filter_kernel(const float* w, float* u, int Cin, int Cout, int KS, int nsub, int transpose_out)
{
    constexpr int A = WinoMat<M, R>::A;
    const long long cc = (long long)Cin * Cout, total = cc * nsub * nsub, ucc = cc * nsub * nsub;
    for (long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x; i0 < total; i0 += (long long)gridDim.x * blockDim.x) {
        // transposed output: consecutive threads walk ci for a fixed co (coalesced stores; the loads of w are strided instead and hit L2)
        const long long i = transpose_out ? (i0 % Cin) * Cout + i0 / Cin : i0;
        const int sub = (int)(i / cc); const long long e = i - sub * cc;
        const int sa = sub / nsub, sb = sub - sa * nsub;
        float g[R][R], t[A][R];
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) {
                const int ky = R * sa + a, kx = R * sb + b;
                g[a][b] = (ky < KS && kx < KS) ? w[(long long)(ky * KS + kx) * cc + e] : 0.f;
            }
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < R; ++k) s = fmaf(WinoMat<M, R>::g(a, k), g[k][b], s);
                t[a][b] = s;
            }
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = 0; b < A; ++b) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < R; ++k) s = fmaf(t[a][k], WinoMat<M, R>::g(b, k), s);
                u[(long long)(a * A + b) * ucc + (transpose_out ? i0 : i)] = s;           // row (sub*Cin + ci), column co
            }
    }
}

// Work mapping shared by the tile transforms: blockIdx.y = (image, tile row), blockIdx.x * 256 + threadIdx.x = (tile column,
// channel vector).  All tensor strides are block-uniform, so every load / store address is one per-thread base pointer
// plus a scalar offset (the first version recomputed the full NHWC index per access: ~350 quarter-rate integer
// multiplies per tile, more VALU time than the HBM time of the bytes it moved).
struct TileIdx { int n, ty, tx, c; long long t; bool ok; };
static __device__ __forceinline__ TileIdx tile_index(int th, int tw, int C4, int N)
{
    TileIdx r;
    const unsigned bx = blockIdx.x, by = blockIdx.y;
    const int idx = bx * 256 + threadIdx.x;
    r.tx = idx / C4; r.c = idx - r.tx * C4;
    r.n = by / th; r.ty = by - r.n * th;
    r.ok = r.tx < tw && r.n < N;
    r.t = ((long long)r.n * th + r.ty) * tw + r.tx;
    return r;
}
static inline dim3 tile_grid(int N, int th, int tw, int C4, int z = 1)
{
    return dim3((unsigned)((tw * C4 + 255) / 256), (unsigned)(N * th), (unsigned)z);
}

// ---- input: one thread = one m x m output tile x VEC channels; alpha x alpha patch (zero outside), V = B^T d B --------
// XB (VEC = 2, pad = 1, nsub = 1): also write the tile's own M x M pixels as bf16 (RNE) into the interior of a zero-bordered padded copy
// xb [N][H + 2][W + 2][C] -- the A operand of the bf16 direct convolution of the same layer (gemm_bf16.hip), which then needs no
// conversion pass of its own over the activations (FCN8S_PREC_BF16_FWD* training: this transform runs anyway, for the weight gradient)
template <int M, int VEC, int R, bool XB = false>
__global__ __launch_bounds__(256) void wino_input_kernel(const VF* __restrict__ x, VF* __restrict__ v, int N, int H, int W, int C4, int pad, int nsub,
                                                         long long slab, unsigned* __restrict__ rbits_out, unsigned* __restrict__ xb = nullptr)
{
    constexpr int A = WinoMat<M, R>::A;
    static_assert(!XB || VEC == 2, "the bf16 copy is written as packed pairs");
    // rbits_out (3x3 layers, pad = 1): also record (x > 0) of the tile's own M x M pixels (patch rows / columns 1..M), in the layout
    // of wino_output_kernel's rbits_out -- for an input that a non-Winograd kernel produced (conv1_1), so that the data gradient
    // of this layer reads 1/32 of that tensor instead of all of it
    constexpr int RW = (M * M * VEC + 31) / 32;
    unsigned rb[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) rb[j] = 0u;
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;      // partial tiles at the bottom / right edge (F(6x6): 512 = 85 * 6 + 2)
    const int sub = blockIdx.z, sa = sub / nsub, sb = sub - sa * nsub;     // sub-filter: patch shifted by (R sa, R sb)
    const int ldv = C4 * nsub * nsub;
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    const int y0 = M * ti.ty + R * sa - pad, x0 = M * ti.tx + R * sb - pad;
    const VF* xp = x + (((long long)ti.n * H + y0) * W + x0) * C4 + ti.c;     // dereferenced only where (row, col) lies inside the image
    bool rok[A], cok[A];
#pragma unroll
    for (int a = 0; a < A; ++a) { rok[a] = (unsigned)(y0 + a) < (unsigned)H; cok[a] = (unsigned)(x0 + a) < (unsigned)W; }
    VF q[A][A];                            // q = B^T d, built column by column so that d is never fully live
#pragma unroll
    for (int b = 0; b < A; ++b) {
        VF d[A];
#pragma unroll
        for (int a = 0; a < A; ++a) d[a] = (rok[a] && cok[b]) ? xp[(a * W + b) * C4] : f4zero();
        if (rbits_out && b >= 1 && b <= M) {
#pragma unroll
            for (int a = 1; a <= M; ++a)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int bit = ((a - 1) * M + (b - 1)) * VEC + i;
                    if (d[a].d[i] > 0.f) rb[bit >> 5] |= 1u << (bit & 31);
                }
        }
        if constexpr (XB) {
            if (b >= 1 && b <= M && cok[b]) {
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                unsigned* xq = xb + (((long long)ti.n * (H + 2) + (y0 + 1)) * (W + 2) + (x0 + b + 1)) * C4 + ti.c;
#pragma unroll
                for (int a = 1; a <= M; ++a)
                    if (rok[a]) {
                        const f32x2_t f = {d[a].d[0], d[a].d[1]};
                        xq[(long long)a * (W + 2) * C4] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
                    }
            }
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, R>::bt(a, k) != 0.f) s = f4fma(WinoMat<M, R>::bt(a, k), d[k], s);
            q[a][b] = s;
        }
    }
    if (rbits_out) {
#pragma unroll
        for (int j = 0; j < RW; ++j) rbits_out[(ti.t * C4 + ti.c) * RW + j] = rb[j];
    }
    VF* vp = v + ti.t * ldv + sub * C4 + ti.c;
    if constexpr (FCN8S_ST16 && VEC == 2 && A % 2 == 0) {
        if ((C4 & 1) == 0 && ((ldv * 2) & 3) == 0) {          // pairs of lanes = pairs of channel groups of one tile, 16-byte aligned rows
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = 0; b < A; b += 2) {
                    VF s0 = f4zero(), s1 = f4zero();
#pragma unroll
                    for (int k = 0; k < A; ++k) {
                        if (WinoMat<M, R>::bt(b, k) != 0.f) s0 = f4fma(WinoMat<M, R>::bt(b, k), q[a][k], s0);
                        if (WinoMat<M, R>::bt(b + 1, k) != 0.f) s1 = f4fma(WinoMat<M, R>::bt(b + 1, k), q[a][k], s1);
                    }
                    store_pair16<XB>(vp + (a * A + b) * slab, slab, s0, s1);
                }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {          // V = q B
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, R>::bt(b, k) != 0.f) s = f4fma(WinoMat<M, R>::bt(b, k), q[a][k], s);
            if constexpr (XB) vstore_nt<VEC>(vp + (a * A + b) * slab, s);      // (V is for the backward pass: it should not push the bf16 copy the convolution reads next out of the caches; -0.06 ms)
            else vp[(a * A + b) * slab] = s;
        }
}

// ---- conv1_1 inside conv1_2's input transform (option "conv1_in_transform") ---------------------------------------------------------
// conv1_1 (3x3, 3 -> 64 channels, bias, ReLU) costs 1.8 GFLOP per image but writes 134 MB that conv1_2's input transform reads right back
// (with the 1.78x patch over-fetch of every input transform).  Here the transform computes the conv1_1 values of its 8 x 8 patch itself, from
// the preprocessed 4-channel image: the block's 8 tiles share a 10 x 52 pixel window staged in LDS (8 KB), the 27 x 64 weights sit
// beside it (7 KB), and a thread builds its two channels' patch column by column -- per column three times nine weight pairs, 30 broadcast reads of
// a pixel (b, g, r, 0) and 432 multiply-adds -- then proceeds exactly like wino_input_kernel<6, 2, 3>: ReLU record of the tile's own 6 x 6 pixels, q = B^T d, V = q B.  conv1_1's
// activation tensor is never written (nothing else reads it: its ReLU mask is the record, its weight gradient needs the image and dY only).
// The multiply-adds run in a fixed order on the VALU, so the values differ from the MFMA kernel's conv1_1 by summation order (1e-7 relative).
// (RB: write the ReLU record -- a template parameter, not a test of the pointer: with the run-time branch hipcc spilled 770 registers)
template <bool RB>
__global__ __launch_bounds__(256) void wino_input_conv1_kernel(const float4* __restrict__ x0, const float* __restrict__ w4, const float* __restrict__ bias,
                                                               VecF<2>* __restrict__ v, int N, int H, int W, long long slab, unsigned* __restrict__ rbits_out)
{
    constexpr int M = 6, A = 8, R = 3, VEC = 2, C4 = 32, RW = (M * M * VEC + 31) / 32, LW = 8 * M + 4;
    __shared__ float4 img[A + 2][LW];
    __shared__ VecF<2> wl[3][9][32];              // [kx][ky * 3 + ci][channel pair]
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;
    const int tid = threadIdx.x, tl = tid >> 5, c = tid & 31;
    const int n = blockIdx.y / th, ty = blockIdx.y - n * th, tx0 = blockIdx.x * 8, tx = tx0 + tl;
    {
        const int gy0 = M * ty - 2, gx0 = M * tx0 - 2;
        for (int i = tid; i < (A + 2) * LW; i += 256) {
            const int r = i / LW, cc = i - r * LW, gy = gy0 + r, gx = gx0 + cc;
            img[r][cc] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? x0[((long long)n * H + gy) * W + gx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int i = tid; i < 27 * 32; i += 256) {      // (27 x 2 weights per thread in registers on top of the 128 of the transform: 400 spilled)
        const int cp = i & 31, e = i >> 5, kx = e / 9, kc = e - kx * 9, ky = kc / 3, ci = kc - ky * 3;
        wl[kx][kc][cp] = *reinterpret_cast<const VecF<2>*>(w4 + ((ky * 3 + kx) * 4 + ci) * 64 + 2 * cp);
    }
    __syncthreads();
    if (tx >= tw) return;
    const VecF<2> bv = *reinterpret_cast<const VecF<2>*>(bias + 2 * c);
    const int y0 = M * ty - 1, xx0 = M * tx - 1;
    bool rok[A], cok[A];
#pragma unroll
    for (int a = 0; a < A; ++a) { rok[a] = (unsigned)(y0 + a) < (unsigned)H; cok[a] = (unsigned)(xx0 + a) < (unsigned)W; }
    unsigned rb[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) rb[j] = 0u;
    const long long t = ((long long)n * th + ty) * tw + tx;
    VecF<2> q[A][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {
        VecF<2> acc[A];
#pragma unroll
        for (int a = 0; a < A; ++a) acc[a] = bv;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // (the window column is made opaque per (b, kx): hipcc otherwise keeps the pixels two neighbouring patch columns share in registers --
            //  and hoists the reads of whole columns -- for 431 spilled registers; the scheduling barrier bounds what is in flight to one window column)
            int col = M * tl + b + kx, cw = c;
            asm volatile("" : "+v"(col), "+v"(cw));
            VecF<2> wt[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) wt[e] = wl[kx][e][cw];
#pragma unroll
            for (int r = 0; r < A + 2; ++r) {
                const float4 p = img[r][col];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int a = r - ky;
                    if (a >= 0 && a < A) {
                        acc[a] = vfma<2>(p.x, wt[ky * 3 + 0], acc[a]);
                        acc[a] = vfma<2>(p.y, wt[ky * 3 + 1], acc[a]);
                        acc[a] = vfma<2>(p.z, wt[ky * 3 + 2], acc[a]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        VecF<2> d[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const bool in = rok[a] && cok[b];
#pragma unroll
            for (int i = 0; i < VEC; ++i) d[a].d[i] = (in && acc[a].d[i] > 0.f) ? acc[a].d[i] : 0.f;
        }
        if (RB && b >= 1 && b <= M) {
#pragma unroll
            for (int a = 1; a <= M; ++a)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const int bit = ((a - 1) * M + (b - 1)) * VEC + i;
                    if (d[a].d[i] > 0.f) rb[bit >> 5] |= 1u << (bit & 31);
                }
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            VecF<2> s = vzero<2>();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, R>::bt(a, k) != 0.f) s = vfma<2>(WinoMat<M, R>::bt(a, k), d[k], s);
            q[a][b] = s;
        }
    }
    if (RB) {
#pragma unroll
        for (int j = 0; j < RW; ++j) rbits_out[(t * C4 + c) * RW + j] = rb[j];
    }
    VecF<2>* vp = v + t * C4 + c;
#if FCN8S_ST16
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; b += 2) {
            VecF<2> s0 = vzero<2>(), s1 = vzero<2>();
#pragma unroll
            for (int k = 0; k < A; ++k) {
                if (WinoMat<M, R>::bt(b, k) != 0.f) s0 = vfma<2>(WinoMat<M, R>::bt(b, k), q[a][k], s0);
                if (WinoMat<M, R>::bt(b + 1, k) != 0.f) s1 = vfma<2>(WinoMat<M, R>::bt(b + 1, k), q[a][k], s1);
            }
            store_pair16<true>(vp + (a * A + b) * slab, slab, s0, s1);      // (C4 = 32 channel pairs: lanes 2j, 2j + 1 are neighbours of one tile)
        }
#else
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {
            VecF<2> s = vzero<2>();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, R>::bt(b, k) != 0.f) s = vfma<2>(WinoMat<M, R>::bt(b, k), q[a][k], s);
            vstore_nt<2>(vp + (a * A + b) * slab, s);      // (non-temporal: 59.20 / 59.09 / 58.93 -> 58.94 / 58.82 / 58.75 ms per step in alternating runs on one box)
        }
#endif
}

// ---- F(4x4,3x3) backward pair: the data-gradient conv needs V = B^T dy B, the weight gradient dM = A dy A^T of the same
// tensor (dM's 4x4 tile is the inside of V's 6x6 patch): both from one read of dy -------------------------------------------
// POOL: dy is not materialised -- it is the max-pool backward of dpool [N,H/2,W/2,C] routed by the argmax bytes the forward
// output transform kept (dy[pixel] = dpool[window] if the pixel is the window's first maximum and that maximum is > 0).
template <int M, int VEC, bool POOL>
__global__ __launch_bounds__(256) void wino_input_dout_kernel(const VF* __restrict__ x, VF* __restrict__ v, VF* __restrict__ dm,
                                                              int N, int H, int W, int C4, long long slab, const unsigned char* __restrict__ pidx)
{
    constexpr int A = M + 2, NW = M / 2 + 2;           // NW: pool windows one patch row / column touches
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;      // partial tiles at the bottom / right edge (F(6x6): 512 = 85 * 6 + 2)
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    const int y0 = M * ti.ty - 1, x0 = M * ti.tx - 1;
    const VF* xp = x + (((long long)ti.n * H + y0) * W + x0) * C4 + ti.c;
    bool rok[A], cok[A];
#pragma unroll
    for (int a = 0; a < A; ++a) { rok[a] = (unsigned)(y0 + a) < (unsigned)H; cok[a] = (unsigned)(x0 + a) < (unsigned)W; }
    VF q[A][A], p[A][M];
    // POOL: the patch rows y0..y0+M+1 = M ty - 1 .. M ty + M touch window rows (M/2) ty - 1 .. (M/2) ty + M/2 (NW of them); same for columns
    const int Hp = H / 2, Wp = W / 2;
    const long long wbase = (((long long)ti.n * Hp + ((M / 2) * ti.ty - 1)) * Wp + ((M / 2) * ti.tx - 1)) * C4 + ti.c;
#pragma unroll
    for (int b = 0; b < A; ++b) {
        VF d[A];
        if (POOL) {
            const int wc = (b + 1) >> 1;                                   // window column 0..3 of patch column b; (x0 + b) & 1 == (b + 1) & 1
#pragma unroll
            for (int wr = 0; wr < NW; ++wr) {                              // window rows; patch rows 2wr-1, 2wr
                const bool wok = (unsigned)((M / 2) * ti.ty - 1 + wr) < (unsigned)Hp && (unsigned)((M / 2) * ti.tx - 1 + wc) < (unsigned)Wp;
                const long long wo = wbase + ((long long)wr * Wp + wc) * C4;
                VF g = f4zero();
                unsigned char id[VEC];
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) id[i] = 4;
                if (wok) { g = x[wo]; _Pragma("unroll") for (int i = 0; i < VEC; ++i) id[i] = pidx[wo * VEC + i]; }
                _Pragma("unroll") for (int half = 0; half < 2; ++half) {
                    const int a = 2 * wr - 1 + half;                       // patch row; its parity inside the window is (a + 1) & 1 == half ^ 1 ... see pos
                    if (a < 0 || a >= A) continue;
                    const int pos = ((a + 1) & 1) * 2 + ((b + 1) & 1);     // (row parity, column parity) of the pixel inside its window
                    _Pragma("unroll") for (int i = 0; i < VEC; ++i) d[a].d[i] = id[i] == pos ? g.d[i] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < A; ++a) d[a] = (rok[a] && cok[b]) ? xp[(a * W + b) * C4] : f4zero();
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, 3>::bt(a, k) != 0.f) s = f4fma(WinoMat<M, 3>::bt(a, k), d[k], s);
            q[a][b] = s;
        }
        if (b >= 1 && b <= M) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                VF s = f4zero();
#pragma unroll
                for (int k = 0; k < M; ++k) if (WinoMat<M, 3>::at(k, a) != 0.f) s = f4fma(WinoMat<M, 3>::at(k, a), d[k + 1], s);
                p[a][b - 1] = s;
            }
        }
    }
    const long long o = ti.t * C4 + ti.c;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, 3>::bt(b, k) != 0.f) s = f4fma(WinoMat<M, 3>::bt(b, k), q[a][k], s);
            v[o + (a * A + b) * slab] = s;
        }
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < M; ++k) if (WinoMat<M, 3>::at(k, b) != 0.f) s = f4fma(WinoMat<M, 3>::at(k, b), p[a][k], s);
            dm[o + (a * A + b) * slab] = s;
        }
}

// ---- output: one thread = one tile x VEC channels; Y = A^T M A, then the conv epilogue --------------------------------
// DROPOUT is a template parameter: the inlined Philox rounds (fc6 only) otherwise cost every launch their registers.
// NT: the activation / pool stores are non-temporal -- chosen by the launcher for outputs too large to be of use to the next kernel from the caches
// (training batches: transforms -0.2 ms per step; a single image's maps are better left temporal: 1.822 against 1.831 ms per prediction)
template <int M, int VEC, bool DROPOUT, int R, bool POOL, bool NT = false>
__global__ __launch_bounds__(256, 3) void wino_output_kernel(const VF* __restrict__ m, const VF* __restrict__ bias, const VF* __restrict__ addend,
                                                          const VF* __restrict__ mask, float mask_scale, int relu, VF* __restrict__ y,
                                                          int N, int H, int W, int C4, float keep, unsigned long long seed, unsigned int stream_id,
                                                          long long slab, VF* __restrict__ pool, unsigned char* __restrict__ pidx,
                                                          unsigned* __restrict__ rbits_out, const unsigned* __restrict__ rbits_in)
{
    constexpr int A = WinoMat<M, R>::A;
    // ReLU bit masks: a forward launch can record (y > 0) as one bit per element, RW words per (tile, channel vector); the data-
    // gradient launch that would read y back as its mask (same geometry, same thread mapping) reads those words instead (1/32 of the bytes)
    constexpr int RW = (M * M * VEC + 31) / 32;
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;      // partial tiles at the bottom / right edge (F(6x6): 512 = 85 * 6 + 2)
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    unsigned rb[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) rb[j] = rbits_in ? rbits_in[(ti.t * C4 + ti.c) * RW + j] : 0u;
    const VF* mp = m + ti.t * C4 + ti.c;
    VF q[M][A];                            // q = A^T M, column by column
#pragma unroll
    for (int b = 0; b < A; ++b) {
        VF col[A];
#pragma unroll
        for (int a = 0; a < A; ++a) col[a] = vload_nt<VEC>(mp + (a * A + b) * slab);       // read exactly once, fully coalesced: non-temporal (-0.15 ms/step; on the gathers' dV, the pool gradients, the loss kernel's logits and conv1_1's dZ the same hint measured neutral to slower)
#pragma unroll
        for (int o = 0; o < M; ++o) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, R>::at(o, k) != 0.f) s = f4fma(WinoMat<M, R>::at(o, k), col[k], s);
            q[o][b] = s;
        }
    }
    const VF bv = bias ? bias[ti.c] : f4zero();
    const long long off0 = (((long long)ti.n * H + M * ti.ty) * W + M * ti.tx) * C4 + ti.c;
    constexpr int PW = POOL ? M / 2 : 1;       // POOL: the launch also writes the 2x2/2 max-pool (a template flag keeps the other 80 % of the launches lean)
    VF pmax[PW][PW];                       // the 2x2/2 max-pool of this tile (tile origins are even), written below
    unsigned char parg[PW][PW][VEC];           // pidx != nullptr: which window element is the FIRST maximum (0..3; 4 = max not > 0, i.e. no
                                               // gradient through the ReLU) -- the routing rule of maxpool_bwd_kernel, kept for the backward pass
#pragma unroll
    for (int oy = 0; oy < M; ++oy)
#pragma unroll
        for (int ox = 0; ox < M; ++ox) {
            VF v = bv;
#pragma unroll
            for (int k = 0; k < A; ++k) if (WinoMat<M, R>::at(ox, k) != 0.f) v = f4fma(WinoMat<M, R>::at(ox, k), q[oy][k], v);
            const long long off = off0 + (oy * W + ox) * C4;
            const bool inside = M * ti.ty + oy < H && M * ti.tx + ox < W;              // false only in partial edge tiles
            if (!inside) { if (POOL && (oy & 1) == 0 && (ox & 1) == 0) { pmax[POOL ? oy / 2 : 0][POOL ? ox / 2 : 0] = v; _Pragma("unroll") for (int i = 0; i < VEC; ++i) parg[POOL ? oy / 2 : 0][POOL ? ox / 2 : 0][i] = 0; } continue; }
            if (addend) { const VF ad = addend[off]; _Pragma("unroll") for (int i = 0; i < VEC; ++i) v.d[i] += ad.d[i]; }
            if (relu) { _Pragma("unroll") for (int i = 0; i < VEC; ++i) v.d[i] = fmaxf(v.d[i], 0.f); }
            if (rbits_in) {
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) {
                    constexpr int dummy = 0; (void)dummy;
                    const int bit = (oy * M + ox) * VEC + i;
                    v.d[i] = ((rb[bit >> 5] >> (bit & 31)) & 1u) ? v.d[i] * mask_scale : 0.f;
                }
            } else if (mask) {
                const VF k = mask[off];
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) v.d[i] = k.d[i] > 0.f ? v.d[i] * mask_scale : 0.f;
            }
            if (DROPOUT) {                     // same Philox stream as the direct kernel's epilogue: element index NHWC
                const unsigned long long e = (unsigned long long)off * VEC;
                const float ik = 1.f / keep;
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) v.d[i] = philox_uniform(e + i, seed, stream_id) < keep ? v.d[i] * ik : 0.f;
            }
            if (!POOL || y) { if constexpr (NT) vstore_nt<VEC>(y + off, v); else y[off] = v; }          // POOL launches may skip the full-resolution tensor (nobody reads it: see model.hip forward())
                     // POOL launches may skip the full-resolution tensor (nobody reads it: see model.hip forward())
            if (rbits_out) {
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) {
                    const int bit = (oy * M + ox) * VEC + i;
                    if (v.d[i] > 0.f) rb[bit >> 5] |= 1u << (bit & 31);
                }
            }
            if (POOL) {
                constexpr int dummy = 0; (void)dummy;
                const int py = POOL ? oy / 2 : 0, px = POOL ? ox / 2 : 0;
                if ((oy & 1) == 0 && (ox & 1) == 0) {
                    pmax[py][px] = v;
                    _Pragma("unroll") for (int i = 0; i < VEC; ++i) parg[py][px][i] = 0;
                } else {
                    _Pragma("unroll") for (int i = 0; i < VEC; ++i)
                        if (v.d[i] > pmax[py][px].d[i]) { pmax[py][px].d[i] = v.d[i]; parg[py][px][i] = (unsigned char)((oy & 1) * 2 + (ox & 1)); }
                }
            }
        }
    if (POOL && pool) {
        const int Hp = H / 2, Wp = W / 2;
#pragma unroll
        for (int py = 0; py < PW; ++py)
#pragma unroll
            for (int px = 0; px < PW; ++px)
            if ((M / 2) * ti.ty + py < Hp && (M / 2) * ti.tx + px < Wp) {                // H, W even: a window is inside or outside as a whole
                const long long po = (((long long)ti.n * Hp + (M / 2) * ti.ty + py) * Wp + (M / 2) * ti.tx + px) * C4 + ti.c;
                if constexpr (NT) vstore_nt<VEC>(pool + po, pmax[py][px]); else pool[po] = pmax[py][px];
                if (pidx) { _Pragma("unroll") for (int i = 0; i < VEC; ++i) pidx[po * VEC + i] = pmax[py][px].d[i] > 0.f ? parg[py][px][i] : (unsigned char)4; }
            }
    }
    if (rbits_out) {
#pragma unroll
        for (int j = 0; j < RW; ++j) rbits_out[(ti.t * C4 + ti.c) * RW + j] = rb[j];
    }
}

// ---- forward: output transform of conv L fused with the input transform of conv L+1 (round 4) ---------------------------------------
// Inside a VGG block conv L's output Y is read by nobody but conv L+1's input transform, on the same tile grid (same H, W, F(6x6,3x3)
// on both sides): Y = relu(A^T M A + bias) per 6x6 tile, then V' = B^T d B over the 8x8 patch = the tile plus a one-pixel ring from its
// eight neighbours.  This kernel goes from M to V' without Y ever reaching memory: a block owns a strip of TX tile columns (plus one
// ring column on either side, computed but not emitted) x CPB channel pairs and WALKS DOWN the tile rows of one image.  In iteration r
// every thread turns the 64 M values of tile (r, column) into its 6x6 Y tile (registers); the tile of iteration r - 1 is still in
// registers (yprev), row 5 of the one before too (top); what a thread needs from its left / right neighbours -- their columns 5 / 0 of
// tile row r - 1 and the corner pixels of rows r - 2 and r -- travels through LDS (18 values per thread).  Then tile (r - 1) has its whole
// 8x8 patch and is transformed and stored.  Every M value is read once (the ring columns twice: 2 / TX), the halo rows above and below a
// block's row range are recomputed (2 per range).  Arithmetic: the loops of wino_output_kernel and wino_input_kernel, in their order --
// the V' written here has the bits the two kernels produce.  Pixels outside the image (partial edge tiles, the ring around the image)
// enter the patch as zeros, as SAME padding wants.  rbits_out: the ReLU record of Y the backward pass masks with (layout of wino_output_kernel).
#ifndef OI_NT_STORE
#define OI_NT_STORE 1          // non-temporal stores of V': 2.62 instead of 2.74 ms per step over the seven launches (they measured worse in the single transforms)
#endif
#ifndef OI_WPS
#define OI_WPS 2               // waves per SIMD the register allocation aims at (1 = up to 512 registers, no spills; measured below)
#endif
template <int CPB, int NCOL>
__global__ __launch_bounds__(CPB * NCOL, OI_WPS == 1 ? 1 : 512 / (CPB * NCOL)) void wino_out_in_kernel(const VecF<2>* __restrict__ m, const VecF<2>* __restrict__ bias, VecF<2>* __restrict__ v,
                                                                    unsigned* __restrict__ rbits_out, int N, int H, int W, int C4, long long slab, int rows_per_block)
{
    constexpr int VEC = 2, M = 6, A = 8, TX = NCOL - 2, RW = (M * M * VEC + 31) / 32, SLOTS = 24;
    typedef WinoMat<6, 3> WM;
    // one array [tile column][24 slots][channel pair] (slots 16-21: row 5 of the thread's own tile of two iterations ago, parked here rather than in registers): slots 0-5 = column 0 of the column's tile (rows 0..5), 6-11 = its column 5, 12 / 13 = row 5's
    // pixels 0 / 5 of the tile above it, 14 / 15 = row 0's pixels 0 / 5 of the tile below.  A thread addresses it through three bases (its own
    // column and the two neighbours) plus compile-time slot offsets, which the LDS instructions carry as immediates.
    __shared__ VF lds[NCOL * SLOTS * CPB];
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;
    const int j = threadIdx.x / CPB, cpl = threadIdx.x - j * CPB;
    const int ngroups = C4 / CPB;
    const int strip = blockIdx.x / ngroups, cg = blockIdx.x - strip * ngroups;
    const int c = cg * CPB + cpl;
    const int tx = strip * TX - 1 + j;
    const int chunks = (th + rows_per_block - 1) / rows_per_block;
    const int n = blockIdx.y / chunks, r0 = (blockIdx.y - n * chunks) * rows_per_block;
    const int r1 = r0 + rows_per_block < th ? r0 + rows_per_block : th;
    const bool colok = tx >= 0 && tx < tw;
    const bool inner = j >= 1 && j <= TX && colok;
    const VF bv = bias[c];
    VF* const lmine = lds + (j * SLOTS) * CPB + cpl;
    const VF* const lleft = lmine - SLOTS * CPB;            // (only dereferenced by inner threads: 1 <= j <= TX)
    const VF* const lright = lmine + SLOTS * CPB;
    VF yprev[M][M], ynew[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a) { lmine[(16 + a) * CPB] = f4zero(); _Pragma("unroll") for (int b = 0; b < M; ++b) yprev[a][b] = f4zero(); }

    for (int r = r0 - 1; r <= r1; ++r) {
        // (the 2 x 64 slab offsets are block-uniform products: hoisted out of this loop they would sit in 256 registers for its whole
        //  duration -- the first build spilled exactly those; opaque per iteration, they are recomputed on the scalar unit when needed)
        long long slab_a = slab, slab_b = slab;
        asm volatile("" : "+s"(slab_a), "+s"(slab_b));
        // ---- A: Y of tile (r, tx) ------------------------------------------------------------------------------------------------
        const bool valid = colok && r >= 0 && r < th;
        const long long t = ((long long)n * th + r) * tw + tx;
        if (valid) {
            const VF* mp = m + t * C4 + c;
            // Y = A^T M A accumulated column by column of M: q_b = A^T M[:, b] (six values), then the rank-one update Y[oy][ox] += q_b[oy] A[b][ox].
            // Each Y element sees its terms in the order b = 0 .. 7 on top of the bias -- the fma chain of wino_output_kernel, same bits -- and the
            // 6 x 8 intermediate of that kernel never exists (the previous tile and this one already fill 156 of the 256 registers).
#pragma unroll
            for (int oy = 0; oy < M; ++oy) _Pragma("unroll") for (int ox = 0; ox < M; ++ox) ynew[oy][ox] = bv;
#pragma unroll
            for (int b = 0; b < A; ++b) {
                VF col[A];
#pragma unroll
                for (int a = 0; a < A; ++a) col[a] = vload_nt<VEC>(mp + (a * A + b) * slab_a);       // (plain loads: 2.96 instead of 2.74 ms per step over the seven launches)
#pragma unroll
                for (int o = 0; o < M; ++o) {
                    VF sacc = f4zero();
#pragma unroll
                    for (int k = 0; k < A; ++k) if (WM::at(o, k) != 0.f) sacc = f4fma(WM::at(o, k), col[k], sacc);
#pragma unroll
                    for (int ox = 0; ox < M; ++ox) if (WM::at(ox, b) != 0.f) ynew[o][ox] = f4fma(WM::at(ox, b), sacc, ynew[o][ox]);
                }
                if (b == 3) __builtin_amdgcn_sched_barrier(0);      // the tile's 64 loads in two batches of 32: all at once, on top of two tiles of state, spilled 36 registers (6 now)
            }
            unsigned rb[RW];
#pragma unroll
            for (int w_ = 0; w_ < RW; ++w_) rb[w_] = 0u;
#pragma unroll
            for (int oy = 0; oy < M; ++oy)
#pragma unroll
                for (int ox = 0; ox < M; ++ox) {
                    const bool inside = M * r + oy < H && M * tx + ox < W;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float val = inside ? fmaxf(ynew[oy][ox].d[i], 0.f) : 0.f;
                        ynew[oy][ox].d[i] = val;
                        const int bit = (oy * M + ox) * VEC + i;
                        if (val > 0.f) rb[bit >> 5] |= 1u << (bit & 31);
                    }
                }
            if (rbits_out && inner && r >= r0 && r < r1) {
#pragma unroll
                for (int w_ = 0; w_ < RW; ++w_) rbits_out[(t * C4 + c) * RW + w_] = rb[w_];
            }
        } else {
#pragma unroll
            for (int a = 0; a < M; ++a) _Pragma("unroll") for (int b = 0; b < M; ++b) ynew[a][b] = f4zero();
        }
        lmine[14 * CPB] = ynew[0][0]; lmine[15 * CPB] = ynew[0][M - 1];
        __syncthreads();
        // ---- B: the patch of tile (r - 1, tx) is complete: V' = B^T d B -----------------------------------------------------------
        if (inner && r - 1 >= r0 && r - 1 < r1) {
            // (row by row of B^T d: the patch IS yprev / top / ynew plus the two LDS columns, so only one row of the intermediate is live)
            VF left[A], right[A], top[M];
#pragma unroll
            for (int a = 0; a < M; ++a) top[a] = lmine[(16 + a) * CPB];
            left[0] = lleft[13 * CPB]; right[0] = lright[12 * CPB];
#pragma unroll
            for (int a = 0; a < M; ++a) { left[a + 1] = lleft[(6 + a) * CPB]; right[a + 1] = lright[a * CPB]; }
            left[A - 1] = lleft[15 * CPB]; right[A - 1] = lright[14 * CPB];
            VF* vp = v + (t - tw) * C4 + c;                       // tile (r - 1, tx)
#pragma unroll
            for (int a = 0; a < A; ++a) {
                VF qa[A];
#pragma unroll
                for (int b = 0; b < A; ++b) {
                    VF sacc = f4zero();
#pragma unroll
                    for (int k = 0; k < A; ++k) if (WM::bt(a, k) != 0.f) {
                        const VF& pe = b == 0 ? left[k] : b == A - 1 ? right[k] : k == 0 ? top[b - 1] : k == A - 1 ? ynew[0][b - 1] : yprev[k - 1][b - 1];
                        sacc = f4fma(WM::bt(a, k), pe, sacc);
                    }
                    qa[b] = sacc;
                }
#pragma unroll
                for (int b = 0; b < A; ++b) {
                    VF sacc = f4zero();
#pragma unroll
                    for (int k = 0; k < A; ++k) if (WM::bt(b, k) != 0.f) sacc = f4fma(WM::bt(b, k), qa[k], sacc);
                    if (OI_NT_STORE) { typedef float vt2 __attribute__((ext_vector_type(2))); vt2 o2 = {sacc.d[0], sacc.d[1]}; __builtin_nontemporal_store(o2, reinterpret_cast<vt2*>(vp + (a * A + b) * slab_b)); }
                    else vp[(a * A + b) * slab_b] = sacc;
                }
            }
        }
        __syncthreads();
        // ---- C: tile r becomes the previous one --------------------------------------------------------------------------------------
        lmine[12 * CPB] = yprev[M - 1][0]; lmine[13 * CPB] = yprev[M - 1][M - 1];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            lmine[(16 + a) * CPB] = yprev[M - 1][a];
            lmine[a * CPB] = ynew[a][0]; lmine[(6 + a) * CPB] = ynew[a][M - 1];
        }
#pragma unroll
        for (int a = 0; a < M; ++a) _Pragma("unroll") for (int b = 0; b < M; ++b) yprev[a][b] = ynew[a][b];
    }
}

// ---- weight gradient: dM = A dY A^T (alpha x alpha from the m x m output-gradient tile) ------------------------------
// POOL: dy is not materialised -- it is the max-pool backward of dpool [N,H/2,W/2,C] routed by the argmax bytes the forward output
// transform kept (see wino_input_dout_kernel); tile origins are even, so a tile covers whole windows.
template <int M, int VEC, int R, bool POOL>
__global__ __launch_bounds__(256) void wino_dout_kernel(const VF* __restrict__ dy, VF* __restrict__ dm, int N, int H, int W, int C4, long long slab,
                                                        const unsigned char* __restrict__ pidx)
{
    constexpr int A = WinoMat<M, R>::A;
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;      // partial tiles at the bottom / right edge (F(6x6): 512 = 85 * 6 + 2)
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    const VF* yp = dy + (((long long)ti.n * H + M * ti.ty) * W + M * ti.tx) * C4 + ti.c;
    const int Hp = H / 2, Wp = W / 2;
    // POOL: the (M/2)^2 windows of the tile -- gradient and argmax bytes -- are all requested up front (one load latency per
    // thread instead of one per window column: the kernel spent 40 % of its time on these 15 % of its bytes)
    VF pg[POOL ? M / 2 : 1][POOL ? M / 2 : 1];
    unsigned pid[POOL ? M / 2 : 1][POOL ? M / 2 : 1];
    if (POOL) {
#pragma unroll
        for (int wr = 0; wr < M / 2; ++wr)
#pragma unroll
            for (int wc = 0; wc < M / 2; ++wc) {
                const int py = (M / 2) * ti.ty + wr, px = (M / 2) * ti.tx + wc;
                const bool wok = py < Hp && px < Wp;
                const long long wo = (((long long)ti.n * Hp + py) * Wp + px) * C4 + ti.c;
                pg[wr][wc] = f4zero();
                pid[wr][wc] = 0x04040404u;                     // 4 = no position (window outside the map, or its maximum was <= 0)
                if (wok) {
                    pg[wr][wc] = dy[wo];
                    if (VEC == 4) pid[wr][wc] = *reinterpret_cast<const unsigned*>(pidx + wo * 4);
                    else if (VEC == 2) pid[wr][wc] = *reinterpret_cast<const unsigned short*>(pidx + wo * 2);
                    else { unsigned w = 0; _Pragma("unroll") for (int i = 0; i < VEC; ++i) w |= (unsigned)pidx[wo * VEC + i] << (8 * i); pid[wr][wc] = w; }
                }
            }
    }
    VF q[A][M];                            // q = A dY  (A = (A^T)^T)
#pragma unroll
    for (int ox = 0; ox < M; ++ox) {
        VF col[M];
        if (POOL) {
#pragma unroll
            for (int wr = 0; wr < M / 2; ++wr) {
                const VF g = pg[wr][ox / 2];
                const unsigned w = pid[wr][ox / 2];
                _Pragma("unroll") for (int half = 0; half < 2; ++half) {
                    const unsigned pos = half * 2 + (ox & 1);
                    _Pragma("unroll") for (int i = 0; i < VEC; ++i) col[2 * wr + half].d[i] = ((w >> (8 * i)) & 0xffu) == pos ? g.d[i] : 0.f;
                }
            }
        } else {
#pragma unroll
        for (int oy = 0; oy < M; ++oy) col[oy] = (M * ti.ty + oy < H && M * ti.tx + ox < W) ? yp[(oy * W + ox) * C4] : f4zero();
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < M; ++k) if (WinoMat<M, R>::at(k, a) != 0.f) s = f4fma(WinoMat<M, R>::at(k, a), col[k], s);
            q[a][ox] = s;
        }
    }
    VF* dp = dm + ti.t * C4 + ti.c;
    if constexpr (FCN8S_ST16 && VEC == 2 && A % 2 == 0) {
        if ((C4 & 1) == 0) {
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = 0; b < A; b += 2) {          // dM = q A^T, two positions per 16-byte store
                    VF s0 = f4zero(), s1 = f4zero();
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        if (WinoMat<M, R>::at(k, b) != 0.f) s0 = f4fma(WinoMat<M, R>::at(k, b), q[a][k], s0);
                        if (WinoMat<M, R>::at(k, b + 1) != 0.f) s1 = f4fma(WinoMat<M, R>::at(k, b + 1), q[a][k], s1);
                    }
                    store_pair16<true>(dp + (a * A + b) * slab, slab, s0, s1);
                }
            return;
        }
    }
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < A; ++b) {          // dM = q A^T
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < M; ++k) if (WinoMat<M, R>::at(k, b) != 0.f) s = f4fma(WinoMat<M, R>::at(k, b), q[a][k], s);
            vstore_nt<VEC>(dp + (a * A + b) * slab, s);      // non-temporal: the weight-gradient GEMMs that read dM next run 2 % faster (profiles/r04_nt_stores_ab.txt)
        }
}

// ---- data gradient as the adjoint of the forward algorithm (F(6x6,3x3)): dx = sum over tiles of B dV B^T, overlap-added ----------
// Forward: V = B^T d B per tile (8x8 patches, stride 6).  Its adjoint maps dV = dM U^T (dM = A dY A^T, the tensor the weight gradient
// needs anyway) back to the input: every tile contributes the 8x8 patch B dV B^T, and neighbouring patches overlap by two pixels.
// One thread = one 6x6 output region x VEC channels, computed as a gather: the inner 6x6 of its own patch plus the border rows /
// columns / corners of the eight neighbouring patches.  Row 0 and row 7 of B have a single non-zero (B^T(0,0), B^T(7,7)), so a
// neighbour's border row needs only its eight values of position row 7 (or 0): 100 loads per thread instead of 64, the extra ones
// from slabs other threads of the same wave stream anyway.  Compared with transforming dY a second time (V' = B^T dY_patch B) this
// removes one [P][T][C] write per layer from the backward pass.
template <int VEC>
__global__ __launch_bounds__(256, 2) void wino_dgrad_output_kernel(const VF* __restrict__ dv, const VF* __restrict__ addend,
                                                                   const VF* __restrict__ mask, float mask_scale, const unsigned* __restrict__ rbits_in,
                                                                   VF* __restrict__ y, int N, int H, int W, int C4, long long slab)
{
    constexpr int M = 6, A = 8;
    typedef WinoMat<6, 3> WM;
    constexpr int RW = (M * M * VEC + 31) / 32;
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    const long long o = ti.t * C4 + ti.c;
    unsigned rb[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) rb[j] = rbits_in ? rbits_in[o * RW + j] : 0u;
    VF tt[M][A];                           // tt[i][b] = sum_a B^T(a, i+1) dV[a][b]   (patch rows 1..6)
#pragma unroll
    for (int b = 0; b < A; ++b) {
        VF col[A];
#pragma unroll
        for (int a = 0; a < A; ++a) col[a] = dv[(a * A + b) * slab + o];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            VF s = f4zero();
#pragma unroll
            for (int a = 0; a < A; ++a) if (WM::bt(a, i + 1) != 0.f) s = f4fma(WM::bt(a, i + 1), col[a], s);
            tt[i][b] = s;
        }
    }
    // neighbours' border contributions
    const bool up = ti.ty > 0, down = ti.ty + 1 < th, left = ti.tx > 0, right = ti.tx + 1 < tw;
    VF nrow[2][M], ncol[2][M], corner[2][2];      // [0] = top / left, [1] = bottom / right
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const bool okr = e == 0 ? up : down, okc = e == 0 ? left : right;
        const long long orow = o + (e == 0 ? -(long long)tw : (long long)tw) * C4, ocol = o + (e == 0 ? -1 : 1) * (long long)C4;
        const int pa = e == 0 ? 7 : 0;                 // the neighbour's position row / column that reaches into this region
        const float edge = WM::bt(pa, pa);             // the single non-zero of B's row `pa`
        VF r[A], c[A];
#pragma unroll
        for (int k = 0; k < A; ++k) { r[k] = okr ? dv[(pa * A + k) * slab + orow] : f4zero(); c[k] = okc ? dv[(k * A + pa) * slab + ocol] : f4zero(); }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            VF sr = f4zero(), sc = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WM::bt(k, j + 1) != 0.f) { sr = f4fma(WM::bt(k, j + 1), r[k], sr); sc = f4fma(WM::bt(k, j + 1), c[k], sc); }
            _Pragma("unroll") for (int v = 0; v < VEC; ++v) { sr.d[v] *= edge; sc.d[v] *= edge; }
            nrow[e][j] = sr; ncol[e][j] = sc;
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {                  // corners: (row neighbour e, column neighbour f)
            const bool okf = f == 0 ? left : right;
            const int pb = f == 0 ? 7 : 0;
            const long long oc = orow + (f == 0 ? -1 : 1) * (long long)C4;
            VF v = (okr && okf) ? dv[(pa * A + pb) * slab + oc] : f4zero();
            _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] *= edge * WM::bt(pb, pb);
            corner[e][f] = v;
        }
    }
    const long long off0 = (((long long)ti.n * H + M * ti.ty) * W + M * ti.tx) * C4 + ti.c;
#pragma unroll
    for (int oy = 0; oy < M; ++oy)
#pragma unroll
        for (int ox = 0; ox < M; ++ox) {
            VF v = f4zero();
#pragma unroll
            for (int b = 0; b < A; ++b) if (WM::bt(b, ox + 1) != 0.f) v = f4fma(WM::bt(b, ox + 1), tt[oy][b], v);
            if (oy == 0)     { _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += nrow[0][ox].d[k]; }
            if (oy == M - 1) { _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += nrow[1][ox].d[k]; }
            if (ox == 0)     { _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += ncol[0][oy].d[k]; }
            if (ox == M - 1) { _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += ncol[1][oy].d[k]; }
            if ((oy == 0 || oy == M - 1) && (ox == 0 || ox == M - 1)) { _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += corner[oy == 0 ? 0 : 1][ox == 0 ? 0 : 1].d[k]; }
            if (!(M * ti.ty + oy < H && M * ti.tx + ox < W)) continue;                   // partial edge tiles
            const long long off = off0 + (oy * W + ox) * C4;
            if (addend) { const VF ad = addend[off]; _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += ad.d[k]; }
            if (rbits_in) {
                _Pragma("unroll") for (int k = 0; k < VEC; ++k) {
                    const int bit = (oy * M + ox) * VEC + k;
                    v.d[k] = ((rb[bit >> 5] >> (bit & 31)) & 1u) ? v.d[k] * mask_scale : 0.f;
                }
            } else if (mask) {
                const VF mk = mask[off];
                _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] = mk.d[k] > 0.f ? v.d[k] * mask_scale : 0.f;
            }
            vstore_nt<VEC>(y + off, v);          // (non-temporal, with wino_output_kernel's: transforms 13.04 -> 12.83 ms per step, profiles/r04_nt_stores_ab.txt)
        }
}

// ---- the same gather, followed at once by the NEXT layer's weight-gradient transform ---------------------------------------------
// Inside a block the data gradient of conv L is the output gradient of conv L-1 at the same resolution and on the same 6x6 tile grid:
// the 6x6 region a thread has just gathered is exactly the tile whose dM = A dZ A^T the weight gradient (and the adjoint data gradient) of
// L-1 needs.  This kernel applies the ReLU bits of L-1 to the region and transforms it in registers, so dZ of L-1 is neither written
// nor read back (2 x 4 N H W C bytes per layer pair): 100 loads + 64 stores per thread instead of (100 + 36) + (36 + 64).
// The region is transformed row by row (r[oy][:] = dZ[oy][:] A^T takes the place of tt[oy][:]) to stay within the registers of two waves per SIMD.
template <int VEC>
__global__ __launch_bounds__(256, 2) void wino_dgrad_output_dout_kernel(const VF* __restrict__ dv, const unsigned* __restrict__ rbits_in,
                                                                        VF* __restrict__ dm, int N, int H, int W, int C4, long long slab_v, long long slab_m)
{
    constexpr int M = 6, A = 8;
    typedef WinoMat<6, 3> WM;
    constexpr int RW = (M * M * VEC + 31) / 32;
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    const long long o = ti.t * C4 + ti.c;
    unsigned rb[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) rb[j] = rbits_in[o * RW + j];
    VF tt[M][A];                           // tt[i][b] = sum_a B^T(a, i+1) dV[a][b]   (patch rows 1..6)
#pragma unroll
    for (int b = 0; b < A; ++b) {
        VF col[A];
#pragma unroll
        for (int a = 0; a < A; ++a) col[a] = dv[(a * A + b) * slab_v + o];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            VF s = f4zero();
#pragma unroll
            for (int a = 0; a < A; ++a) if (WM::bt(a, i + 1) != 0.f) s = f4fma(WM::bt(a, i + 1), col[a], s);
            tt[i][b] = s;
        }
    }
    const bool up = ti.ty > 0, down = ti.ty + 1 < th, left = ti.tx > 0, right = ti.tx + 1 < tw;
    // the neighbours' border contributions (28 values per thread) wait in LDS, in a slot of the thread's own, for the loop that adds them: held in
    // registers next to tt and r they spilled 29 registers to scratch, and that scratch traffic reached the fabric (WRITE_SIZE: 1217 MB per conv3
    // launch for 992 MB of dM).  [slot][thread]: consecutive lanes, consecutive 8-byte words.
    __shared__ VF park[28][256];
    VF (*const pk)[256] = reinterpret_cast<VF (*)[256]>(&park[0][threadIdx.x]);      // pk[slot][0] = this thread's slot
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const bool okr = e == 0 ? up : down, okc = e == 0 ? left : right;
        const long long orow = o + (e == 0 ? -(long long)tw : (long long)tw) * C4, ocol = o + (e == 0 ? -1 : 1) * (long long)C4;
        const int pa = e == 0 ? 7 : 0;
        const float edge = WM::bt(pa, pa);
        VF r[A], c[A];
#pragma unroll
        for (int k = 0; k < A; ++k) { r[k] = okr ? dv[(pa * A + k) * slab_v + orow] : f4zero(); c[k] = okc ? dv[(k * A + pa) * slab_v + ocol] : f4zero(); }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            VF sr = f4zero(), sc = f4zero();
#pragma unroll
            for (int k = 0; k < A; ++k) if (WM::bt(k, j + 1) != 0.f) { sr = f4fma(WM::bt(k, j + 1), r[k], sr); sc = f4fma(WM::bt(k, j + 1), c[k], sc); }
            _Pragma("unroll") for (int v = 0; v < VEC; ++v) { sr.d[v] *= edge; sc.d[v] *= edge; }
            pk[e * M + j][0] = sr; pk[2 * M + e * M + j][0] = sc;
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const bool okf = f == 0 ? left : right;
            const int pb = f == 0 ? 7 : 0;
            const long long oc = orow + (f == 0 ? -1 : 1) * (long long)C4;
            VF v = (okr && okf) ? dv[(pa * A + pb) * slab_v + oc] : f4zero();
            _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] *= edge * WM::bt(pb, pb);
            pk[4 * M + e * 2 + f][0] = v;
        }
    }
    VF r[M][A];                            // r[oy][b] = sum_ox dZ[oy][ox] A^T(ox, b)
#pragma unroll
    for (int oy = 0; oy < M; ++oy) {
        VF z[M];
#pragma unroll
        for (int ox = 0; ox < M; ++ox) {
            VF v = f4zero();
#pragma unroll
            for (int b = 0; b < A; ++b) if (WM::bt(b, ox + 1) != 0.f) v = f4fma(WM::bt(b, ox + 1), tt[oy][b], v);
            if (oy == 0)     { const VF t_ = pk[ox][0];             _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += t_.d[k]; }
            if (oy == M - 1) { const VF t_ = pk[M + ox][0];         _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += t_.d[k]; }
            if (ox == 0)     { const VF t_ = pk[2 * M + oy][0];     _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += t_.d[k]; }
            if (ox == M - 1) { const VF t_ = pk[3 * M + oy][0];     _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += t_.d[k]; }
            if ((oy == 0 || oy == M - 1) && (ox == 0 || ox == M - 1)) { const VF t_ = pk[4 * M + (oy == 0 ? 0 : 1) * 2 + (ox == 0 ? 0 : 1)][0]; _Pragma("unroll") for (int k = 0; k < VEC; ++k) v.d[k] += t_.d[k]; }
            const bool inside = M * ti.ty + oy < H && M * ti.tx + ox < W;              // partial edge tiles
            _Pragma("unroll") for (int k = 0; k < VEC; ++k) {
                const int bit = (oy * M + ox) * VEC + k;
                v.d[k] = (inside && ((rb[bit >> 5] >> (bit & 31)) & 1u)) ? v.d[k] : 0.f;
            }
            z[ox] = v;
        }
#pragma unroll
        for (int b = 0; b < A; ++b) {
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < M; ++k) if (WM::at(k, b) != 0.f) s = f4fma(WM::at(k, b), z[k], s);
            r[oy][b] = s;
        }
    }
    VF* dp = dm + o;
#if FCN8S_ST16_GATHER
    if constexpr (VEC == 2 && A % 2 == 0) {
        if ((C4 & 1) == 0) {
#pragma unroll
            for (int b = 0; b < A; b += 2)
#pragma unroll
                for (int a = 0; a < A; ++a) {          // two neighbouring positions (a, b), (a, b + 1) per 16-byte store (store_pair16)
                    VF s0 = f4zero(), s1 = f4zero();
#pragma unroll
                    for (int k = 0; k < M; ++k) if (WM::at(k, a) != 0.f) { s0 = f4fma(WM::at(k, a), r[k][b], s0); s1 = f4fma(WM::at(k, a), r[k][b + 1], s1); }
                    store_pair16<true>(dp + (a * A + b) * slab_m, slab_m, s0, s1);
                }
            return;
        }
    }
#endif
#pragma unroll
    for (int b = 0; b < A; ++b)
#pragma unroll
        for (int a = 0; a < A; ++a) {          // dM[a][b] = sum_oy A^T(oy, a) r[oy][b]
            VF s = f4zero();
#pragma unroll
            for (int k = 0; k < M; ++k) if (WM::at(k, a) != 0.f) s = f4fma(WM::at(k, a), r[k][b], s);
            vstore_nt<VEC>(dp + (a * A + b) * slab_m, s);      // (non-temporal, as in wino_dout_kernel)
        }
}

// ---- dg_sub = G^T dU_sub G, scattered back to the taps (3a+i, 3b+j) < KS of the KS x KS filter gradient -----------------
template <int M, int R>
__global__ void wino_dfilter_kernel(const float* du, float* dw, int Cin, int Cout, int KS, int nsub)
{
    constexpr int A = WinoMat<M, R>::A;
    const long long cc = (long long)Cin * Cout, total = cc * nsub * nsub, ucc = total;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int sub = (int)(i / cc); const long long e = i - sub * cc;
        const int sa = sub / nsub, sb = sub - sa * nsub;
        float t[R][A];
#pragma unroll
        for (int b = 0; b < A; ++b) {
            float col[A];
#pragma unroll
            for (int a = 0; a < A; ++a) col[a] = du[(long long)(a * A + b) * ucc + i];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < A; ++k) s = fmaf(WinoMat<M, R>::g(k, r), col[k], s);
                t[r][b] = s;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < A; ++k) s = fmaf(t[r][k], WinoMat<M, R>::g(k, q), s);
                const int ky = R * sa + r, kx = R * sb + q;
                if (ky < KS && kx < KS) dw[(long long)(ky * KS + kx) * cc + e] = s;
            }
    }
}

// ---- adjoint data gradient of the 7x7 layer (2 x 2 grid of 4x4 sub-filters through F(4x4,4x4)) --------------------------------
// Forward: tile (ty, tx), sub-filter (sa, sb) reads the 7x7 patch at (4 ty + 4 sa - 3, 4 tx + 4 sb - 3), V = B^T d B, and the GEMM
// sums over (sub, ci).  Backward: dV[xi][t][sub * Cin + ci] = dM[xi][t] U[xi]^T comes from the transposed-B GEMM, each patch gets
// B dV B^T, and the patches (stride 4, size 7) overlap-add into dx.  As a gather: one thread = one 4x4 region (Ry, Rx) of dx x VEC
// channels; a patch row range [4u - 3, 4u + 3], u = ty' + sa, meets the region's rows for u = Ry (patch rows 3..6 = region rows 0..3)
// and u = Ry + 1 (patch rows 0..2 = region rows 1..3), each reached by (ty', sa) = (u, 0) and (u - 1, 1): 4 x 4 = 16 patches per
// region, each transformed only in the rows / columns the region needs.  dV is read ~4x (L2), 205 MB at 16 x 1024x512.
template <int KY, int KX, int VEC>
static __device__ __forceinline__ void sub44_gather(const VF* __restrict__ dv, long long slab, int ldv, int C4, const TileIdx& ti, int th, int tw,
                                                    VF (&out)[4][4])
{
    typedef WinoMat<4, 4> WM;
    constexpr int SA = KY & 1, DTY = KY == 0 ? 0 : KY == 1 ? -1 : KY == 2 ? 1 : 0, A0 = KY < 2 ? 3 : 0, R0 = KY < 2 ? 0 : 1, NR = KY < 2 ? 4 : 3;
    constexpr int SB = KX & 1, DTX = KX == 0 ? 0 : KX == 1 ? -1 : KX == 2 ? 1 : 0, B0 = KX < 2 ? 3 : 0, S0 = KX < 2 ? 0 : 1, NS = KX < 2 ? 4 : 3;
    const int ty = ti.ty + DTY, tx = ti.tx + DTX;
    if ((unsigned)ty >= (unsigned)th || (unsigned)tx >= (unsigned)tw) return;
    const VF* p = dv + (((long long)ti.n * th + ty) * tw + tx) * ldv + (SA * 2 + SB) * C4 + ti.c;
    VF tmp[NR][7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        VF g[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) g[i] = p[(i * 7 + j) * slab];
#pragma unroll
        for (int r = 0; r < NR; ++r) {                 // rows: dd[a][.] = sum_i B^T[i][a] dV[i][.]
            VF t = f4zero();
#pragma unroll
            for (int i = 0; i < 7; ++i) if (WM::bt(i, A0 + r) != 0.f) t = f4fma(WM::bt(i, A0 + r), g[i], t);
            tmp[r][j] = t;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NS; ++c) {                 // columns: dd[a][b] = sum_j (.)[a][j] B^T[j][b]
            VF t = out[R0 + r][S0 + c];
#pragma unroll
            for (int j = 0; j < 7; ++j) if (WM::bt(j, B0 + c) != 0.f) t = f4fma(WM::bt(j, B0 + c), tmp[r][j], t);
            out[R0 + r][S0 + c] = t;
        }
}
template <int VEC>
__global__ __launch_bounds__(256) void wino_dgrad_output_sub44_kernel(const VF* __restrict__ dv, VF* __restrict__ dx, int N, int H, int W, int C4, long long slab)
{
    const int th = H / 4, tw = W / 4;
    const TileIdx ti = tile_index(th, tw, C4, N);
    if (!ti.ok) return;
    const int ldv = C4 * 4;
    VF out[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[r][c] = f4zero();
#define FCN8S_S44ROW(KY_) sub44_gather<KY_, 0, VEC>(dv, slab, ldv, C4, ti, th, tw, out); sub44_gather<KY_, 1, VEC>(dv, slab, ldv, C4, ti, th, tw, out); \
                          sub44_gather<KY_, 2, VEC>(dv, slab, ldv, C4, ti, th, tw, out); sub44_gather<KY_, 3, VEC>(dv, slab, ldv, C4, ti, th, tw, out)
    FCN8S_S44ROW(0); FCN8S_S44ROW(1); FCN8S_S44ROW(2); FCN8S_S44ROW(3);
#undef FCN8S_S44ROW
    VF* o = dx + (((long long)ti.n * H + 4 * ti.ty) * W + 4 * ti.tx) * C4 + ti.c;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) o[(r * W + c) * C4] = out[r][c];
}

#undef f4fma
#undef f4zero
#undef VF

// ---- launchers (tile = 2 or 4; KS = 3, or 7 = grid of r x r sub-filters, r = wino_r(7)) ------------------------------------
// sub-filter size used for a KS x KS kernel: 3 for the 3x3 layers; fc6 (7x7): 4 (2x2 grid of 4x4 sub-filters, F(4x4,4x4); a 3x3 grid of
// 3x3 sub-filters through F(4x4,3x3) was measured slower in round 1: 20.25 instead of 12.25 multiplies per output)
int wino_r(int KS) { return KS == 3 ? 3 : 4; }
int wino_nsub(int KS) { const int r = wino_r(KS); return (KS + r - 1) / r; }
// 32-bit words of the ReLU bit mask a [N,H,W,C] output of tile size `tile` needs (launch_wino_output's rbits_out / rbits_in)
size_t wino_rbits_words(int tile, int N, int H, int W, int C)
{
    const int vec = tile == 2 ? 4 : 2;
    const size_t T = (size_t)N * ((H + tile - 1) / tile) * ((W + tile - 1) / tile);
    return T * (size_t)(C / vec) * (size_t)((tile * tile * vec + 31) / 32);
}
int wino_alpha(int tile, int KS) { return tile + (tile == 6 ? 3 : wino_r(KS)) - 1; }
// Distance in floats between the slabs of two Winograd positions of a [P][T][C] tensor.  T*C alone is a large power of two
// for this network (conv1_2: 2^25 floats): the 36 stores of one tile would then hit the same HBM channel and bank at the
// same time.  The skew (4 KiB + 256 B) staggers the slabs across channels.
long long wino_slab(long long T, int C)
{
    return T * C + 1088;
}
void launch_wino_filter(int tile, const float* w, float* u, int Cin, int Cout, int KS, hipStream_t s, int transpose_out)
{
    const int nsub = wino_nsub(KS);
    const int g = wcap((long long)Cin * Cout * nsub * nsub);
    if (nsub != 1) transpose_out = 0;
    if (tile == 6)                    hipLaunchKernelGGL((wino_filter_kernel<6, 3>), dim3(g), dim3(256), 0, s, w, u, Cin, Cout, KS, nsub, transpose_out);
    else if (tile == 4 && wino_r(KS) == 4) hipLaunchKernelGGL((wino_filter_kernel<4, 4>), dim3(g), dim3(256), 0, s, w, u, Cin, Cout, KS, nsub, transpose_out);
    else if (tile == 4)               hipLaunchKernelGGL((wino_filter_kernel<4, 3>), dim3(g), dim3(256), 0, s, w, u, Cin, Cout, KS, nsub, transpose_out);
    else                              hipLaunchKernelGGL((wino_filter_kernel<2, 3>), dim3(g), dim3(256), 0, s, w, u, Cin, Cout, KS, nsub, transpose_out);
}
void launch_wino_dgrad_output(const float* dv, const float* addend, const float* mask, float mask_scale, const unsigned* rbits_in, float* y,
                              int N, int H, int W, int C, hipStream_t s)
{
    constexpr int M_ = 6;
    const long long T = (long long)N * ((H + M_ - 1) / M_) * ((W + M_ - 1) / M_);
    // two channels per lane (one per lane was measured slower, and the ReLU bit words are laid out per (tile, channel pair))
    hipLaunchKernelGGL((wino_dgrad_output_kernel<2>), tile_grid(N, (H + M_ - 1) / M_, (W + M_ - 1) / M_, C / 2), dim3(256), 0, s,
                       (const VecF<2>*)dv, (const VecF<2>*)addend, (const VecF<2>*)mask, mask_scale, rbits_in, (VecF<2>*)y, N, H, W, C / 2, wino_slab(T, C) / 2);
}
// dv: [49][T][4 * C] (T = N * H/4 * W/4 tiles, columns = sub-filter x channel), dx: [N,H,W,C]; H, W % 4 == 0, C % 2 == 0
void launch_wino_dgrad_output_dout(const float* dv, const unsigned* rbits_in, float* dm, int N, int H, int W, int C, hipStream_t s)
{
    constexpr int M_ = 6;
    const long long T = (long long)N * ((H + M_ - 1) / M_) * ((W + M_ - 1) / M_);
    g_last_kernel = "wino_dgrad_output_dout_kernel<2>";
    hipLaunchKernelGGL((wino_dgrad_output_dout_kernel<2>), tile_grid(N, (H + M_ - 1) / M_, (W + M_ - 1) / M_, C / 2), dim3(256), 0, s,
                       (const VecF<2>*)dv, rbits_in, (VecF<2>*)dm, N, H, W, C / 2, wino_slab(T, C) / 2, wino_slab(T, C) / 2);
}

void launch_wino_dgrad_output_sub44(const float* dv, float* dx, int N, int H, int W, int C, hipStream_t s)
{
    const long long T = (long long)N * (H / 4) * (W / 4);
    hipLaunchKernelGGL((wino_dgrad_output_sub44_kernel<2>), tile_grid(N, H / 4, W / 4, C / 2), dim3(256), 0, s,
                       (const VecF<2>*)dv, (VecF<2>*)dx, N, H, W, C / 2, wino_slab(T, 4 * C) / 2);
}
bool launch_wino_input_dout(int tile, const float* dy, float* v, float* dm, int N, int H, int W, int C, hipStream_t s, const unsigned char* pidx)
{
    if ((tile != 4 && tile != 6) || H % 2 || W % 2 || C % 2) return false;
    const int th = (H + tile - 1) / tile, tw = (W + tile - 1) / tile;
#define FCN8S_WFUSE(M_, V_, P_) hipLaunchKernelGGL((wino_input_dout_kernel<M_, V_, P_>), tile_grid(N, th, tw, C / V_), dim3(256), 0, s, \
        (const VecF<V_>*)dy, (VecF<V_>*)v, (VecF<V_>*)dm, N, H, W, C / V_, wino_slab((long long)N * th * tw, C) / V_, pidx)
    if (tile == 4) { if (pidx) FCN8S_WFUSE(4, 2, true); else FCN8S_WFUSE(4, 2, false); }
    else           { if (pidx) FCN8S_WFUSE(6, 1, true); else FCN8S_WFUSE(6, 1, false); }      // 8x8 + 8x6 values per lane: one channel per lane
#undef FCN8S_WFUSE
    return true;
}
void launch_wino_input(int tile, const float* x, float* v, int N, int H, int W, int C, int KS, hipStream_t s, unsigned* rbits_out)
{
    if (KS != 3) rbits_out = nullptr;
    const int nsub = wino_nsub(KS), pad = (KS - 1) / 2, n2 = nsub * nsub;
#define FCN8S_WIN(M_, V_, R_) hipLaunchKernelGGL((wino_input_kernel<M_, V_, R_>), tile_grid(N, (H + M_ - 1) / M_, (W + M_ - 1) / M_, C / V_, n2), dim3(256), 0, s, \
        (const VecF<V_>*)x, (VecF<V_>*)v, N, H, W, C / V_, pad, nsub, wino_slab((long long)N * ((H + M_ - 1) / M_) * ((W + M_ - 1) / M_), C * n2) / V_, rbits_out)
    if (tile == 6)                    FCN8S_WIN(6, 2, 3);
    else if (tile == 4 && wino_r(KS) == 4) FCN8S_WIN(4, 2, 4);
    else if (tile == 4)               FCN8S_WIN(4, 2, 3);
    else                              FCN8S_WIN(2, 4, 3);
#undef FCN8S_WIN
}
// conv1_2's V straight from the preprocessed image x0 [N,H,W,4], conv1_1's padded kernel w4 [9][4][64] and bias (64 channels; see wino_input_conv1_kernel)
void launch_wino_input_conv1(const float* x0, const float* w4, const float* bias, float* v, int N, int H, int W, hipStream_t s, unsigned* rbits_out)
{
    const int th = (H + 5) / 6, tw = (W + 5) / 6;
    g_last_kernel = "wino_input_conv1_kernel";
    const dim3 grid((unsigned)((tw + 7) / 8), (unsigned)(N * th));
    const long long slab = wino_slab((long long)N * th * tw, 64) / 2;
    if (rbits_out) hipLaunchKernelGGL(wino_input_conv1_kernel<true>, grid, dim3(256), 0, s, (const float4*)x0, w4, bias, (VecF<2>*)v, N, H, W, slab, rbits_out);
    else           hipLaunchKernelGGL(wino_input_conv1_kernel<false>, grid, dim3(256), 0, s, (const float4*)x0, w4, bias, (VecF<2>*)v, N, H, W, slab, rbits_out);
}
// F(6x6,3x3) input transform that also fills the interior of the padded bf16 copy xb [N][H + 2][W + 2][C] (see wino_input_kernel, XB)
void launch_wino_input_xb(const float* x, float* v, unsigned short* xb, int N, int H, int W, int C, hipStream_t s, unsigned* rbits_out)
{
    const int th = (H + 5) / 6, tw = (W + 5) / 6;
    g_last_kernel = "wino_input_kernel<6, 2, 3, true>";
    hipLaunchKernelGGL((wino_input_kernel<6, 2, 3, true>), tile_grid(N, th, tw, C / 2, 1), dim3(256), 0, s,
                       (const VecF<2>*)x, (VecF<2>*)v, N, H, W, C / 2, 1, 1, wino_slab((long long)N * th * tw, C) / 2, rbits_out, (unsigned*)xb);
}
void launch_wino_output(int tile, const float* m, const float* bias, const float* addend, const float* mask, float mask_scale,
                        int relu, float* y, int N, int H, int W, int C, int dropout, float keep, unsigned long long seed,
                        unsigned int stream_id, hipStream_t s, float* pool, unsigned char* pidx, int KS, unsigned* rbits_out, const unsigned* rbits_in)
{
#define FCN8S_WOUT(M_, V_, D_, R_) if (pool) FCN8S_WOUT2(M_, V_, D_, R_, true, false); else FCN8S_WOUT2(M_, V_, D_, R_, false, false)
#define FCN8S_WOUT_NT(M_, V_, D_, R_) if (pool) FCN8S_WOUT2(M_, V_, D_, R_, true, true); else FCN8S_WOUT2(M_, V_, D_, R_, false, true)
#define FCN8S_WOUT2(M_, V_, D_, R_, P_, NT_) hipLaunchKernelGGL((wino_output_kernel<M_, V_, D_, R_, P_, NT_>), tile_grid(N, (H + M_ - 1) / M_, (W + M_ - 1) / M_, C / V_), dim3(256), 0, s, \
        (const VecF<V_>*)m, (const VecF<V_>*)bias, (const VecF<V_>*)addend, (const VecF<V_>*)mask, mask_scale, relu, (VecF<V_>*)y, N, H, W, C / V_, keep, seed, stream_id, \
        wino_slab((long long)N * ((H + M_ - 1) / M_) * ((W + M_ - 1) / M_), C) / V_, (VecF<V_>*)pool, pidx, rbits_out, rbits_in)
    const bool big = (double)N * H * W * C * (pool ? 1.0 : 4.0) >= 64e6;          // bytes of what this launch writes (the pool, or the activation): beyond any cache's use to the next kernel
    if (tile == 6)                    { if (dropout) FCN8S_WOUT(6, 2, true, 3); else if (big) FCN8S_WOUT_NT(6, 2, false, 3); else FCN8S_WOUT(6, 2, false, 3); }
    else if (tile == 4 && wino_r(KS) == 4) { if (dropout) FCN8S_WOUT(4, 2, true, 4); else FCN8S_WOUT(4, 2, false, 4); }
    else if (tile == 4)               { if (dropout) FCN8S_WOUT(4, 2, true, 3); else FCN8S_WOUT(4, 2, false, 3); }
    else                              { if (dropout) FCN8S_WOUT(2, 4, true, 3); else FCN8S_WOUT(2, 4, false, 3); }
#undef FCN8S_WOUT
#undef FCN8S_WOUT_NT
#undef FCN8S_WOUT2
}
// m: M of conv L [64][T][C] (its GEMM's output), v: V of conv L+1 [64][T][C]; both F(6x6,3x3) on [N,H,W,C]; C % 64 == 0.  Returns false
// if the shape is not covered (the caller then runs the two kernels).
// always = false: not when the launch would have to cut images into row ranges shorter than four tile rows to fill the chip (a range
// recomputes one halo row above and one below it): a single 1024x512 image, conv5_x at any batch.  Measured with the kernel as it is now
// (6 spilled registers; its first build spilled 52 and lost on every launch with row ranges): 16 x 1024x512 58.9-59.1 against 59.7-59.8 ms
// per step (transforms 13.26 against 13.96), 4 x 2048x1024 58.7 against 59.0-59.3; one image, forced: 1.91 against 1.88 ms per prediction.
bool launch_wino_out_in(const float* m, const float* bias, float* v, unsigned* rbits_out, int N, int H, int W, int C, hipStream_t s, bool always)
{
    if (C % 64 || !bias) return false;
    const int th = (H + 5) / 6, tw = (W + 5) / 6, C4 = C / 2;
    const long long T = (long long)N * th * tw;
    // 14 emitted tile columns x 32 channel pairs per block, or 6 x 64: whichever wastes fewer columns on this width (ring + last strip)
    const int cost14 = (tw + 13) / 14 * 16, cost6 = (tw + 5) / 6 * 8;
    const bool wide = C % 128 != 0 || cost14 <= cost6;
    // (block shapes measured on the seven launches of a 16 x 1024x512 step, profiles/r04_fuse_out_in_ab.txt: 512 threads as 14 x 32 + ring
    //  2735 us, 6 x 64 + ring 2718, 256 threads as 6 x 32 or 14 x 16 + ring, two blocks per CU, 2819 / 2832)
    const int TX = wide ? 14 : 6, CPB = wide ? 32 : 64, threads = 512;
    const int strips = (tw + TX - 1) / TX, groups = C4 / CPB;
    // enough blocks to fill the chip: cut an image's tile rows into ranges when N x strips x groups is small
    const long long want = 256LL * (512 / threads);
    long long cols = (long long)strips * groups * N;
    int chunks = (int)((want + cols - 1) / cols); if (chunks < 1) chunks = 1; if (chunks > th) chunks = th;
    const int rpb = (th + chunks - 1) / chunks;
    chunks = (th + rpb - 1) / rpb;
    if (!always && chunks > 1 && rpb < 4) return false;
    const dim3 grid((unsigned)(strips * groups), (unsigned)(N * chunks));
    g_last_kernel = "wino_out_in_kernel";
#define FCN8S_OI(CPB_, NCOL_) hipLaunchKernelGGL((wino_out_in_kernel<CPB_, NCOL_>), grid, dim3(CPB_ * NCOL_), 0, s, (const VecF<2>*)m, (const VecF<2>*)bias, (VecF<2>*)v, rbits_out, N, H, W, C4, wino_slab(T, C) / 2, rpb)
    if (CPB == 32) FCN8S_OI(32, 16);
    else FCN8S_OI(64, 8);
#undef FCN8S_OI
    return true;
}
void launch_wino_dout(int tile, const float* dy, float* dm, int N, int H, int W, int C, hipStream_t s, int KS, const unsigned char* pidx)
{
#define FCN8S_WDOUT(M_, V_, R_) hipLaunchKernelGGL((wino_dout_kernel<M_, V_, R_, false>), tile_grid(N, (H + M_ - 1) / M_, (W + M_ - 1) / M_, C / V_), dim3(256), 0, s, \
        (const VecF<V_>*)dy, (VecF<V_>*)dm, N, H, W, C / V_, wino_slab((long long)N * ((H + M_ - 1) / M_) * ((W + M_ - 1) / M_), C) / V_, nullptr)
    // (8-byte lanes; 16-byte lanes were measured 5 % slower for the plain variant and 7 % slower for the pool-routing one)
    if (tile == 6 && pidx)       hipLaunchKernelGGL((wino_dout_kernel<6, 2, 3, true>), tile_grid(N, (H + 5) / 6, (W + 5) / 6, C / 2), dim3(256), 0, s,
                                                         (const VecF<2>*)dy, (VecF<2>*)dm, N, H, W, C / 2, wino_slab((long long)N * ((H + 5) / 6) * ((W + 5) / 6), C) / 2, pidx);
    else if (tile == 6)               FCN8S_WDOUT(6, 2, 3);
    else if (tile == 4 && wino_r(KS) == 4) FCN8S_WDOUT(4, 2, 4);
    else if (tile == 4)               FCN8S_WDOUT(4, 2, 3);
    else                              FCN8S_WDOUT(2, 4, 3);
#undef FCN8S_WDOUT
}
void launch_wino_dfilter(int tile, const float* du, float* dw, int Cin, int Cout, int KS, hipStream_t s)
{
    const int nsub = wino_nsub(KS);
    const int g = wcap((long long)Cin * Cout * nsub * nsub);
    if (tile == 6)                    hipLaunchKernelGGL((wino_dfilter_kernel<6, 3>), dim3(g), dim3(256), 0, s, du, dw, Cin, Cout, KS, nsub);
    else if (tile == 4 && wino_r(KS) == 4) hipLaunchKernelGGL((wino_dfilter_kernel<4, 4>), dim3(g), dim3(256), 0, s, du, dw, Cin, Cout, KS, nsub);
    else if (tile == 4)               hipLaunchKernelGGL((wino_dfilter_kernel<4, 3>), dim3(g), dim3(256), 0, s, du, dw, Cin, Cout, KS, nsub);
    else                              hipLaunchKernelGGL((wino_dfilter_kernel<2, 3>), dim3(g), dim3(256), 0, s, du, dw, Cin, Cout, KS, nsub);
}

}  // namespace fcn8s
