// bf16-operand / fp32-accumulate SAME convolution for the two dense contractions fc6 (7x7, 512 -> 4096) and
// fc7 (1x1, 4096 -> 4096) on v_mfma_f32_32x32x16_bf16 -- BASELINE.json config 5 ("bf16 fwd / fp32 accum with MFMA
// fc6/fc7").  Optional precision mode (FCN8S_PREC_BF16_FC); the default path stays exact fp32.
//
// Semantics (what the oracle restates): both operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32),
// products and sums are fp32, bias / ReLU / dropout are applied in fp32, the output tensor is fp32.
//
// Data layout:
//   activations  fp32 NHWC in HBM; converted to a bf16 copy once per layer (f32_to_bf16_kernel) or, without that copy, on the
//                way into LDS ([row][k], 80-byte rows -> conflict-free ds_read_b128 of one lane's 8 consecutive k);
//   weights      re-laid out once per forward pass by w_to_bf16_tiles_kernel into bf16 K-tile-major blocks
//                wt[k / 32][cout][k % 32]: the B tile of a block (128 couts x 32 k) is one contiguous 8 KB run.
// Block = 256 threads (4 wave64) -> 128 pixels x 128 couts, wave tile 64 x 64 = 2 x 2 MFMA tiles, 8 MFMAs per wave
// per K-tile, global -> register -> LDS double buffering with one barrier per K-tile.
#include "fcn8s_internal.h"
#include <string>
#include <algorithm>

namespace fcn8s {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BFK = 32;          // K-tile depth (bf16 elements)

// w[K][Cout] fp32 (HWIO flattened: K = (ky, kx, ci)) -> wt[K/32][Cout][32] bf16
__global__ __launch_bounds__(256) void w_to_bf16_tiles_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt, int K, int Cout)
{
    __shared__ float tile[BFK][65];
    const int k0 = blockIdx.y * BFK, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (tid >> 6) + 4 * j, c = tid & 63;
        tile[k][c] = (c0 + c < Cout) ? w[(long long)(k0 + k) * Cout + c0 + c] : 0.f;
    }
    __syncthreads();
    const int c = tid >> 2, kq = (tid & 3) * 8;
    if (c0 + c < Cout) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)tile[kq + i][c];
        *reinterpret_cast<bf16x8*>(wt + ((long long)blockIdx.y * Cout + c0 + c) * BFK + kq) = o;
    }
}

void launch_w_to_bf16_tiles(const float* w, unsigned short* wt, int K, int Cout, hipStream_t s)
{
    dim3 grid((unsigned)((Cout + 63) / 64), (unsigned)(K / BFK));
    hipLaunchKernelGGL(w_to_bf16_tiles_kernel, grid, dim3(256), 0, s, w, wt, K, Cout);
}

// activations fp32 -> bf16 (RNE), 8 elements per thread: the GEMM then fetches half the bytes per A tile from L2 (with fp32 A tiles
// the 128 x 128 kernel was L2-bandwidth-bound: 24 KB per K-tile and block)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float4* __restrict__ x, bf16x8* __restrict__ y, long long n8)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = x[2 * i], b = x[2 * i + 1];
        bf16x8 o;
        o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
        o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
        y[i] = o;
    }
}
void launch_f32_to_bf16(const float* x, unsigned short* y, long long n, hipStream_t s)       // n % 8 == 0
{
    long long b = (n / 8 + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)b), dim3(256), 0, s, (const float4*)x, (bf16x8*)y, n / 8);
}

static __device__ __forceinline__ unsigned xcd_run(unsigned p, unsigned total)
{
    const unsigned q = total >> 3, r = total & 7u, xcd = p & 7u, i = p >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// ABF16: p.xh holds the activations already converted to bf16 (launch_f32_to_bf16); else fp32 p.x is converted on the way into LDS
template <int BM, int BN, bool ABF16>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const Bf16ConvArgs p)
{
    constexpr int LDK = BFK + 8;                   // bf16 elements per LDS row (80 B)
    constexpr int TM = BM / 2 / 32, TN = BN / 2 / 32;
    constexpr int AEL = ABF16 ? 8 : 4;             // A elements per 16-byte global load
    constexpr int A_LD = BM * (BFK / AEL) / 256;   // 16-byte global loads of A per thread and K-tile
    constexpr int B_LD = BN * (BFK / 8) / 256;     // 16-byte global loads of B per thread and K-tile
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (BM + BN) * LDK];
    unsigned short* As = smem;
    unsigned short* Bs = smem + 2 * BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned ntm = (unsigned)((p.M + BM - 1) / BM), ntn = (unsigned)(p.Cout / BN);
    const unsigned lid = xcd_run(blockIdx.x, gridDim.x);
    // tile order chosen by the launcher: whichever operand panel is the expensive one to re-fetch stays put behind
    // one XCD's L2 (fc6: the 205 MB filter bank -> M fastest; fc7: the 134 MB activation -> N fastest)
    const unsigned tmi = p.m_fastest ? lid % ntm : lid / ntn;
    const unsigned tni = p.m_fastest ? lid / ntm : lid % ntn;
    const long long m0 = (long long)tmi * BM;
    const int n0 = (int)tni * BN;
    const int HW = p.H * p.W, pad = (p.K - 1) / 2;

    int a_y[A_LD], a_x[A_LD];
    long long a_img[A_LD];
    bool a_ok[A_LD], a_val[A_LD], a_ldok[A_LD];
    const void* a_ptr[A_LD];
    const int a_c4 = (tid % (BFK / AEL)) * AEL;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int row = (tid + i * 256) / (BFK / AEL);
        const long long m = m0 + row;
        a_ok[i] = m < p.M;
        const long long mm = a_ok[i] ? m : 0;
        const int n = (int)(mm / HW), r = (int)(mm - (long long)n * HW);
        a_y[i] = r / p.W; a_x[i] = r - a_y[i] * p.W;
        a_img[i] = (long long)n * HW;
    }
    int f_ci0 = 0, f_ty = 0, f_tx = 0;
    auto set_tap = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_y[i] + f_ty - pad, ix = a_x[i] + f_tx - pad;
            a_val[i] = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const long long pix = a_val[i] ? a_img[i] + (long long)iy * p.W + ix : 0;
            a_ptr[i] = ABF16 ? (const void*)(p.xh + pix * p.Cin + a_c4) : (const void*)(p.x + pix * p.Cin + a_c4);
        }
    };
    set_tap();
    const unsigned short* b_ptr = p.wt + (long long)n0 * BFK + tid * 8;     // + kt * Cout * 32 per K-tile, + i * 2048 per slot

    f32x4 ra[A_LD];                                // ABF16: the same 16 bytes hold 8 bf16
    bf16x8 rb[B_LD];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            ra[i] = ABF16 ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned short*>(a_ptr[i]) + f_ci0)
                          : *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a_ptr[i]) + f_ci0);
            a_ldok[i] = a_val[i];
        }
        const unsigned short* bp = b_ptr + (long long)kt * p.Cout * BFK;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) rb[i] = *reinterpret_cast<const bf16x8*>(bp + i * 2048);
        f_ci0 += BFK;
        if (f_ci0 == p.Cin) {
            f_ci0 = 0;
            if (++f_tx == p.K) { f_tx = 0; ++f_ty; }
            set_tap();                  // one tap past the end computes pointers that are never dereferenced
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int row = (tid + i * 256) / (BFK / AEL);
            f32x4 v = ra[i];
            if (!a_ldok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};          // (all-zero bits are +0 in bf16 too)
            if (ABF16) *reinterpret_cast<f32x4*>(&As[(buf * BM + row) * LDK + a_c4]) = v;
            else {
                bf16x4 h;
                h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
                *reinterpret_cast<bf16x4*>(&As[(buf * BM + row) * LDK + a_c4]) = h;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            *reinterpret_cast<bf16x8*>(&Bs[(buf * BN + (f >> 2)) * LDK + (f & 3) * 8]) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const unsigned short* A = As + (buf * BM + wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5) * 8;
        const unsigned short* B = Bs + (buf * BN + wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < BFK / 16; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[tm] = *reinterpret_cast<const bf16x8*>(A + tm * 32 * LDK + ks * 16);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<const bf16x8*>(B + tn * 32 * LDK + ks * 16);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
        }
    };

    const int nkt = p.K * p.K * p.Cin / BFK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        compute(cur);
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue (fp32): bias, ReLU, dropout keyed by the element offset -- the same Philox stream as the fp32 path
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                const long long off = m * p.Cout + col;
                float v = acc[tm][tn][r] + bv;
                if (p.relu) v = v > 0.f ? v : 0.f;
                if (p.dropout) v = philox_uniform((unsigned long long)off, p.seed, p.stream_id) < p.keep_prob ? v / p.keep_prob : 0.f;
                p.y[off] = v;
            }
        }
    }
}

// =====================================================================================================================================
// 256 x 256 tile, 8 waves, LDS-DMA, two staggered wave groups (round 3; tools/labs/bf16_lab.hip has the stand-alone version and its diagnosis)
// =====================================================================================================================================
// y[M][Cout] = epilogue( A[M][K] * Wt[Cout][K]^T ) with K = (ty, tx, ci): A row m, tap (ty, tx) is pixel (y + ty, x + tx) of a ZERO-PADDED bf16
// copy of the activations (launch_f32_to_bf16_padded), so that every tap of every row is an in-bounds, unconditional load -- LDS-DMA has no
// predication to offer -- and Wt is the kernel transposed to [Cout][K] in bf16 (launch_w_to_bf16_t): both operands are k-contiguous and their
// LDS images are 64-byte rows (BK = 32 bf16) with the XOR-swizzled 16-byte chunks of igemm.hip's A image.
//   * five 32 KB stages = all 160 KB of LDS, four K-tiles in flight, counted vmcnt (12 / 8 / 4 / 0), one barrier per tick;
//   * the two row groups (waves 0-3: rows 0..127, waves 4-7: rows 128..255; one wave of each group per SIMD) are staggered by one
//     barrier: while one group reads its fragments of K-tile kt (12 ds_read_b128), the other issues the 16 MFMAs of its previous tile;
//   * 128 x 64 output per wave = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs), 196 VGPRs in all, no spills.
// Lab, random data in [-1, 1): 1040-1065 TFLOP/s at fc7's shape, 940 at fc6's (the 128 x 128 kernel below: 680 / 880 in the model).  The same
// loop without its loads reaches 1200, its MFMAs alone 1320 -- the data-dependent power ceiling of the bf16 pipe (guide 5.4 rule 25), not 2500.
namespace {
constexpr int G_BM = 256, G_BN = 256, G_BK = 32, G_ROWB = G_BK * 2, G_ABYTES = G_BM * G_ROWB;

static __device__ __forceinline__ void glds16b(const void* sbase, unsigned voff, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory", "m0");
}
template <int N> static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}

// The transposed bf16 kernels are stored as K-chunk PLANES, wt[K / 32][Cout][32]: the 64-byte slice (one output channel, 32 consecutive k) that is one LDS row of a
// K-tile lies next to the slices of the neighbouring output channels, so an LDS-DMA instruction (16 rows) reads 1 KB of contiguous memory -- whole 128-byte lines.
// With wt[Cout][K] every row was half a line whose other half belongs to the next K-tile: a lab that paired the rows into whole lines (same bytes, wrong data)
// cut the DMA-only time of fc6's forward product from 1.21 to 0.95 ms (profiles/r05_bf16_conv_tile_ab.txt).
#ifndef W_PLANES
#define W_PLANES 1
#endif
// w[K][Cout] fp32 (HWIO flattened) -> wt bf16 (planes [K / 32][Cout][32]; W_PLANES 0: [Cout][K])
__global__ __launch_bounds__(256) void w_to_bf16_t_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt, int K, int Cout)
{
    __shared__ float tile[32][65];
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (tid >> 6) + 4 * j, c = tid & 63;
        tile[k][c] = (c0 + c < Cout) ? w[(long long)(k0 + k) * Cout + c0 + c] : 0.f;
    }
    __syncthreads();
    const int c = tid >> 2, kq = (tid & 3) * 8;
    if (c0 + c < Cout) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)tile[kq + i][c];
        if (W_PLANES) *reinterpret_cast<bf16x8*>(wt + ((long long)blockIdx.y * Cout + c0 + c) * 32 + kq) = o;
        else *reinterpret_cast<bf16x8*>(wt + (long long)(c0 + c) * K + k0 + kq) = o;
    }
}
void launch_w_to_bf16_t(const float* w, unsigned short* wt, int K, int Cout, hipStream_t s)      // K % 32 == 0
{
    dim3 grid((unsigned)((Cout + 63) / 64), (unsigned)(K / 32));
    hipLaunchKernelGGL(w_to_bf16_t_kernel, grid, dim3(256), 0, s, w, wt, K, Cout);
}

// x [N][H][W][C] fp32 -> xp [N][H + 2 pad][W + 2 pad][C] bf16 (RNE), zero border; 8 channels per thread
// ps8: plane stride of xp in 8-element units (channel-chunk planes [C / 32][rows][32], see Bf16Conv256Args), 0 = [rows][C]
__global__ __launch_bounds__(256) void f32_to_bf16_padded_kernel(const float4* __restrict__ x, bf16x8* __restrict__ xp, int N, int H, int W, int C8, int pad, long long ps8)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long long total = (long long)N * Hp * Wp * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8); long long t = i / C8;
        const int xx = (int)(t % Wp) - pad; t /= Wp;
        const int yy = (int)(t % Hp) - pad; const int n = (int)(t / Hp);
        bf16x8 o;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const long long src = (((long long)n * H + yy) * W + xx) * C8 + c;
            const float4 a = x[2 * src], b = x[2 * src + 1];
            o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
            o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.f;
        }
        if (ps8) xp[(long long)(c >> 2) * ps8 + (i / C8) * 4 + (c & 3)] = o; else xp[i] = o;
    }
}
void launch_f32_to_bf16_padded(const float* x, unsigned short* xp, int N, int H, int W, int C, int pad, hipStream_t s, long long ps)     // C % 8 == 0 (planes: C % 32 == 0)
{
    const long long total = (long long)N * (H + 2 * pad) * (W + 2 * pad) * (C / 8);
    long long b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1;
    g_last_kernel = "f32_to_bf16_padded_kernel";
    hipLaunchKernelGGL(f32_to_bf16_padded_kernel, dim3((unsigned)b), dim3(256), 0, s, (const float4*)x, (bf16x8*)xp, N, H, W, C / 8, pad, ps / 8);
}

// The same conversion for an output gradient dY, with the column sums of dY (the layer's bias gradient, exact fp32) taken on the way: the backward pass
// would otherwise read the fp32 tensor twice (conversion + launch_colsum; 1.3 ms per step at 4 x 2048x1024).  A block = 256 / C8 pixel lanes x C8
// channel octets walks the interior pixels with a block-uniform stride; each thread keeps eight running sums, the block adds its pixel lanes in LDS in
// lane order and stores one partial row, and launch_colsum adds the partial rows (fixed order) into db (+=): reproducible.
__global__ __launch_bounds__(256) void f32_to_bf16_padded_colsum_kernel(const float4* __restrict__ x, bf16x8* __restrict__ xp, float* __restrict__ partial,
                                                                        int H, int W, int C8, int pad, int lanes, int rows_per_block, int nrows, long long ps8)
{
    __shared__ float red[256 * 8];
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int c = threadIdx.x % C8, pl = threadIdx.x / C8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pl < lanes) {
        // a block owns `rows_per_block` image rows (n, y): no per-pixel index arithmetic beyond an add (one division per row)
        for (int r = blockIdx.x * rows_per_block; r < (blockIdx.x + 1) * rows_per_block && r < nrows; ++r) {
            const int n = r / H, yy = r - n * H;
            const float4* src = x + (long long)r * W * C8 * 2;
            const long long t0 = ((long long)n * Hp + yy + pad) * Wp + pad;         // padded position of the row's first pixel
            bf16x8* dst = xp + t0 * C8;
            for (int xx = pl; xx < W; xx += lanes) {
                const float4 a = src[2 * (xx * C8 + c)], b = src[2 * (xx * C8 + c) + 1];
                bf16x8 o;
                o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
                o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
                if (ps8) xp[(long long)(c >> 2) * ps8 + (t0 + xx) * 4 + (c & 3)] = o; else dst[xx * C8 + c] = o;
                acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
    __syncthreads();
    if (pl == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = acc[i];
            for (int l = 1; l < lanes; ++l) v += red[(l * C8 + c) * 8 + i];
            partial[(long long)blockIdx.x * C8 * 8 + c * 8 + i] = v;
        }
    }
}
// the interior of xp is written here; its border must be zero already (a per-layer buffer zeroed at allocation).  db[c] += sum over pixels of x[., c]
bool launch_f32_to_bf16_padded_colsum(const float* x, unsigned short* xp, float* db, int N, int H, int W, int C, int pad, hipStream_t s, long long ps)
{
    const int C8 = C / 8;
    if (C % 8 || C8 > 256) return false;
    const int lanes = 256 / C8;
    const int nrows = N * H;
    int rpb = (16 * lanes + W - 1) / W;                  // >= 16 pixels per thread
    if (rpb < 1) rpb = 1;
    while ((nrows + rpb - 1) / rpb > 2048) rpb *= 2;     // (the partial rows are added up by launch_colsum: block-order, reproducible)
    const int blocks = (nrows + rpb - 1) / rpb;
    float* partial = det_scratch(s, (size_t)blocks * C);
    if (!partial) return false;
    g_last_kernel = "f32_to_bf16_padded_colsum_kernel";
    hipLaunchKernelGGL(f32_to_bf16_padded_colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)x, (bf16x8*)xp, partial, H, W, C8, pad, lanes, rpb, nrows, ps / 8);
    launch_colsum(partial, db, blocks, C, s);            // db[c] += sum over the blocks' partial rows
    return true;
}

// BN = 256 / 128 / 64 output channels per block (round 5: the 64- and 128-channel layers of blocks 1-2 and every data gradient run on this kernel
// too).  The two row groups of 128 rows stay; inside a group the four waves are laid out WR x (4 / WR): BN = 256 -> 1 x 4 (a wave owns 128 rows x 64
// columns, 4 x 2 accumulators, the round-3 kernel), BN = 128 -> 2 x 2 (64 x 64, 2 x 2), BN = 64 -> 4 x 1 (32 x 64, 1 x 2).  The B image of a stage has BN
// rows: waves whose chunks lie beyond it issue no B loads (the vmcnt accounting is per wave: NB = its number of B instructions).
// Rows m >= M (a partial last row tile) read the tile's first row's window and are not stored.  Epilogue: bias, ReLU, dropout as before; for the data gradients an
// optional addend (the skip path's gradient) and the ReLU mask of the layer input (`mask` > 0, an fp32 activation tensor of the output's shape).
// Stages: five for BN = 256 (all 160 KB of LDS, one block per CU, the round-3 kernel); four / three for BN = 64 / 128 (80 / 72 KB: TWO blocks per CU, so
// that one block's prologue and its 64 - 128 KB epilogue run under the other's K loop -- with 18 .. 36 K-tiles per tile in the 64- and 128-channel
// layers that is a third of a block's life).
template <int BN>
__global__ __launch_bounds__(512, BN == 256 ? 2 : 4) void conv_bf16_256_kernel(const Bf16Conv256Args p)
{
    constexpr int NS = BN == 256 ? 5 : (BN == 128 ? 3 : 4);                       // LDS stages; NS - 1 K-tiles in flight
    constexpr int WR = BN == 256 ? 1 : (BN == 128 ? 2 : 4), WCN = 4 / WR;         // waves of a group: WR along rows x WCN along columns
    constexpr int TM = 4 / WR, TN = 2;                                            // 32 x 32 accumulator tiles per wave
    constexpr int STAGE = (G_BM + BN) * G_ROWB;
    static_assert(WCN * 64 == BN, "a wave owns 64 columns");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];        // 160 / 72 / 80 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, wr = wq / WCN, wn = wq % WCN;
    const unsigned ntm = (unsigned)((p.M + G_BM - 1) / G_BM), ntn = (unsigned)(p.Cout / BN);
    const unsigned lid = xcd_run(blockIdx.x, gridDim.x);
    const unsigned tmi = p.m_fastest ? lid % ntm : lid / ntn, tni = p.m_fastest ? lid / ntm : lid % ntn;
    const long long m0 = (long long)tmi * G_BM; const int n0 = (int)tni * BN;
    const int HW = p.H * p.W, Hp = p.H + p.K - 1, Wp = p.W + p.K - 1, Ktot = p.K * p.K * p.Cin;

    // LDS-DMA: wave w fills 16-row chunks 2w, 2w + 1 of the A image and (if they exist) of the B image of a stage
    constexpr int NBMAX = 2;
    // B instructions of this wave (wave-uniform; BN = 256: always two, a compile-time constant as in the round-3 kernel)
    const int nb = BN == 256 ? 2 : ((wave * 2 + 1) * 16 < BN ? 2 : 0);
    unsigned a_voff[2], b_voff[NBMAX];
    // The padded copy may be larger than 4 GiB (64 x 1024x512 x 64 channels): the 32-bit per-lane byte offsets are taken from the window of the tile's FIRST row, whose
    // position pp0 goes into the 64-bit scalar base.  Rows of a tile are consecutive output pixels, so their windows lie within a few map rows of pp0.
    long long pp0;
    {
        const int n = (int)(m0 / HW), r = (int)(m0 - (long long)n * HW), y = r / p.W, x = r - y * p.W;
        pp0 = ((long long)n * Hp + y) * Wp + x;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + lane / 4, pc = lane % 4;
        const int lc = pc ^ ((row >> 2) & 3);                                  // logical 16-byte chunk stored at physical chunk pc
        long long m = m0 + row; if (m >= p.M) m = m0;
        const int n = (int)(m / HW), r = (int)(m - (long long)n * HW), y = r / p.W, x = r - y * p.W;
        const long long pp = ((long long)n * Hp + y) * Wp + x - pp0;          // top-left pixel of the row's tap window in the padded copy, from the tile's first
        a_voff[i] = p.xp_ps ? (unsigned)((pp * 32 + lc * 8) * 2) : (unsigned)((pp * p.Cin + lc * 8) * 2);
        b_voff[i] = W_PLANES ? (unsigned)(((row < BN ? row : 0) * 32 + lc * 8) * 2) : (unsigned)(((long long)(row < BN ? row : 0) * Ktot + lc * 8) * 2);
    }
    // split K (p.ksplit > 1, blockIdx.y): this block reduces K-tiles [kt0, kt0 + nkt) and stores its raw accumulators into slab blockIdx.y (launcher: only
    // launches whose epilogue is the identity -- fc6's data gradient: 64 row x column tiles of 256 x 256 for a 200 704-deep reduction)
    const int nkt_all = Ktot / G_BK;
    const int kt0 = p.ksplit > 1 ? (int)((long long)nkt_all * blockIdx.y / p.ksplit) : 0;
    const int nkt = (p.ksplit > 1 ? (int)((long long)nkt_all * (blockIdx.y + 1) / p.ksplit) : nkt_all) - kt0;
    const unsigned short* b_base = W_PLANES ? p.wt + (long long)n0 * 32 + (long long)kt0 * p.Cout * 32 : p.wt + (long long)n0 * Ktot + (long long)kt0 * G_BK;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // K-tiles are issued strictly in order, one call per tile: the tap position advances incrementally
    const int cpt = p.Cin / G_BK;                                      // K-tiles per tap
    int i_kt = 0, i_ci = (kt0 % cpt) * G_BK, i_tx = (kt0 / cpt) % p.K, i_ty = (kt0 / cpt) / p.K;
    auto issue = [&]() {
        const unsigned st = lds0 + (unsigned)((i_kt % NS) * STAGE);
        const unsigned short* ga = p.xp_ps ? p.xp + (long long)(i_ci >> 5) * p.xp_ps + (pp0 + (long long)i_ty * Wp + i_tx) * 32 : p.xp + (pp0 + (long long)i_ty * Wp + i_tx) * p.Cin + i_ci;
        const unsigned short* gb = b_base + (W_PLANES ? (long long)i_kt * p.Cout * 32 : (long long)i_kt * G_BK);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16b(ga, a_voff[i], st + (wave * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < NBMAX; ++i) if (i < nb) glds16b(gb, b_voff[i], st + G_ABYTES + (wave * 2 + i) * 1024);
        ++i_kt; i_ci += G_BK;
        if (i_ci == p.Cin) { i_ci = 0; if (++i_tx == p.K) { i_tx = 0; ++i_ty; } }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_row[TM], b_row[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a_row[tm] = grp * 128 + wr * (TM * 32) + tm * 32 + (lane & 31);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b_row[tn] = wn * 64 + tn * 32 + (lane & 31);
    bf16x8 af[2][TM], bfr[2][TN];
    auto load_frags = [&](int kt) {
        const unsigned char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int r = a_row[tm], pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                af[ks][tm] = *reinterpret_cast<const bf16x8*>(st + r * G_ROWB + pc * 16);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int r = b_row[tn], pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                bfr[ks][tn] = *reinterpret_cast<const bf16x8*>(st + G_ABYTES + r * G_ROWB + pc * 16);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads are DONE in this tick: the stage may be refilled two ticks later
    };
    auto mfma_phase = [&]() {
        // priority over the partner wave of the SIMD, which is in its load phase: without it that wave's ds_read / LDS-DMA / waitcnt
        // issue delays the MFMA stream (lab, same box: 900 -> 948 TFLOP/s at fc7's shape, 847 -> 902 at fc6's)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][tm], bfr[ks][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // this wave's pieces of K-tile kt have landed when at most (tiles still allowed in flight) x (2 + nb) of its loads are outstanding
    auto wait_n = [&](int tiles) {
        if (BN == 256 || nb == 2) { if (tiles >= 3) wait_vm<12>(); else if (tiles == 2) wait_vm<8>(); else if (tiles == 1) wait_vm<4>(); else wait_vm<0>(); }
        else { if (tiles >= 3) wait_vm<6>(); else if (tiles == 2) wait_vm<4>(); else if (tiles == 1) wait_vm<2>(); else wait_vm<0>(); }
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) if (t < nkt) issue();
    wait_n(nkt > NS - 2 ? NS - 2 : nkt - 1);
    __builtin_amdgcn_s_barrier();
    // tick t: group g runs step t - g; even steps read the fragments of K-tile step / 2, odd steps multiply them.  Even ticks 2 j issue the
    // LDS-DMA of K-tile j + 4 (its stage was last read in tick 2 j - 1); odd ticks 2 kt + 1 wait for K-tile kt + 1.  Each group runs its own
    // straight-line loop (one loop with per-tick branches made hipcc copy the accumulators around: 10x slower), and sched_barrier keeps
    // hipcc from hoisting a group's MFMAs above the barrier that opens its MFMA phase.
    auto tick_end = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    auto wait_tile = [&](int kt) {      // this wave's pieces of K-tile kt have landed; up to NS - 2 newer tiles may still be in flight
        const int newer = nkt - 1 - kt;
        wait_n(newer > NS - 2 ? NS - 2 : (newer < 0 ? 0 : newer));
    };
    if (grp == 0) {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + NS - 1 < nkt) issue();
            load_frags(kt);
            tick_end();
            mfma_phase();
            wait_tile(kt + 1);
            tick_end();
        }
        tick_end();
    } else {
        if (NS - 1 < nkt) issue();
        tick_end();
        for (int kt = 0; kt < nkt; ++kt) {
            load_frags(kt);
            wait_tile(kt + 1);
            tick_end();
            if (kt + NS < nkt) issue();
            mfma_phase();
            tick_end();
        }
    }

    if (p.ksplit > 1) {
        float* part = p.part + (long long)blockIdx.y * p.M * p.Cout;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + grp * 128 + wr * (TM * 32) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M) part[m * p.Cout + n0 + wn * 64 + tn * 32 + (lane & 31)] = acc[tm][tn][r];
                }
        return;
    }
    // epilogue (fp32): bias, skip-path addend, ReLU, the ReLU mask of a data gradient, dropout keyed by the element offset -- the same Philox stream as the fp32
    // path.  Through LDS, row-major, like the flat-position kernel's: the MFMA layout gives a lane one column of sixteen scattered rows (4-byte stores and mask
    // loads -- 0.12-0.14 ms of fc7's 0.40-0.42 with the K loop switched off, profiles/r05_bf16_conv_tile_ab.txt); each wave parks one 32 x 32 tile at a time
    // (raw accumulators) in its own patch of the free stage buffers, then a lane owns (row, 8 consecutive channels): 16-byte loads and stores.
    // p.yb: the consumer's padded bf16 copy of this output (pixel (n, y, x) -> pixel (n, y + pad, x + pad) of [N][H + 2 pad][W + 2 pad][Cout], border never written).
    __builtin_amdgcn_s_barrier();                                          // (every wave is behind its last fragment reads: the loops end in a barrier)
    constexpr int LDP = 36;
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * LDP);
    const int prow0 = lane >> 2, pc8 = (lane & 3) * 8;
    const int Hq = p.H + 2 * p.yb_pad, Wq = p.W + 2 * p.yb_pad;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const long long mt = m0 + grp * 128 + wr * (TM * 32) + tm * 32;      // first row of this 32-row tile
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDP + (lane & 31)] = acc[tm][tn][r];
            __builtin_amdgcn_wave_barrier();
            const int col8 = n0 + wn * 64 + tn * 32 + pc8;
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = p.bias ? p.bias[col8 + k] : 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const long long m = mt + prow0 + 16 * it;
                if (m < p.M) {
                    const float4 u0 = *reinterpret_cast<const float4*>(&patch[(prow0 + 16 * it) * LDP + pc8]);
                    const float4 u1 = *reinterpret_cast<const float4*>(&patch[(prow0 + 16 * it) * LDP + pc8 + 4]);
                    float v[8] = {u0.x + bv[0], u0.y + bv[1], u0.z + bv[2], u0.w + bv[3], u1.x + bv[4], u1.y + bv[5], u1.z + bv[6], u1.w + bv[7]};
                    const long long off = m * p.Cout + col8;
                    if (p.addend) {
                        const float4 a0 = *reinterpret_cast<const float4*>(p.addend + off), a1 = *reinterpret_cast<const float4*>(p.addend + off + 4);
                        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
                    }
                    if (p.relu) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
                    }
                    if (p.mask) {
                        const float4 q0 = *reinterpret_cast<const float4*>(p.mask + off), q1 = *reinterpret_cast<const float4*>(p.mask + off + 4);
                        const float mk[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = mk[k] > 0.f ? v[k] * p.mask_scale : 0.f;
                    }
                    if (p.dropout) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = philox_uniform((unsigned long long)(off + k), p.seed, p.stream_id) < p.keep_prob ? v[k] / p.keep_prob : 0.f;
                    }
                    if (p.y) {
                        *reinterpret_cast<float4*>(p.y + off) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(p.y + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                    if (p.yb) {
                        const int n = (int)(m / HW), rem = (int)(m - (long long)n * HW), yy = rem / p.W, xx = rem - yy * p.W;
                        const long long q = ((long long)n * Hq + yy + p.yb_pad) * Wq + xx + p.yb_pad;
                        bf16x8 o;
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = (__bf16)v[k];
                        *reinterpret_cast<bf16x8*>(p.yb + (p.yb_ps ? (long long)(col8 >> 5) * p.yb_ps + q * 32 + (col8 & 31) : q * p.Cout + col8)) = o;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();                               // (the patch is this wave's own: the next tile overwrites it)
        }
    }
}

// The epilogue of the flat-position kernels, through LDS, row-major.  NW waves laid out WRN (rows) x WCN (columns), a wave owns TM x 2 accumulator tiles of
// 32 x 32: 32 TM positions x 64 columns; the block's row tile is tmi (BM = WRN * TM * 32 positions from q0), its columns start at n0.
// Epilogue through LDS, row-major: the MFMA layout gives a lane one column of 16 scattered rows -- 4-byte stores, 2-byte mask loads, and for the 64-channel
// layers (six K-tiles per tile) that epilogue WAS the kernel: conv1_2's forward pass took 1.24 ms with its MFMAs and LDS reads switched off and 1.29 with its
// LDS-DMA switched off, 1.34 complete (profiles/r05_bf16_conv_tile_ab.txt).  Each wave parks one 32 x 32 half of its tile (fp32, raw accumulators) in its own
// patch of the free stage buffers; then a lane owns (row, 8 consecutive channels): bias, skip-path addend, ReLU, the mask (16 bytes of the layer's bf16
// input copy, or 32 of the fp32 activation), two 16-byte fp32 stores and / or one 16-byte store into the consumer's bf16 copy, and the column sums.
// Border positions and positions beyond the last image: nothing stored in y, zeros in the copy.
template <int TM, int WCN, int WRN>
static __device__ __forceinline__ void conv_rows_epilogue(const Bf16Conv256Args& p, f32x16 (&acc)[TM][2], unsigned char* smem, int tid, long long q0, int n0, unsigned tmi)
{
    constexpr int NW = WCN * WRN, BN = WCN * 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave / WCN, wn = wave % WCN;
    const int Hp = p.H + 2, Wp = p.W + 2;
    const long long R = (long long)p.N * Hp * Wp;
    __syncthreads();                                                       // every wave is done reading the stage buffers
    constexpr int LDP = 36;
    float* patch = reinterpret_cast<float*>(smem) + wave * (32 * LDP);
    const long long HpWp = (long long)Hp * Wp;
    const int prow0 = lane >> 2, pc8 = (lane & 3) * 8;
    float csum[2][8];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int k = 0; k < 8; ++k) csum[tn][k] = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        long long qrow[2], pixv[2]; bool valid[2];
        {
            const long long qb = q0 + wr * (TM * 32) + tm * 32 + prow0;
            int n = (int)(qb / HpWp), rem = (int)(qb - (long long)n * HpWp), yy = rem / Wp, xx = rem - yy * Wp;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                qrow[it] = qb + 16 * it;
                valid[it] = qrow[it] < R && yy >= 1 && yy <= p.H && xx >= 1 && xx <= p.W;
                pixv[it] = ((long long)n * p.H + yy - 1) * p.W + xx - 1;
                xx += 16;
                while (xx >= Wp) { xx -= Wp; ++yy; }
                while (yy >= Hp) { yy -= Hp; ++n; }
            }
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDP + (lane & 31)] = acc[tm][tn][r];
            __builtin_amdgcn_wave_barrier();
            const int col8 = n0 + wn * 64 + tn * 32 + pc8;
            float bv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bv[k] = p.bias ? p.bias[col8 + k] : 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float4 u0 = *reinterpret_cast<const float4*>(&patch[(prow0 + 16 * it) * LDP + pc8]);
                const float4 u1 = *reinterpret_cast<const float4*>(&patch[(prow0 + 16 * it) * LDP + pc8 + 4]);
                float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
                if (valid[it]) {
                    const long long off = pixv[it] * p.Cout + col8;
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += bv[k];
                    if (p.addend) {
                        const float4 a0 = *reinterpret_cast<const float4*>(p.addend + off), a1 = *reinterpret_cast<const float4*>(p.addend + off + 4);
                        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
                    }
                    if (p.relu) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
                    }
                    // (the ReLU mask of a data gradient: the sign of the layer's padded bf16 INPUT copy, which has this kernel's geometry -- row q, half the bytes of
                    //  the fp32 activation and no pixel arithmetic; bf16 keeps fp32's exponent range, so x > 0 <=> bf16(x) > 0 for every normal x)
                    if (p.mask16) {
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 mk = *reinterpret_cast<const s16x8*>(p.mask16 + (p.mask16_ps ? (long long)(col8 >> 5) * p.mask16_ps + qrow[it] * 32 + (col8 & 31) : qrow[it] * p.Cout + col8));
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = mk[k] > 0 ? v[k] * p.mask_scale : 0.f;
                    } else if (p.mask) {
                        const float4 m0 = *reinterpret_cast<const float4*>(p.mask + off), m1 = *reinterpret_cast<const float4*>(p.mask + off + 4);
                        const float mk[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = mk[k] > 0.f ? v[k] * p.mask_scale : 0.f;
                    }
                    if (p.dropout) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = philox_uniform((unsigned long long)(off + k), p.seed, p.stream_id) < p.keep_prob ? v[k] / p.keep_prob : 0.f;
                    }
                    if (p.y) {
                        *reinterpret_cast<float4*>(p.y + off) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(p.y + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = 0.f;
                }
                // p.yb: the consumer is another 3 x 3 layer on the same map, so its padded bf16 copy has THIS geometry and position q of the output is row q of that copy
                // (rows q >= R of a last partial row tile lie in the zeroed guard rows behind the copy and are written with the zeros they hold)
                if (p.yb) {
                    bf16x8 o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = (__bf16)v[k];
                    *reinterpret_cast<bf16x8*>(p.yb + (p.yb_ps ? (long long)(col8 >> 5) * p.yb_ps + qrow[it] * 32 + (col8 & 31) : qrow[it] * p.Cout + col8)) = o;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) csum[tn][k] += v[k];
            }
            __builtin_amdgcn_wave_barrier();                               // (the patch is this wave's own: the next tile overwrites it)
        }
    }
    // p.colpart: the column sums of this tile's stored values (fp32, before any rounding: the consumer layer's bias gradient when this is a data gradient whose
    // fp32 output nobody reads) -- lane, wave, block in a fixed order; one partial row per row tile, added up by launch_colsum.
    if (p.colpart) {
        float* red = reinterpret_cast<float*>(smem) + NW * (32 * LDP);        // behind the patches
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float t = csum[tn][k];
                t += __shfl_xor(t, 4); t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);      // the 16 lanes that share these columns
                if (lane < 4) red[wave * 64 + tn * 32 + pc8 + k] = t;
            }
        __syncthreads();
        if (tid < BN) {
            const int cw_ = tid / 64, c = tid % 64;
            float t = 0.f;
#pragma unroll
            for (int r_ = 0; r_ < WRN; ++r_) t += red[(r_ * WCN + cw_) * 64 + c];
            p.colpart[(long long)tmi * p.Cout + n0 + tid] = t;
        }
    }
}

// 3 x 3 layers on the 64-column tile, round 5: the same product over FLAT positions of the padded map, one K-tile = one filter row.
// conv_bf16_256_kernel<64> fetches a 256-row A tile per tap -- nine times per channel chunk -- and runs 4 MFMAs per wave between two barriers: its matrix
// pipes were 0.21 busy (profiles/r05_c5_bf16_train_pmc_clock_summary.txt).  Here the rows of a tile are 256 consecutive positions q of the padded map
// [N][H + 2][W + 2] (border positions are computed and not stored: 0.3 % of them at 2048x1024, 9 % at 32x64), so the window of position q under tap
// (ty, tx) is row q + (ty - 1) Wp + (tx - 1) of the padded copy and the three taps of a filter row read the SAME 258 rows moved by one: a K-tile
// (ty, 32 channels) brings 272 rows of A once (17 LDS-DMA instructions) and the three 64 x 32 weight slices, and feeds 12 MFMAs per wave.
// A third of the A traffic, a third of the barriers.  Two stages (58 KB), two blocks per CU; plain loop, one barrier per K-tile.
template <int BN>
__global__ __launch_bounds__(512, 4) void conv_bf16_rows_kernel(const Bf16Conv256Args p)
{
    // BN = 64: 256 positions x 64 columns per block, eight waves of 32 x 64; BN = 128: 128 positions x 128 columns, waves 4 x 2 of 32 x 64
    constexpr int WCN = BN / 64, BM = 256 / WCN, AROWS = BM + 16, NAC = AROWS / 16, NBC = 3 * BN / 16;
    // LDS: THREE stages of the A image (the activation rows, from HBM / a cold L2: two K-tiles ahead) and TWO of the B image (the weight slices, L2-hot: one
    // K-tile ahead) -- 76.8 KB, still two blocks per CU.  With two stages of both, a tile's LDS-DMA had one K-tile of MFMAs to land in and did not: the loop
    // with its MFMAs switched off took 0.46 ms at conv4_2's shape, with its DMA switched off 0.50, complete 0.63 (profiles/r05_bf16_conv_tile_ab.txt).
    constexpr int ABYTES = AROWS * G_ROWB, BBYTES = 3 * BN * G_ROWB, NSA = 3, NSB = 2, BOFF = NSA * ABYTES;
    constexpr int NAI = (NAC + 7) / 8, NBI = (NBC + 7) / 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSA * ABYTES + NSB * BBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCN, wn = wave % WCN;
    const int Hp = p.H + 2, Wp = p.W + 2, Ktot = 9 * p.Cin;
    const long long R = (long long)p.N * Hp * Wp;
    const unsigned ntn = (unsigned)(p.Cout / BN);
    const unsigned lid = xcd_run(blockIdx.x, gridDim.x);
    const unsigned tmi = lid / ntn, tni = lid % ntn;               // column tiles of one row tile are neighbours: they share its A rows behind one L2
    const long long q0 = (long long)tmi * BM; const int n0 = (int)tni * BN;
    // A chunks (16 rows each) c = wave + 8 i < NAC; B chunks (tx, 16 couts) c = wave + 8 i < NBC
    unsigned a_voff[NAI], b_voff[NBI]; unsigned a_dst[NAI], b_dst[NBI]; int b_tx[NBI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int chunk = wave + 8 * i, row = chunk * 16 + lane / 4, pc = lane % 4, lc = pc ^ ((row >> 2) & 3);
        a_voff[i] = p.xp_ps ? (unsigned)((row * 32 + lc * 8) * 2) : (unsigned)(((long long)row * p.Cin + lc * 8) * 2);
        a_dst[i] = (unsigned)(chunk * 1024);
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int c0 = wave + 8 * i, c = c0 < NBC ? c0 : 0, tx = c / (BN / 16), c4 = c % (BN / 16), row = c4 * 16 + lane / 4, pc = lane % 4, lc = pc ^ ((row >> 2) & 3);
        b_voff[i] = W_PLANES ? (unsigned)((row * 32 + lc * 8) * 2) : (unsigned)(((long long)row * Ktot + lc * 8) * 2);
        b_dst[i] = (unsigned)(tx * (BN * G_ROWB) + c4 * 1024);
        b_tx[i] = tx;
    }
    // (row 0 of the A image of filter row ty is padded position q0 - Wp - 1 + ty Wp: the guard rows in front of the copy make that readable for q0 = 0)
    const unsigned short* a_base = p.xp + (q0 - Wp - 1) * (p.xp_ps ? 32 : p.Cin);
    const unsigned short* b_base = W_PLANES ? p.wt + (long long)n0 * 32 : p.wt + (long long)n0 * Ktot;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int nci = p.Cin / G_BK, nkt_all = 3 * nci;               // K-tile kt = (channel chunk kt / 3, filter row kt % 3)
    // split K (p.ksplit > 1, blockIdx.y; maps of a few hundred positions -- conv4_x / conv5_x of ONE image -- whose 10-70 blocks would each walk 48 K-tiles on a
    // mostly idle chip): this block reduces K-tiles [kbeg, kbeg + nkt) and stores its raw accumulators into slab blockIdx.y; rows_splitk_epilogue_kernel adds them
    const int kbeg = p.ksplit > 1 ? (int)((long long)nkt_all * blockIdx.y / p.ksplit) : 0;
    const int nkt = (p.ksplit > 1 ? (int)((long long)nkt_all * (blockIdx.y + 1) / p.ksplit) : nkt_all) - kbeg;
    auto issue_a = [&](int kt, int sa) {
        const int ci = (kt / 3) * G_BK, ty = kt % 3;
        const unsigned st = lds0 + (unsigned)(sa * ABYTES);
        const unsigned short* ga = p.xp_ps ? a_base + (long long)(ci >> 5) * p.xp_ps + (long long)ty * Wp * 32 : a_base + (long long)ty * Wp * p.Cin + ci;
#pragma unroll
        for (int i = 0; i < NAI; ++i) if (wave + 8 * i < NAC) glds16b(ga, a_voff[i], st + a_dst[i]);
    };
    auto issue_b = [&](int kt, int sb) {
        const int ci = (kt / 3) * G_BK, ty = kt % 3;
        const unsigned st = lds0 + (unsigned)(BOFF + sb * BBYTES);
#pragma unroll
        for (int i = 0; i < NBI; ++i) if (wave + 8 * i < NBC) glds16b(W_PLANES ? b_base + (((long long)(ty * 3 + b_tx[i]) * p.Cin + ci) >> 5) * p.Cout * 32 : b_base + (long long)(ty * 3 + b_tx[i]) * p.Cin + ci, b_voff[i], st + b_dst[i]);
    };
    // this wave's A instructions per K-tile (chunks wave, wave + 8, ...): what may stay in flight at the head of a tile is exactly the A image two tiles ahead
    int na = 0;
#pragma unroll
    for (int i = 0; i < NAI; ++i) na += (wave + 8 * i < NAC) ? 1 : 0;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int arow = wr * 32 + (lane & 31);
    issue_b(kbeg, 0); issue_a(kbeg, 0);
    if (nkt > 1) issue_a(kbeg + 1, 1);
    int sa = 0, sa2 = 2;                                           // A stage of tile kt / of tile kt + 2
    for (int kt = 0; kt < nkt; ++kt) {
        // in issue order the youngest instructions are A(kt + 1): everything older -- A(kt), B(kt) -- has landed once only they are left
        if (kt + 1 < nkt) { if (na == 3) wait_vm<3>(); else if (na == 2) wait_vm<2>(); else if (na == 1) wait_vm<1>(); else wait_vm<0>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                              // tile kt has landed everywhere; everybody is done reading tile kt - 1's stages
        const unsigned char* st = smem + sa * ABYTES;
        const unsigned char* stb = smem + BOFF + (kt & 1) * BBYTES;
        bf16x8 af[3][2], bfr[3][2][2];
        auto frags = [&](int tx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int r = arow + tx, pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                af[tx][ks] = *reinterpret_cast<const bf16x8*>(st + r * G_ROWB + pc * 16);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int rb = wn * 64 + tn * 32 + (lane & 31), pb = (2 * ks + (lane >> 5)) ^ ((rb >> 2) & 3);
                    bfr[tx][ks][tn] = *reinterpret_cast<const bf16x8*>(stb + tx * (BN * G_ROWB) + rb * G_ROWB + pb * 16);
                }
            }
        };
        auto mfmas = [&](int tx) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tx][ks], bfr[tx][ks][tn], acc[tn], 0, 0, 0);
        };
        // the next tile's LDS-DMA first, then the fragments of two taps together (12 reads in flight), the third tap's under the first tap's MFMAs -- pinned:
        // left alone the compiler keeps 72 registers and waits for one read in front of every MFMA (conv4_2's data gradient 0.68 -> 0.65 ms; all 18 reads
        // first: 0.67; the DMA issue moved behind the reads: 0.71)
        // Where the LDS-DMA of the coming tiles is issued: 0 = at the head of the tile, 1 = behind the first twelve fragment reads, 2 = behind the first tap's
        // MFMAs and the last reads, 3 = behind the second tap's MFMAs.  B behind the reads, A behind the first MFMAs (12) is the best of six placements by
        // 0.4 % in a same-box A/B (tools/ab_variants.sh; profiles/r05_bf16_conv_tile_ab.txt) -- what calls from different boxes had shown as +-3 % was the boxes.
#ifndef ROWS_VARIANT
#define ROWS_VARIANT 12
#endif
        constexpr int PB = ROWS_VARIANT / 10, PA = ROWS_VARIANT % 10;
        auto dma = [&](int pos) {
            if (PB == pos && kt + 1 < nkt) issue_b(kbeg + kt + 1, (kt + 1) & 1);
            if (PA == pos && kt + 2 < nkt) issue_a(kbeg + kt + 2, sa2);
        };
        dma(0);
        frags(0); frags(1);
        __builtin_amdgcn_sched_barrier(0);
        dma(1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        frags(2);
        __builtin_amdgcn_sched_barrier(0);
        dma(2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        dma(3);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(2);
        sa = sa + 1 == NSA ? 0 : sa + 1; sa2 = sa2 + 1 == NSA ? 0 : sa2 + 1;
    }
    if (p.ksplit > 1) {
        // raw accumulators of this K range, rows = flat positions of the padded map (border positions included: the reducing kernel knows which are pixels)
        const long long Rq = (long long)p.N * Hp * Wp;
        float* part = p.part + (long long)blockIdx.y * Rq * p.Cout;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long q = q0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (q < Rq) part[q * p.Cout + n0 + wn * 64 + tn * 32 + (lane & 31)] = acc[tn][r];
            }
        return;
    }
    {
        f32x16 accw[1][2] = {{acc[0], acc[1]}};
        conv_rows_epilogue<1, WCN, 8 / WCN>(p, accw, smem, tid, q0, n0, tmi);
    }
}

// The slabs of conv_bf16_rows_kernel's split-K form added in split order, with the forward epilogue of a 3 x 3 layer: row q of a slab is flat position q of the padded
// map [N][H + 2][W + 2]; interior positions get bias and ReLU and are stored as the fp32 pixel and / or as row q of the consumer's padded bf16 copy (same geometry),
// border positions store zeros into the copy.  8 columns per thread.
__global__ __launch_bounds__(256) void rows_splitk_epilogue_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, unsigned short* __restrict__ yb,
                                                                   long long yb_ps, int N, int H, int W, int Cout, int nsplit, int relu)
{
    const int c8n = Cout / 8, Hp = H + 2, Wp = W + 2;
    const long long R = (long long)N * Hp * Wp, total = R * c8n, slab = R * Cout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long q = i / c8n; const int col8 = (int)(i - q * c8n) * 8;
        const int n = (int)(q / ((long long)Hp * Wp)), rem = (int)(q - (long long)n * Hp * Wp), yy = rem / Wp, xx = rem - yy * Wp;
        const bool valid = yy >= 1 && yy <= H && xx >= 1 && xx <= W;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid) {
            for (int sp = 0; sp < nsplit; ++sp) {
                const float4 a = *reinterpret_cast<const float4*>(part + sp * slab + q * Cout + col8), b = *reinterpret_cast<const float4*>(part + sp * slab + q * Cout + col8 + 4);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] += bias ? bias[col8 + k] : 0.f; if (relu) v[k] = v[k] > 0.f ? v[k] : 0.f; }
            if (y) {
                const long long off = (((long long)n * H + yy - 1) * W + xx - 1) * Cout + col8;
                *reinterpret_cast<float4*>(y + off) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(y + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        if (yb) {
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (__bf16)v[k];
            *reinterpret_cast<bf16x8*>(yb + (yb_ps ? (long long)(col8 >> 5) * yb_ps + q * 32 + (col8 & 31) : q * Cout + col8)) = o;
        }
    }
}

// The split-K slabs of conv_bf16_256_kernel added in split order, with the FORWARD epilogue (bias, ReLU; the fp32 output and / or the consumer's bf16 copy of an
// unpadded map): fc6 / fc7 at batch 1 are 2 row tiles x 16 column tiles behind reductions of 784 / 128 K-tiles -- 32 blocks on 256 CUs -- and run split into
// slabs like fc6's data gradient (round 6: fc6 at one 1024x512 image 0.32 -> see profiles/r06_bf16_infer.txt).  8 columns per thread, reproducible.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, unsigned short* __restrict__ yb,
                                                              long long yb_ps, long long M, int Cout, int nsplit, int relu)
{
    const int c8n = Cout / 8;
    const long long total = M * c8n, slab = M * Cout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / c8n; const int col8 = (int)(i - m * c8n) * 8;
        const long long off = m * Cout + col8;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bias ? bias[col8 + k] : 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 a = *reinterpret_cast<const float4*>(part + sp * slab + off), b = *reinterpret_cast<const float4*>(part + sp * slab + off + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
        }
        if (y) { *reinterpret_cast<float4*>(y + off) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(y + off + 4) = make_float4(v[4], v[5], v[6], v[7]); }
        if (yb) {
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (__bf16)v[k];
            *reinterpret_cast<bf16x8*>(yb + (yb_ps ? (long long)(col8 >> 5) * yb_ps + m * 32 + (col8 & 31) : off)) = o;
        }
    }
}

// mode 0 never, 1 when it fills the chip (the round-3 rule, 256-column tiles only), 2 whenever the shapes allow, 3 = 2 with the 128- and 64-column
// tiles and a partial last row tile as well (the bf16_train mode)
bool conv_bf16_256_ok(long long M, int Cin, int Cout, int mode)
{
    if (mode == 0 || Cin % G_BK || Cin % 8) return false;
    if (mode >= 3) return Cout % 64 == 0 && M >= 1;
    if (M % G_BM || Cout % G_BN) return false;
    return mode >= 2 || (M / G_BM) * (Cout / G_BN) >= 128;         // fewer tiles than half the CUs: the 128 x 128 kernel fills the chip better
}

// rows of a row tile of conv_bf16_rows_kernel as launch_conv_bf16_256 picks it (= positions per partial row of `colpart`)
// Which form of conv_bf16_rows_kernel a 3 x 3 layer takes: 256 positions x 64 columns, or (rows_bn = 128, A/B) 128 x 128 where Cout % 128 == 0.
void conv_bf16_rows_tile(int Cout, int rows_bn, int* bn, int* bm)
{
    *bn = 64; *bm = 256;
    if (rows_bn == 128 && Cout % 128 == 0) { *bn = 128; *bm = 128; }
}
int conv_bf16_rows_bm(int Cout, int rows_bn) { int bn, bm; conv_bf16_rows_tile(Cout, rows_bn, &bn, &bm); return bm; }

bool launch_conv_bf16_256(const Bf16Conv256Args& a0, hipStream_t s)
{
    Bf16Conv256Args a = a0;
    a.M = (long long)a.N * a.H * a.W;
    if (!conv_bf16_256_ok(a.M, a.Cin, a.Cout, a.any_shape ? 3 : 2)) return false;
    // 32-bit per-lane byte offsets: inside a tile's window of the padded copy (a few map rows; the tile's position is in the 64-bit base), and over the weights
    if ((double)(G_BM + (a.K + 2) * (a.W + a.K - 1)) * (a.xp_ps ? 32 : a.Cin) * 2.0 >= 4294967296.0 || (double)a.Cout * a.K * a.K * a.Cin * 2.0 >= 4294967296.0) return false;
    const double abytes = 2.0 * a.M * a.K * a.K * a.Cin, bbytes = 2.0 * a.K * a.K * a.Cin * a.Cout;
    a.m_fastest = bbytes > abytes;             // the larger operand's panel stays put behind one XCD's L2 while the other one streams
    // a long reduction behind few output tiles and an identity epilogue (fc6's data gradient): keep the wide tile and split K over blockIdx.y into slabs that
    // a second kernel adds in split order (reproducible: no atomics)
    a.ksplit = 1;
    {
        const long long rt0 = (a.M + G_BM - 1) / G_BM;
        const long long nkt_all = (long long)a.K * a.K * a.Cin / G_BK;
        const bool plain = !a.bias && !a.addend && !a.mask && !a.relu && !a.dropout && !a.yb && a.y && !a.colpart;
        // ... and the forward pass of the same shapes at batch 1 (bias, ReLU, the consumer's copy of an UNPADDED map): the slabs are added by splitk_epilogue_kernel
        // (note: the bias is added FIRST there, before the slabs -- in the one-pass kernel it is added last; both are the fp32 sum of the same terms to round-off)
        const bool fwd_epi = !plain && a.K != 3 && !a.addend && !a.mask && !a.dropout && !a.colpart && (a.y || a.yb) && (!a.yb || a.yb_pad == 0) && a.Cout % 8 == 0;      // (3 x 3 layers: the flat-position kernel)
        if (a.any_shape && fwd_epi && a.Cout % 256 == 0 && rt0 * (a.Cout / 256) <= 64 && nkt_all >= 64) {
            long long ks = 256 / (rt0 * (a.Cout / 256));
            if (ks > 8) ks = 8;
            if (ks > nkt_all / 16) ks = nkt_all / 16;            // at least 16 K-tiles per slab
            float* part = ks >= 2 ? det_scratch(s, (size_t)(ks * a.M * a.Cout)) : nullptr;
            if (part) {
                Bf16Conv256Args k = a;
                k.ksplit = (int)ks; k.part = part;
                g_last_kernel = "conv_bf16_256_kernel<256>";
                hipLaunchKernelGGL(conv_bf16_256_kernel<256>, dim3((unsigned)(rt0 * (a.Cout / 256)), (unsigned)ks), dim3(512), 0, s, k);
                const long long n8 = a.M * (a.Cout / 8);
                long long blocks = (n8 + 255) / 256; if (blocks > 4096) blocks = 4096;
                hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, s, part, a.bias, a.y, a.yb, a.yb_ps, a.M, a.Cout, (int)ks, a.relu);
                return true;
            }
        }
        if (a.any_shape && plain && a.Cout % 256 == 0 && rt0 * (a.Cout / 256) < 128 && nkt_all >= 512) {
            long long ks = 256 / (rt0 * (a.Cout / 256));
            if (ks > 8) ks = 8;
            float* part = ks >= 2 ? det_scratch(s, (size_t)(ks * a.M * a.Cout)) : nullptr;
            if (part) {
                a.ksplit = (int)ks; a.part = part;
                g_last_kernel = "conv_bf16_256_kernel<256>";
                hipLaunchKernelGGL(conv_bf16_256_kernel<256>, dim3((unsigned)(rt0 * (a.Cout / 256)), (unsigned)ks), dim3(512), 0, s, a);
                launch_det_reduce(a.y, part, a.M, a.Cout, a.Cout, a.M * (long long)a.Cout, (int)ks, false, s);
                return true;
            }
        }
    }
    // the widest column tile that divides Cout -- unless that leaves CUs without a block (fc6's data gradient: 32 row tiles x 512 / 256 = 64 blocks for
    // 200 704-deep dot products): then the narrower tiles, which multiply the block count
    const long long rt = (a.M + G_BM - 1) / G_BM;
    int bn = a.Cout % 256 == 0 ? 256 : (a.Cout % 128 == 0 ? 128 : 64);
    if (a.any_shape) while (bn > 64 && rt * (a.Cout / bn) < 256) bn /= 2;
    // ... and short reductions (the 3 x 3 layers: 18 .. 144 K-tiles) take the narrow tiles anyway: those run two blocks per CU, so that one block's prologue
    // and epilogue hide under the other's K loop.  Measured at 4 x 2048x1024 (profiles/r05_bf16_conv_tile_ab.txt): data gradient of conv3_2 1.48 ms with
    // 256 columns, 1.17 with 128, 1.08 with 64; conv4_2 1.01 / 0.84 / 0.94; fc6 forward (784 K-tiles) 1.56 / 1.64 / 2.26.
    const long long ktot = (long long)a.K * a.K * a.Cin;
    const int cap = ktot <= 2304 ? 64 : (ktot <= 4608 ? 128 : 256);      // (K = 1 and K = 7 launches; the 3 x 3 layers of the training pass take conv_bf16_rows_kernel below)
    if (a.any_shape) while (bn > cap && bn > 64) bn /= 2;
    if (a.K == 3 && a.any_shape && a.guarded && (!a.yb || a.yb_pad == 1)) {
        const long long R = (long long)a.N * (a.H + 2) * (a.W + 2);
        // (the 128-column form, 128 positions x 128 columns, halves the A re-reads of the wide layers and measured 1-2 % SLOWER: 39.23 against 38.71 ms per step)
        int tbn, tbm; conv_bf16_rows_tile(a.Cout, a.rows_bn, &tbn, &tbm);
        const unsigned blocks = (unsigned)(((R + tbm - 1) / tbm) * (a.Cout / tbn));
        // few blocks behind many K-tiles (conv5_x of one image: 24 blocks x 48 K-tiles, 0.046 -> 0.030 ms; conv4_x's 72 blocks gained nothing from three slabs): split K over blockIdx.y into slabs, forward epilogue only
        // (bias, ReLU, fp32 pixels and / or the consumer's copy); the sum is bias LAST here as in the one-pass kernel, its terms in slab order
        const long long nkt_all = 3LL * (a.Cin / G_BK);
        if (tbn == 64 && blocks <= 48 && nkt_all >= 24 && !a.addend && !a.mask && !a.mask16 && !a.dropout && !a.colpart && (a.y || a.yb)) {
            long long ks = 256 / blocks; if (ks > 8) ks = 8; if (ks > nkt_all / 6) ks = nkt_all / 6;
            float* part = ks >= 2 ? det_scratch(s, (size_t)(ks * R * a.Cout)) : nullptr;
            if (part) {
                Bf16Conv256Args k = a; k.ksplit = (int)ks; k.part = part;
                g_last_kernel = "conv_bf16_rows_kernel<64>";
                hipLaunchKernelGGL(conv_bf16_rows_kernel<64>, dim3(blocks, (unsigned)ks), dim3(512), 0, s, k);
                const long long n8 = R * (a.Cout / 8);
                long long eb = (n8 + 255) / 256; if (eb > 4096) eb = 4096;
                hipLaunchKernelGGL(rows_splitk_epilogue_kernel, dim3((unsigned)eb), dim3(256), 0, s, part, a.bias, a.y, a.yb, a.yb_ps, a.N, a.H, a.W, a.Cout, (int)ks, a.relu);
                return true;
            }
        }
        if (tbn == 128) { g_last_kernel = "conv_bf16_rows_kernel<128>"; hipLaunchKernelGGL(conv_bf16_rows_kernel<128>, dim3(blocks), dim3(512), 0, s, a); }
        else { g_last_kernel = "conv_bf16_rows_kernel<64>"; hipLaunchKernelGGL(conv_bf16_rows_kernel<64>, dim3(blocks), dim3(512), 0, s, a); }
        return true;
    }
    if (!a.y || a.colpart) return false;       // (only the flat-position kernel runs without an fp32 output or takes column sums)
    const unsigned blocks = (unsigned)(rt * (a.Cout / bn));
    if (bn == 256) { g_last_kernel = "conv_bf16_256_kernel<256>"; hipLaunchKernelGGL(conv_bf16_256_kernel<256>, dim3(blocks), dim3(512), 0, s, a); }
    else if (bn == 128) { g_last_kernel = "conv_bf16_256_kernel<128>"; hipLaunchKernelGGL(conv_bf16_256_kernel<128>, dim3(blocks), dim3(512), 0, s, a); }
    else { g_last_kernel = "conv_bf16_256_kernel<64>"; hipLaunchKernelGGL(conv_bf16_256_kernel<64>, dim3(blocks), dim3(512), 0, s, a); }
    return true;
}

// w[K*K][Cin][Cout] fp32 (HWIO) -> wt[Cin][(K*K taps, flipped)][Cout] bf16: the kernel of the convolution that IS the data gradient of a SAME
// convolution, dX[p][ci] = sum_{t', co} dYpad[p + t'][co] * w[K*K - 1 - t'][ci][co], laid out [output channel = ci][k = (t', co)] for conv_bf16_256_kernel
__global__ __launch_bounds__(256) void w_to_bf16_flip_t_kernel(const float4* __restrict__ w, bf16x8* __restrict__ wt, int KK, int Cin, int Cout8)
{
    const long long total = (long long)Cin * KK * Cout8;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % Cout8); long long t = i / Cout8;
        const int tp = (int)(t % KK), ci = (int)(t / KK);
        const long long src = ((long long)(KK - 1 - tp) * Cin + ci) * Cout8 + c8;
        const float4 a = w[2 * src], b = w[2 * src + 1];
        bf16x8 o;
        o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
        o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
        if (W_PLANES) { const long long k = (long long)tp * Cout8 * 8 + c8 * 8; wt[((k >> 5) * Cin + ci) * 4 + ((k & 31) >> 3)] = o; }      // planes [k / 32][ci][32]
        else wt[i] = o;
    }
}
// The same through an LDS tile (planes layout, Cin % 32 == 0, Cout % 64 == 0): a block = one tap, 32 input channels, 64 output channels (two k-chunks); it reads
// 32 rows of 256 bytes and writes, per k-chunk, the 32 rows ci of that plane -- 2 KB of contiguous memory -- where the kernel above scatters 64-byte pieces
// Cin * 64 bytes apart.
__global__ __launch_bounds__(256) void w_to_bf16_flip_t_tiled_kernel(const float4* __restrict__ w, bf16x8* __restrict__ wt, int KK, int Cin, int Cout)
{
    __shared__ float tile[32][68];
    const int tp = blockIdx.z, ci0 = blockIdx.y * 32, co0 = blockIdx.x * 64, tid = threadIdx.x;
    const long long src = ((long long)(KK - 1 - tp) * Cin + ci0) * Cout + co0;        // w[(KK - 1 - tp)][ci0 ..][co0 ..]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = tid + 256 * j, r = f >> 4, c4 = f & 15;
        const float4 v = w[(src + (long long)r * Cout) / 4 + c4];
        tile[r][c4 * 4] = v.x; tile[r][c4 * 4 + 1] = v.y; tile[r][c4 * 4 + 2] = v.z; tile[r][c4 * 4 + 3] = v.w;
    }
    __syncthreads();
    const int kc = tid >> 7, ci = (tid >> 2) & 31, oct = tid & 3;
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)tile[ci][kc * 32 + oct * 8 + i];
    const long long k = (long long)tp * Cout + co0 + kc * 32;                          // first k of this chunk
    wt[((k >> 5) * Cin + ci0 + ci) * 4 + oct] = o;
}
void launch_w_to_bf16_flip_t(const float* w, unsigned short* wt, int K, int Cin, int Cout, hipStream_t s)      // Cout % 8 == 0
{
    if (W_PLANES && Cin % 32 == 0 && Cout % 64 == 0 && (long long)K * K <= 65535 && Cin / 32 <= 65535) {
        hipLaunchKernelGGL(w_to_bf16_flip_t_tiled_kernel, dim3((unsigned)(Cout / 64), (unsigned)(Cin / 32), (unsigned)(K * K)), dim3(256), 0, s, (const float4*)w, (bf16x8*)wt, K * K, Cin, Cout);
        return;
    }
    const long long total = (long long)Cin * K * K * (Cout / 8);
    long long b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1;
    hipLaunchKernelGGL(w_to_bf16_flip_t_kernel, dim3((unsigned)b), dim3(256), 0, s, (const float4*)w, (bf16x8*)wt, K * K, Cin, Cout / 8);
}

// =====================================================================================================================================
// Weight gradient with bf16-rounded operands, fp32 accumulation (round 5, the bf16_train mode)
// =====================================================================================================================================
//   dW[tap][ci][co] = sum over padded pixels q of  Xp[q + off(tap)][ci] * dYp[q][co],      off(ty, tx) = (ty - pad) * Wp + (tx - pad)
// Xp and dYp are the zero-bordered bf16 copies [N][H + 2 pad][W + 2 pad][C] the forward / data-gradient convolutions read anyway.  Because the
// border of dYp is zero, the sum may run over ALL padded positions q as one flat row index: a tap is a plain-row product A^T B whose A
// operand is Xp moved by a constant number of rows -- no per-pixel address arithmetic, no predicates (LDS-DMA has none to offer).  Both buffers
// carry zeroed guard rows in front and behind (pad * Wp + pad + 32), so that the moved / rounded-up row ranges stay inside the allocation and meet
// finite values (times zero).  Both operands are k-STRIDED ([row][channel]); the MFMA fragments (8 consecutive k of one channel per lane) come out of
// LDS through ds_read_b64_tr_b16, the hardware 4 x 4 transpose of 16-bit elements (layout and swizzle of tools/labs/planes_lab.hip, measured in round 4).
// Tile BM x BM channels (128, or 64 for the 64-channel layers), K-tile = 32 rows, four stages, LDS-DMA fills, one tap per blockIdx.z, the rows split
// over blockIdx.x / ntiles chunks whose partial tiles meet in fp32 atomics -- or, in deterministic mode, in one slab per chunk added in chunk order.
namespace {
static __device__ __forceinline__ void glds16w(const void* sbase, unsigned voff, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory", "m0");
}
}
template <int BM>
__global__ __launch_bounds__(256, BM == 128 ? 2 : 4) void wgrad_bf16_kernel(const Bf16WgradArgs p)
{
    constexpr int S = 4, ROWB = BM * 2, PLANE = 16 * ROWB, STAGE = 4 * PLANE;        // planes: A rows 0-15, A rows 16-31, B rows 0-15, B rows 16-31
    constexpr int NI = PLANE / 1024, RPI = 1024 / ROWB;                                // LDS-DMA instructions per plane, rows per instruction
    constexpr int T = BM / 64;                                                         // 32 x 32 tiles per wave and operand
    // 16-byte chunk c of row r sits at chunk c ^ SWZ (r & 3).  A transposing read's 32-lane group takes 64 contiguous bytes of each of four rows: with 128-byte
    // rows (BM = 64) a row's parity picks the bank half and SWZ = 2 the quarter; with 256-byte rows (BM = 128) every row starts at bank 0 and the four rows need
    // the four 64-byte quarters: SWZ = 4 (with 2, rows r and r + 1 share 16 banks: SQ_LDS_BANK_CONFLICT = half of the kernel's LDS cycles).  History: with the
    // copies stored [rows][C] the conflict-free swizzle made fc6's weight gradient 35 % SLOWER in the step (1.71 -> 2.3-2.4 ms, A/B/A on one box) and was
    // reverted; with the copies stored as channel-chunk planes it is 6.5 % faster (1.81 1.84 -> 1.69 1.72 ms, same-box A/B): what had hurt was the global side of
    // the permuted LDS-DMA rows, not the LDS reads (profiles/r05_bf16_conv_tile_ab.txt).
    constexpr int SWZ = BM == 128 ? 4 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntj = p.Cj / BM, ntiles = (p.Ci / BM) * ntj;
    const int tile = blockIdx.x % ntiles, ys = blockIdx.x / ntiles, tap = blockIdx.z;
    const int i0 = (tile / ntj) * BM, j0 = (tile % ntj) * BM;
    const long long t0 = (long long)ys * p.chunk;
    const long long t1 = t0 + p.chunk < p.R ? t0 + p.chunk : p.R;
    const int nkt = (int)((t1 - t0 + 31) / 32);                                       // (a last partial K-tile reads guard rows: zeros in B)
    const int ty = tap / p.K, tx = tap - ty * p.K, pad = (p.K - 1) / 2;
    const long long aoff = (long long)(ty - pad) * p.Wp + (tx - pad);
    // wave 0 / 1: A rows 0-15 / 16-31 of a K-tile, wave 2 / 3: B rows 0-15 / 16-31
    const int ld = wave < 2 ? p.Ci : p.Cj;
    const long long ps = wave < 2 ? p.a_ps : p.b_ps;               // channel-chunk planes [C / 32][rows][32] (0: [rows][C])
    const long long rstep = ps ? 32 : ld;                          // elements from one row to the next
    const unsigned short* mine = wave < 2 ? p.A + (ps ? (long long)(i0 >> 5) * ps + (t0 + aoff + (wave & 1) * 16) * 32 : (t0 + aoff + (wave & 1) * 16) * p.Ci + i0)
                                          : p.B + (ps ? (long long)(j0 >> 5) * ps + (t0 + (wave & 1) * 16) * 32 : (t0 + (wave & 1) * 16) * p.Cj + j0);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = i * RPI + lane / (ROWB / 16), slot = lane % (ROWB / 16), c = slot ^ (SWZ * (row & 3));
        voff[i] = ps ? (unsigned)(((long long)(c >> 2) * ps + row * 32 + (c & 3) * 8) * 2) : (unsigned)(((long long)row * ld + c * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int kt, int stage) {
        const unsigned short* g = mine + (long long)kt * 32 * rstep;
#pragma unroll
        for (int i = 0; i < NI; ++i) glds16w(g, voff[i], lds0 + stage * STAGE + wave * PLANE + i * 1024);
    };
    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // lane l of an MFMA operand: column (l & 31) of the 32-wide tile, k-half g = l >> 5 (rows 8g .. 8g+7 of a 16-row plane).  Its 16-lane group
    // q = l >> 4 covers columns 16 (q & 1) .. +15; inside the group lane i = l & 15 SUPPLIES the address of row 8g + 4r + i / 4, columns
    // 4 (i % 4) .. +3 of the group's 16 (8 bytes) and RECEIVES rows 8g + 4r .. +3 of column i.
    unsigned a_addr[T][2], b_addr[T][2];
    {
        const int q = lane >> 4, i = lane & 15, g = q >> 1;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = 8 * g + 4 * r + (i >> 2);
                const int ca = wm * (BM / 2) + t * 32 + 16 * (q & 1) + 4 * (i & 3), cb = wn * (BM / 2) + t * 32 + 16 * (q & 1) + 4 * (i & 3);
                a_addr[t][r] = (unsigned)(row * ROWB + (((ca >> 3) ^ (SWZ * (row & 3))) * 16) + (ca & 7) * 2);
                b_addr[t][r] = (unsigned)(row * ROWB + (((cb >> 3) ^ (SWZ * (row & 3))) * 16) + (cb & 7) * 2);
            }
    }
#pragma unroll
    for (int t = 0; t < S - 1; ++t) if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + S - 2 < nkt) { if (NI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else if (kt + 1 < nkt) { if (NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        const unsigned sb = lds0 + stage * STAGE;
        // (all transposing reads and their wait in ONE asm statement: the compiler treats an asm's outputs as ready when the statement ends)
        bf16x8 a0[T], a1[T], b0[T], b1[T];
        if constexpr (T == 2) {
            u32x2 r[16];
            asm volatile(
                "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %17\n\tds_read_b64_tr_b16 %2, %18\n\tds_read_b64_tr_b16 %3, %19\n\t"
                "ds_read_b64_tr_b16 %4, %16 offset:4096\n\tds_read_b64_tr_b16 %5, %17 offset:4096\n\tds_read_b64_tr_b16 %6, %18 offset:4096\n\tds_read_b64_tr_b16 %7, %19 offset:4096\n\t"
                "ds_read_b64_tr_b16 %8, %20 offset:8192\n\tds_read_b64_tr_b16 %9, %21 offset:8192\n\tds_read_b64_tr_b16 %10, %22 offset:8192\n\tds_read_b64_tr_b16 %11, %23 offset:8192\n\t"
                "ds_read_b64_tr_b16 %12, %20 offset:12288\n\tds_read_b64_tr_b16 %13, %21 offset:12288\n\tds_read_b64_tr_b16 %14, %22 offset:12288\n\tds_read_b64_tr_b16 %15, %23 offset:12288\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
                  "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                : "v"(sb + a_addr[0][0]), "v"(sb + a_addr[0][1]), "v"(sb + a_addr[T - 1][0]), "v"(sb + a_addr[T - 1][1]),
                  "v"(sb + b_addr[0][0]), "v"(sb + b_addr[0][1]), "v"(sb + b_addr[T - 1][0]), "v"(sb + b_addr[T - 1][1])
                : "memory");
#pragma unroll
            for (int t = 0; t < T; ++t) {
                a0[t] = __builtin_bit_cast(bf16x8, (u32x4){r[2 * t][0], r[2 * t][1], r[2 * t + 1][0], r[2 * t + 1][1]});
                a1[t] = __builtin_bit_cast(bf16x8, (u32x4){r[4 + 2 * t][0], r[4 + 2 * t][1], r[4 + 2 * t + 1][0], r[4 + 2 * t + 1][1]});
                b0[t] = __builtin_bit_cast(bf16x8, (u32x4){r[8 + 2 * t][0], r[8 + 2 * t][1], r[8 + 2 * t + 1][0], r[8 + 2 * t + 1][1]});
                b1[t] = __builtin_bit_cast(bf16x8, (u32x4){r[12 + 2 * t][0], r[12 + 2 * t][1], r[12 + 2 * t + 1][0], r[12 + 2 * t + 1][1]});
            }
        } else {
            u32x2 r[8];
            asm volatile(
                "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\t"
                "ds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %9 offset:2048\n\t"
                "ds_read_b64_tr_b16 %4, %10 offset:4096\n\tds_read_b64_tr_b16 %5, %11 offset:4096\n\t"
                "ds_read_b64_tr_b16 %6, %10 offset:6144\n\tds_read_b64_tr_b16 %7, %11 offset:6144\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                : "v"(sb + a_addr[0][0]), "v"(sb + a_addr[0][1]), "v"(sb + b_addr[0][0]), "v"(sb + b_addr[0][1])
                : "memory");
            a0[0] = __builtin_bit_cast(bf16x8, (u32x4){r[0][0], r[0][1], r[1][0], r[1][1]});
            a1[0] = __builtin_bit_cast(bf16x8, (u32x4){r[2][0], r[2][1], r[3][0], r[3][1]});
            b0[0] = __builtin_bit_cast(bf16x8, (u32x4){r[4][0], r[4][1], r[5][0], r[5][1]});
            b1[0] = __builtin_bit_cast(bf16x8, (u32x4){r[6][0], r[6][1], r[7][0], r[7][1]});
        }
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            }
        stage = stage + 1 == S ? 0 : stage + 1; pre = pre + 1 == S ? 0 : pre + 1;
    }
    float* C = p.C + (long long)tap * p.Ci * p.Cj + (long long)ys * p.split_stride;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j0 + wn * (BM / 2) + j * 32 + (lane & 31);
                if (p.plain_store) C[(long long)row * p.Cj + col] = acc[i][j][r];
                else unsafeAtomicAdd(C + (long long)row * p.Cj + col, acc[i][j][r]);
            }
}

// All nine taps of a 3 x 3 layer in ONE block (64 x 64 channel tile): with one tap per block every tap streams both operands again -- for the 64- and
// 128-channel layers, whose output tiles are tiny and whose row count is huge, that is 9 x (1.08 + 1.08 GB) for conv1_2 at 4 x 2048x1024: the kernel ran
// at the fabric's rate (3 ms).  Here a K-tile of 32 rows brings the 32 rows of dYp once and, per filter row ty, the 34 rows of Xp that its three taps
// read (40 are loaded: whole LDS-DMA instructions of 8 rows): row j of group ty is Xp row q0 - 1 + (ty - 1) Wp + j, and tap (ty, tx) of K-tile row r reads
// group row r + tx.  Nine accumulators (144 VGPRs) per wave, 18 MFMAs per K-tile and wave; 57 KB of LDS in three stages, two blocks per CU.
// NJ = 2 (layers with Cj % 128 == 0): the block owns 64 x 128 channels -- eight waves, the upper four on the second 64 columns of dY -- so the A rows (three
// groups of 40, 79 % of a K-tile's bytes) are fetched once for twice the products: 23.4 KB per K-tile instead of 2 x 19.4.  The lab that switches parts of the K
// loop off had shown the 64 x 64 form bound by its LDS-DMA stream on the 128- / 256-channel layers (conv2_2: 0.63 ms without MFMAs, 0.65 complete, 0.42 without
// DMA).  One block of eight waves per CU (the nine accumulators need 2 waves per SIMD either way), four stages.
template <int NJ>
__global__ __launch_bounds__(256 * NJ, NJ == 1 ? 2 : 1) void wgrad_bf16_taps9_kernel(const Bf16WgradArgs p)
{
    constexpr int S = NJ == 1 ? 3 : 4, ROWB = 128, AROWS = 40, AGRP = AROWS * ROWB, BIMG = 32 * ROWB, STAGE = 3 * AGRP + NJ * BIMG;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3, jh = wave >> 2;                         // jh: which 64-column half of the block's dY tile this wave multiplies
    const int wm = w4 >> 1, wn = w4 & 1;
    const int ntj = p.Cj / (64 * NJ), ntiles = (p.Ci / 64) * ntj;
    const int tile = blockIdx.x % ntiles, ys = blockIdx.x / ntiles;
    const int i0 = (tile / ntj) * 64, j0 = (tile % ntj) * (64 * NJ);
    const long long t0 = (long long)ys * p.chunk;
    const long long t1 = t0 + p.chunk < p.R ? t0 + p.chunk : p.R;
    const int nkt = (int)((t1 - t0 + 31) / 32);
    // waves 0 .. 2: the A rows of filter row ty = wave (five instructions of 8 rows); wave 3 (and 7): the B rows of column half 0 (1), four instructions; waves 4 .. 6: none
    const bool is_a = wave < 3, is_b = w4 == 3;
    const int ld = is_a ? p.Ci : p.Cj;
    const long long ps = is_a ? p.a_ps : p.b_ps;                   // channel-chunk planes [C / 32][rows][32] (0: [rows][C])
    const long long rstep = ps ? 32 : ld;
    const int jb = j0 + jh * 64;
    const unsigned short* mine = is_a ? p.A + (ps ? (long long)(i0 >> 5) * ps + (t0 - 1 + (long long)(wave - 1) * p.Wp) * 32 : (t0 - 1 + (long long)(wave - 1) * p.Wp) * p.Ci + i0)
                                      : p.B + (ps ? (long long)(jb >> 5) * ps + t0 * 32 : t0 * p.Cj + jb);
    unsigned voff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int row = i * 8 + lane / 8, slot = lane % 8, c = slot ^ (2 * (row & 3));
        voff[i] = ps ? (unsigned)(((long long)(c >> 2) * ps + row * 32 + (c & 3) * 8) * 2) : (unsigned)(((long long)row * ld + c * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int kt, int stage) {
        const unsigned short* g = mine + (long long)kt * 32 * rstep;
        if (!is_a && !is_b) return;
        const unsigned dst = lds0 + stage * STAGE + (is_a ? wave * AGRP : 3 * AGRP + jh * BIMG);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16w(g, voff[i], dst + i * 1024);
        if (is_a) glds16w(g, voff[4], dst + 4 * 1024);
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // lane l of an MFMA operand: column (l & 31) of its 32-wide tile, k-half g = l >> 5; inside its 16-lane group lane i supplies the address of
    // row 8g + 4r + i / 4 (+ tx for the A operand of tap column tx), columns 4 (i % 4) .. +3 of the group's 16, and receives four k of column i
    unsigned a_addr[3][2], b_addr[2];
    {
        const int q = lane >> 4, i = lane & 15, g = q >> 1;
        const int ca = wm * 32 + 16 * (q & 1) + 4 * (i & 3), cb = wn * 32 + 16 * (q & 1) + 4 * (i & 3);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 8 * g + 4 * r + (i >> 2);
            b_addr[r] = (unsigned)(3 * AGRP + jh * BIMG + row * ROWB + (((cb >> 3) ^ (2 * (row & 3))) * 16) + (cb & 7) * 2);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int j = row + tx;
                a_addr[tx][r] = (unsigned)(j * ROWB + (((ca >> 3) ^ (2 * (j & 3))) * 16) + (ca & 7) * 2);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < S - 1; ++t) if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed when at most min(S - 2, tiles after it) newer tiles of this wave's instructions are outstanding
        const int newer = nkt - 1 - kt < S - 2 ? nkt - 1 - kt : S - 2;
        if (is_a) { if (newer >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else if (newer == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else if (is_b) { if (newer >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        const unsigned sb = lds0 + stage * STAGE;
        bf16x8 b[2];
        {
            u32x2 r[4];
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %4 offset:2048\n\tds_read_b64_tr_b16 %3, %5 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(sb + b_addr[0]), "v"(sb + b_addr[1]) : "memory");
            b[0] = __builtin_bit_cast(bf16x8, (u32x4){r[0][0], r[0][1], r[1][0], r[1][1]});
            b[1] = __builtin_bit_cast(bf16x8, (u32x4){r[2][0], r[2][1], r[3][0], r[3][1]});
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const unsigned sa = sb + ty * AGRP;
            u32x2 r[12];          // [tx][ks][r]
            asm volatile(
                "ds_read_b64_tr_b16 %0, %12\n\tds_read_b64_tr_b16 %1, %13\n\tds_read_b64_tr_b16 %2, %12 offset:2048\n\tds_read_b64_tr_b16 %3, %13 offset:2048\n\t"
                "ds_read_b64_tr_b16 %4, %14\n\tds_read_b64_tr_b16 %5, %15\n\tds_read_b64_tr_b16 %6, %14 offset:2048\n\tds_read_b64_tr_b16 %7, %15 offset:2048\n\t"
                "ds_read_b64_tr_b16 %8, %16\n\tds_read_b64_tr_b16 %9, %17\n\tds_read_b64_tr_b16 %10, %16 offset:2048\n\tds_read_b64_tr_b16 %11, %17 offset:2048\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11])
                : "v"(sa + a_addr[0][0]), "v"(sa + a_addr[0][1]), "v"(sa + a_addr[1][0]), "v"(sa + a_addr[1][1]), "v"(sa + a_addr[2][0]), "v"(sa + a_addr[2][1])
                : "memory");
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, (u32x4){r[4 * tx + 2 * ks][0], r[4 * tx + 2 * ks][1], r[4 * tx + 2 * ks + 1][0], r[4 * tx + 2 * ks + 1][1]});
                    acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ks], acc[ty * 3 + tx], 0, 0, 0);
                }
        }
        stage = stage + 1 == S ? 0 : stage + 1; pre = pre + 1 == S ? 0 : pre + 1;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* C = p.C + (long long)t * p.Ci * p.Cj + (long long)ys * p.split_stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j0 + jh * 64 + wn * 32 + (lane & 31);
            if (p.plain_store) C[(long long)row * p.Cj + col] = acc[t][r];
            else unsafeAtomicAdd(C + (long long)row * p.Cj + col, acc[t][r]);
        }
    }
}

// dW (K*K x Ci x Cj fp32, TensorFlow's HWIO) is ASSIGNED.  A / B point at padded pixel 0 of their buffers (guard rows in front of it).
bool launch_wgrad_bf16(const Bf16WgradArgs& a0, hipStream_t s)
{
    Bf16WgradArgs a = a0;
    if (a.Ci % 64 || a.Cj % 64 || (a.K & 1) == 0 || a.R < 1) return false;
    // per-lane byte offsets are 32-bit and span the channel-chunk planes one 16-row LDS-DMA instruction touches: TWO planes (128-byte LDS rows: the nine-tap kernel and
    // the 64-channel tile) or FOUR (256-byte rows: the 128-channel tile).  A plane of 64 x 1024x512 is 2.2 GB: two are addressable, four are not.
    auto spans = [&](int planes) { return (double)a.a_ps * 2.0 * (planes - 1) + 65536.0 < 4294967296.0 && (double)a.b_ps * 2.0 * (planes - 1) + 65536.0 < 4294967296.0; };
    if (!spans(2)) return false;      // (plane stride 0 = [rows][C]: offsets stay inside a K-tile's 40 rows)
    // 3 x 3 layers: all nine taps per block (the operands are streamed once instead of nine times).  Measured at 4 x 2048x1024 against one tap per block
    // (profiles/r05_wgrad_taps9_ab.txt): conv1_2 3.08 -> 0.76 ms, conv2_2 1.43 -> 0.67, conv3_2 1.51 -> 0.68 (409 -> 904 TFLOP/s), conv4_2 1.13 -> 0.57
    // (546 -> 1092), conv5_x 0.24 -> 0.20.  (The A/B ran on an environment switch that is gone again: the library reads no environment variable.)
    const bool taps9 = a.K == 3;
    // the nine-tap kernel's 64 x 128 form where it measured faster (same-box A/B, profiles/r05_bf16_conv_tile_ab.txt): the 128- / 256-channel layers with at
    // least two tiles -- conv2_2 0.63 -> 0.55 ms, conv3_2 0.55 -> 0.52, conv3_1 0.33 -> 0.305; conv4_x the same either way, conv5_x 0.168 -> 0.18 (short K
    // loops: one block per CU hides less), conv2_1 (a single tile) 0.33 -> 0.46
    const int nj = (taps9 && a.Cj % 128 == 0 && a.Cj <= 256 && (a.Ci / 64) * (a.Cj / 128) >= 2) ? 2 : 1;
    const int bm = taps9 ? 64 : ((a.Ci % 128 == 0 && a.Cj % 128 == 0 && spans(4)) ? 128 : 64);
    const int taps = a.K * a.K;
    const long long tiles = (long long)(a.Ci / bm) * (a.Cj / (bm * nj)) * (taps9 ? 1 : taps);
    const long long slots = 256LL * (taps9 ? (nj == 2 ? 1 : 2) : (bm == 128 ? 2 : 4));
    // splits: ONE round of resident blocks for the nine-tap kernel (every block ends with 9 x 16 float atomics per lane: with its K loop switched on and the
    // epilogue off conv5_x took 0.13 instead of 0.20 ms, conv3 / conv4 8-13 % less), half a round where all blocks add into one 64 x 64 x 9 tile (conv1_2);
    // same-box A/B (profiles/r05_bf16_conv_tile_ab.txt): two rounds / one / half = conv5_2 0.202 / 0.170 / 0.200, conv4_2 0.527 / 0.505 / 0.657, conv1_2 0.77 / 0.76 / 0.70
    long long want = taps9 ? (tiles == 1 ? slots / 2 : (slots + tiles - 1) / tiles) : (2 * slots + tiles - 1) / tiles;
    const long long maxsplit = (a.R + 1023) / 1024;              // at least 32 K-tiles per block
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    long long chunk = ((a.R + want - 1) / want + 31) / 32 * 32;
    const int nsplit = (int)((a.R + chunk - 1) / chunk);
    a.chunk = chunk;
    const long long slab = (long long)taps * a.Ci * a.Cj;
    float* out = a.C;
    a.split_stride = 0; a.plain_store = nsplit == 1;
    if (nsplit > 1) {
        if (t_deterministic) {
            float* ws = det_scratch(s, (size_t)(nsplit * slab));
            if (!ws) { defer_error(FCN8S_ERR_OOM, "deterministic mode: the weight gradient's slab scratch (%lld floats) cannot be allocated", (long long)nsplit * slab); return true; }
            a.C = ws; a.split_stride = slab; a.plain_store = 1;
        } else hipMemsetAsync(out, 0, (size_t)slab * sizeof(float), s);
    }
    if (taps9) {
        g_last_kernel = "wgrad_bf16_taps9_kernel";
        if (nj == 2) hipLaunchKernelGGL(wgrad_bf16_taps9_kernel<2>, dim3((unsigned)(tiles * nsplit)), dim3(512), 0, s, a);
        else hipLaunchKernelGGL(wgrad_bf16_taps9_kernel<1>, dim3((unsigned)(tiles * nsplit)), dim3(256), 0, s, a);
        if (a.split_stride) launch_det_reduce(out, a.C, 1, (int)slab, (int)slab, slab, nsplit, false, s);
        return true;
    }
    dim3 grid((unsigned)((tiles / taps) * nsplit), 1, (unsigned)taps);
    if (bm == 128) { g_last_kernel = "wgrad_bf16_kernel<128>"; hipLaunchKernelGGL(wgrad_bf16_kernel<128>, grid, dim3(256), 0, s, a); }
    else { g_last_kernel = "wgrad_bf16_kernel<64>"; hipLaunchKernelGGL(wgrad_bf16_kernel<64>, grid, dim3(256), 0, s, a); }
    if (a.split_stride) launch_det_reduce(out, a.C, 1, (int)slab, (int)slab, slab, nsplit, false, s);
    return true;
}

bool launch_conv_bf16(const Bf16ConvArgs& a0, hipStream_t s)
{
    if (a0.Cin % BFK || a0.Cout % 128 || (a0.K & 1) == 0) return false;
    Bf16ConvArgs a = a0;
    a.M = (long long)a.N * a.H * a.W;
    const double ntm = (double)((a.M + 127) / 128), ntn = a.Cout / 128;
    // HBM / fabric bytes of either tile order with ~96 resident blocks behind each of the 8 L2s
    const double abytes = 4.0 * a.M * a.Cin, bbytes = 2.0 * a.K * a.K * a.Cin * a.Cout, R = 96.0;
    const double n_fast = abytes + bbytes * ntm / (R / ntn > 1.0 ? R / ntn : 1.0);
    const double m_fast = bbytes + abytes * ntn / (R / ntm > 1.0 ? R / ntm : 1.0);
    a.m_fastest = m_fast < n_fast;
    g_last_kernel = a.xh ? "conv_bf16_kernel<128, 128, true>" : "conv_bf16_kernel<128, 128, false>";
    dim3 grid((unsigned)(ntm * ntn));
    if (a.xh) hipLaunchKernelGGL((conv_bf16_kernel<128, 128, true>), grid, dim3(256), 0, s, a);
    else      hipLaunchKernelGGL((conv_bf16_kernel<128, 128, false>), grid, dim3(256), 0, s, a);
    return true;
}

}  // namespace fcn8s
