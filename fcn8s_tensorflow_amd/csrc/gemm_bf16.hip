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
// 256 x 256 tile, 8 waves, LDS-DMA, two staggered wave groups (round 3; tools/bf16_lab.hip has the stand-alone version and its diagnosis)
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
constexpr int G_BM = 256, G_BN = 256, G_BK = 32, G_S = 5, G_ROWB = G_BK * 2, G_ABYTES = G_BM * G_ROWB, G_STAGE = (G_BM + G_BN) * G_ROWB;

static __device__ __forceinline__ void glds16b(const void* sbase, unsigned voff, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory", "m0");
}
template <int N> static __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}

// w[K][Cout] fp32 (HWIO flattened) -> wt[Cout][K] bf16
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
        *reinterpret_cast<bf16x8*>(wt + (long long)(c0 + c) * K + k0 + kq) = o;
    }
}
void launch_w_to_bf16_t(const float* w, unsigned short* wt, int K, int Cout, hipStream_t s)      // K % 32 == 0
{
    dim3 grid((unsigned)((Cout + 63) / 64), (unsigned)(K / 32));
    hipLaunchKernelGGL(w_to_bf16_t_kernel, grid, dim3(256), 0, s, w, wt, K, Cout);
}

// x [N][H][W][C] fp32 -> xp [N][H + 2 pad][W + 2 pad][C] bf16 (RNE), zero border; 8 channels per thread
__global__ __launch_bounds__(256) void f32_to_bf16_padded_kernel(const float4* __restrict__ x, bf16x8* __restrict__ xp, int N, int H, int W, int C8, int pad)
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
        xp[i] = o;
    }
}
void launch_f32_to_bf16_padded(const float* x, unsigned short* xp, int N, int H, int W, int C, int pad, hipStream_t s)     // C % 8 == 0
{
    const long long total = (long long)N * (H + 2 * pad) * (W + 2 * pad) * (C / 8);
    long long b = (total + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1;
    g_last_kernel = "f32_to_bf16_padded_kernel";
    hipLaunchKernelGGL(f32_to_bf16_padded_kernel, dim3((unsigned)b), dim3(256), 0, s, (const float4*)x, (bf16x8*)xp, N, H, W, C / 8, pad);
}

__global__ __launch_bounds__(512, 1) void conv_bf16_256_kernel(const Bf16Conv256Args p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[G_S * G_STAGE];       // 160 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const unsigned ntm = (unsigned)(p.M / G_BM), ntn = (unsigned)(p.Cout / G_BN);
    const unsigned lid = xcd_run(blockIdx.x, gridDim.x);
    const unsigned tmi = p.m_fastest ? lid % ntm : lid / ntn, tni = p.m_fastest ? lid / ntm : lid % ntn;
    const long long m0 = (long long)tmi * G_BM; const int n0 = (int)tni * G_BN;
    const int HW = p.H * p.W, Hp = p.H + p.K - 1, Wp = p.W + p.K - 1, Ktot = p.K * p.K * p.Cin;

    // LDS-DMA: wave w fills 16-row chunks 2w, 2w + 1 of the A image and of the B image of a stage
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + lane / 4, pc = lane % 4;
        const int lc = pc ^ ((row >> 2) & 3);                                  // logical 16-byte chunk stored at physical chunk pc
        const long long m = m0 + row;
        const int n = (int)(m / HW), r = (int)(m - (long long)n * HW), y = r / p.W, x = r - y * p.W;
        const long long pp = ((long long)n * Hp + y) * Wp + x;                // top-left pixel of the row's tap window in the padded copy
        a_voff[i] = (unsigned)((pp * p.Cin + lc * 8) * 2);
        b_voff[i] = (unsigned)(((long long)row * Ktot + lc * 8) * 2);
    }
    const unsigned short* b_base = p.wt + (long long)n0 * Ktot;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // K-tiles are issued strictly in order, one call per tile: the tap position advances incrementally
    int i_kt = 0, i_ci = 0, i_tx = 0, i_ty = 0;
    auto issue = [&]() {
        const unsigned st = lds0 + (unsigned)((i_kt % G_S) * G_STAGE);
        const unsigned short* ga = p.xp + ((long long)i_ty * Wp + i_tx) * p.Cin + i_ci;
        const unsigned short* gb = b_base + (long long)i_kt * G_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16b(ga, a_voff[i], st + (wave * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16b(gb, b_voff[i], st + G_ABYTES + (wave * 2 + i) * 1024);
        ++i_kt; i_ci += G_BK;
        if (i_ci == p.Cin) { i_ci = 0; if (++i_tx == p.K) { i_tx = 0; ++i_ty; } }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_row[4], b_row[2];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) a_row[tm] = grp * 128 + tm * 32 + (lane & 31);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) b_row[tn] = wn * 64 + tn * 32 + (lane & 31);
    bf16x8 af[2][4], bfr[2][2];
    auto load_frags = [&](int kt) {
        const unsigned char* st = smem + (kt % G_S) * G_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const int r = a_row[tm], pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                af[ks][tm] = *reinterpret_cast<const bf16x8*>(st + r * G_ROWB + pc * 16);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
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
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][tm], bfr[ks][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    const int nkt = Ktot / G_BK;
#pragma unroll
    for (int t = 0; t < G_S - 1; ++t) if (t < nkt) issue();
    if (nkt > 3) wait_vm<12>(); else if (nkt > 2) wait_vm<8>(); else if (nkt > 1) wait_vm<4>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    // tick t: group g runs step t - g; even steps read the fragments of K-tile step / 2, odd steps multiply them.  Even ticks 2 j issue the
    // LDS-DMA of K-tile j + 4 (its stage was last read in tick 2 j - 1); odd ticks 2 kt + 1 wait for K-tile kt + 1.  Each group runs its own
    // straight-line loop (one loop with per-tick branches made hipcc copy the accumulators around: 10x slower), and sched_barrier keeps
    // hipcc from hoisting a group's MFMAs above the barrier that opens its MFMA phase.
    auto tick_end = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    auto wait_tile = [&](int kt) {      // this wave's pieces of K-tile kt have landed; up to three newer tiles may still be in flight
        if (kt + 3 < nkt) wait_vm<12>(); else if (kt + 2 < nkt) wait_vm<8>(); else if (kt + 1 < nkt) wait_vm<4>(); else wait_vm<0>();
    };
    if (grp == 0) {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + G_S - 1 < nkt) issue();
            load_frags(kt);
            tick_end();
            mfma_phase();
            wait_tile(kt + 1);
            tick_end();
        }
        tick_end();
    } else {
        if (G_S - 1 < nkt) issue();
        tick_end();
        for (int kt = 0; kt < nkt; ++kt) {
            load_frags(kt);
            wait_tile(kt + 1);
            tick_end();
            if (kt + G_S < nkt) issue();
            mfma_phase();
            tick_end();
        }
    }

    // epilogue (fp32): bias, ReLU, dropout keyed by the element offset -- the same Philox stream as the fp32 path
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn * 64 + tn * 32 + (lane & 31);
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + grp * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long long off = m * p.Cout + col;
                float v = acc[tm][tn][r] + bv;
                if (p.relu) v = v > 0.f ? v : 0.f;
                if (p.dropout) v = philox_uniform((unsigned long long)off, p.seed, p.stream_id) < p.keep_prob ? v / p.keep_prob : 0.f;
                p.y[off] = v;
            }
    }
}

bool conv_bf16_256_ok(long long M, int Cin, int Cout, int mode)
{
    if (mode == 0 || M % G_BM || Cout % G_BN || Cin % G_BK || Cin % 8) return false;
    return mode >= 2 || (M / G_BM) * (Cout / G_BN) >= 128;         // fewer tiles than half the CUs: the 128 x 128 kernel fills the chip better
}

bool launch_conv_bf16_256(const Bf16Conv256Args& a0, hipStream_t s)
{
    Bf16Conv256Args a = a0;
    a.M = (long long)a.N * a.H * a.W;
    if (!conv_bf16_256_ok(a.M, a.Cin, a.Cout, 2)) return false;
    const double abytes = 2.0 * a.M * a.K * a.K * a.Cin, bbytes = 2.0 * a.K * a.K * a.Cin * a.Cout;
    a.m_fastest = bbytes > abytes;             // the larger operand's panel stays put behind one XCD's L2 while the other one streams
    g_last_kernel = "conv_bf16_256_kernel";
    hipLaunchKernelGGL(conv_bf16_256_kernel, dim3((unsigned)((a.M / G_BM) * (a.Cout / G_BN))), dim3(512), 0, s, a);
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
