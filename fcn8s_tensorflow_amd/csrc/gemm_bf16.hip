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
