// Lab (round 4): the two-piece ("f32x2") GEMMs with their operands PRE-SPLIT into bf16 planes (hi = bf16(x), lo = bf16(x - hi))
// by whoever produced them, so that the K loop holds no VALU work: LDS-DMA fills, 16-byte fragment reads, three v_mfma_f32_32x32x16_bf16 per
// fragment pair (lo*hi + hi*lo + hi*hi), fp32 accumulate.  Two kernels, the two shapes the Winograd-domain step needs:
//   gemm_planes   Y[z][m][n] = sum_k A[z][m][k] B[z][n][k]       both operands k-contiguous  (forward V U, adjoint data gradient dM U^T)
//   wgrad_planes  C[z][i][j] = sum_t A[z][t][i] B[z][t][j]       both operands k-STRIDED     (weight gradient V^T dM): fragments through
//                                                                ds_read_b64_tr_b16 (hardware 4x4 transpose of 16-bit elements)
// Checked against a float64 evaluation of the same three piece products, then timed on conv3_2 / conv4_2 / fc6-like batched shapes next
// to the step's current in-loop-split kernels (profiles/r03_layer_bench_bf16_fwd_x2.txt: 220-290 "TFLOP/s").
//   hipcc -O3 --offload-arch=gfx950 tools/planes_lab.hip -o scratch/planes_lab && scratch/planes_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

static __device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory", "m0");
}
template <int N> static __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
static __device__ __forceinline__ f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

struct GemmArgs { const u16 *ahi, *alo, *bhi, *blo; float* y; long long M; int N, K; long long a_bs, b_bs, y_bs; };   // batch strides in elements

// ---- both operands k-contiguous: tile 128 x 128, 4 waves (2 x 2) of 64 x 64, BK = 16 bf16 = 32-byte rows, 3 stages of 16 KB ------------------
// LDS image of one plane tile: [128 rows][2 chunks of 16 B]; chunk slot = h ^ ((row >> 3) & 1), so that the 16 rows a 16-lane group reads
// with one ds_read_b128 spread over all 64 banks.  One LDS-DMA instruction = 32 rows.
template <int S>
__global__ __launch_bounds__(256, 3) void gemm_planes(const GemmArgs p)
{
    constexpr int BM = 128, BN = 128, PT = 128 * 32;            // bytes per plane tile
    constexpr int STAGE = 4 * PT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = p.N / BN;
    const long long m0 = (long long)(blockIdx.x / ntn) * BM; const int n0 = (blockIdx.x % ntn) * BN;
    const int z = blockIdx.z;
    const u16* planes[4] = {p.ahi + z * p.a_bs + m0 * p.K, p.alo + z * p.a_bs + m0 * p.K, p.bhi + z * p.b_bs + (long long)n0 * p.K, p.blo + z * p.b_bs + (long long)n0 * p.K};
    // wave w fills plane tile w: 4 instructions of 32 rows each
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + (lane >> 1), slot = lane & 1, h = slot ^ ((row >> 3) & 1);
        long long r = row; if (wave < 2 && m0 + r >= p.M) r = p.M - 1 - m0;       // rows past the end read the last row (never stored)
        voff[i] = (unsigned)((r * p.K + h * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const u16* mine = planes[wave];
    auto issue = [&](int kt, int stage) {
        const u16* g = mine + kt * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(g, voff[i], lds0 + stage * STAGE + wave * PT + i * 1024);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int aoff[2], boff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + (lane & 31), rb = wn * 64 + t * 32 + (lane & 31), h = lane >> 5;
        aoff[t] = ra * 32 + ((h ^ ((ra >> 3) & 1)) * 16);
        boff[t] = rb * 32 + ((h ^ ((rb >> 3) & 1)) * 16);
    }
    const int nkt = p.K / 16;
#pragma unroll
    for (int t = 0; t < S - 1; ++t) if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + S - 2 < nkt) wait_vmcnt<(S - 2) * 4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        const unsigned char* st = smem + stage * STAGE;
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = *reinterpret_cast<const bf16x8*>(st + aoff[t]); al[t] = *reinterpret_cast<const bf16x8*>(st + PT + aoff[t]);
            bh[t] = *reinterpret_cast<const bf16x8*>(st + 2 * PT + boff[t]); bl[t] = *reinterpret_cast<const bf16x8*>(st + 3 * PT + boff[t]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
        stage = stage + 1 == S ? 0 : stage + 1; pre = pre + 1 == S ? 0 : pre + 1;
    }
    float* Y = p.y + z * p.y_bs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) Y[m * p.N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}

// ---- the same GEMM on an INTERLEAVED layout: per row and group of 16 k, 16 hi values then 16 lo values (64 bytes) -- a row piece of one
// K = 16 step is one contiguous 64-byte run as in the fp32 kernel (32-byte pieces of separate planes fetch badly), the LDS image is that
// kernel's A image ([rows][4 chunks], chunk slot = c ^ ((row >> 2) & 3)): chunks 0, 1 = hi (k 0-7, 8-15), chunks 2, 3 = lo.
struct GemmIArgs { const u16 *a, *b; float* y; long long M; int N, K; long long a_bs, b_bs, y_bs; };
template <int S>
__global__ __launch_bounds__(256, 3) void gemm_inter(const GemmIArgs p)
{
    constexpr int BM = 128, BN = 128, PT = 128 * 64, STAGE = 2 * PT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = p.N / BN;
    const long long m0 = (long long)(blockIdx.x / ntn) * BM; const int n0 = (blockIdx.x % ntn) * BN;
    const int z = blockIdx.z;
    const long long ld = 2LL * p.K;                                  // u16 per row
    const u16* src = wave < 2 ? p.a + z * p.a_bs + (m0 + (wave & 1) * 64) * ld : p.b + z * p.b_bs + ((long long)n0 + (wave & 1) * 64) * ld;
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + (lane >> 2), pc = lane & 3, rr = (wave & 1) * 64 + row, c = pc ^ ((rr >> 2) & 3);
        voff[i] = (unsigned)(((long long)row * ld + c * 8) * 2);          // (the lab's row counts are padded to whole tiles)
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int kt, int stage) {
        const u16* g = src + kt * 32;
        const unsigned base = lds0 + stage * STAGE + (wave >> 1) * PT + (wave & 1) * 64 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(g, voff[i], base + i * 1024);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int ahi_o[2], alo_o[2], bhi_o[2], blo_o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + (lane & 31), rb = wn * 64 + t * 32 + (lane & 31), h = lane >> 5;
        ahi_o[t] = ra * 64 + ((h ^ ((ra >> 2) & 3)) * 16); alo_o[t] = ra * 64 + (((2 + h) ^ ((ra >> 2) & 3)) * 16);
        bhi_o[t] = PT + rb * 64 + ((h ^ ((rb >> 2) & 3)) * 16); blo_o[t] = PT + rb * 64 + (((2 + h) ^ ((rb >> 2) & 3)) * 16);
    }
    const int nkt = p.K / 16;
#pragma unroll
    for (int t = 0; t < S - 1; ++t) if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + S - 2 < nkt) wait_vmcnt<(S - 2) * 4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        const unsigned char* st = smem + stage * STAGE;
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = *reinterpret_cast<const bf16x8*>(st + ahi_o[t]); al[t] = *reinterpret_cast<const bf16x8*>(st + alo_o[t]);
            bh[t] = *reinterpret_cast<const bf16x8*>(st + bhi_o[t]); bl[t] = *reinterpret_cast<const bf16x8*>(st + blo_o[t]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
        stage = stage + 1 == S ? 0 : stage + 1; pre = pre + 1 == S ? 0 : pre + 1;
    }
    float* Y = p.y + z * p.y_bs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) Y[m * p.N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}

// ---- both operands k-strided ([t][channel] planes): tile 128 x 128 of C, BK = 16 rows of t, fragments through ds_read_b64_tr_b16 -------------
// LDS image of one plane tile: [16 t-rows][16 chunks of 16 B] (256-byte rows); chunk slot = c ^ (2 * (row & 3)): the four rows a 16-lane
// group transposes start in different banks.  One LDS-DMA instruction = 4 rows.
static __device__ __forceinline__ void tr_read(unsigned lds_addr, unsigned& lo, unsigned& hi)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    lo = v[0]; hi = v[1];
}
struct WgradArgs { const u16 *ahi, *alo, *bhi, *blo; float* c; long long T; int Ci, Cj; long long a_bs, b_bs, c_bs; int chunk, nsplit; };
template <int S>
__global__ __launch_bounds__(256, 3) void wgrad_planes(const WgradArgs p)
{
    constexpr int BM = 128, BN = 128, PT = 16 * 256, STAGE = 4 * PT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntj = p.Cj / BN, ntiles = (p.Ci / BM) * ntj;
    const int tile = blockIdx.x % ntiles, ys = blockIdx.x / ntiles, z = blockIdx.z;
    const int i0 = (tile / ntj) * BM, j0 = (tile % ntj) * BN;
    const long long t0 = (long long)ys * p.chunk, t1 = t0 + p.chunk < p.T ? t0 + p.chunk : p.T;
    const int nkt = (int)((t1 - t0) / 16);                        // (the lab's T is a multiple of 16 per chunk)
    const u16* planes[4] = {p.ahi + z * p.a_bs + t0 * p.Ci + i0, p.alo + z * p.a_bs + t0 * p.Ci + i0, p.bhi + z * p.b_bs + t0 * p.Cj + j0, p.blo + z * p.b_bs + t0 * p.Cj + j0};
    const int ld = wave < 2 ? p.Ci : p.Cj;
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 4 + (lane >> 4), slot = lane & 15, c = slot ^ (2 * (row & 3));
        voff[i] = (unsigned)(((long long)row * ld + c * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const u16* mine = planes[wave];
    auto issue = [&](int kt, int stage) {
        const u16* g = mine + (long long)kt * 16 * ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(g, voff[i], lds0 + stage * STAGE + wave * PT + i * 1024);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // lane l of an MFMA operand: column (l & 31) of the 32-wide tile, k-half g = l >> 5 (rows 8g .. 8g+7 of the K-tile).  Its 16-lane group
    // q = l >> 4 covers columns 16 (q & 1) .. +15; inside the group lane i = l & 15 SUPPLIES the address of row 8g + 4r + i / 4, columns
    // 4 (i % 4) .. +3 of the group's 16 (8 bytes) and RECEIVES rows 8g + 4r .. +3 of column i.
    unsigned a_addr[2][2], b_addr[2][2];          // [tile][r]
    {
        const int q = lane >> 4, i = lane & 15, g = q >> 1;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = 8 * g + 4 * r + (i >> 2);
                const int ca = wm * 64 + t * 32 + 16 * (q & 1) + 4 * (i & 3), cb = wn * 64 + t * 32 + 16 * (q & 1) + 4 * (i & 3);     // first column (elements)
                a_addr[t][r] = (unsigned)(row * 256 + (((ca >> 3) ^ (2 * (row & 3))) * 16) + (ca & 7) * 2);
                b_addr[t][r] = (unsigned)(row * 256 + (((cb >> 3) ^ (2 * (row & 3))) * 16) + (cb & 7) * 2);
            }
    }
#pragma unroll
    for (int t = 0; t < S - 1; ++t) if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + S - 2 < nkt) wait_vmcnt<(S - 2) * 4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        const unsigned sb = lds0 + stage * STAGE;
        // all sixteen transposing reads and their wait in ONE asm statement: the compiler treats an asm's outputs as ready when the statement
        // ends, so a separate s_waitcnt behind it would come after the moves that already read the (not yet written) registers
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x2 r[16];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %17\n\tds_read_b64_tr_b16 %2, %18\n\tds_read_b64_tr_b16 %3, %19\n\t"
            "ds_read_b64_tr_b16 %4, %16 offset:4096\n\tds_read_b64_tr_b16 %5, %17 offset:4096\n\tds_read_b64_tr_b16 %6, %18 offset:4096\n\tds_read_b64_tr_b16 %7, %19 offset:4096\n\t"
            "ds_read_b64_tr_b16 %8, %20 offset:8192\n\tds_read_b64_tr_b16 %9, %21 offset:8192\n\tds_read_b64_tr_b16 %10, %22 offset:8192\n\tds_read_b64_tr_b16 %11, %23 offset:8192\n\t"
            "ds_read_b64_tr_b16 %12, %20 offset:12288\n\tds_read_b64_tr_b16 %13, %21 offset:12288\n\tds_read_b64_tr_b16 %14, %22 offset:12288\n\tds_read_b64_tr_b16 %15, %23 offset:12288\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
              "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
            : "v"(sb + a_addr[0][0]), "v"(sb + a_addr[0][1]), "v"(sb + a_addr[1][0]), "v"(sb + a_addr[1][1]),
              "v"(sb + b_addr[0][0]), "v"(sb + b_addr[0][1]), "v"(sb + b_addr[1][0]), "v"(sb + b_addr[1][1])
            : "memory");
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = __builtin_bit_cast(bf16x8, (u32x4){r[2 * t][0], r[2 * t][1], r[2 * t + 1][0], r[2 * t + 1][1]});
            al[t] = __builtin_bit_cast(bf16x8, (u32x4){r[4 + 2 * t][0], r[4 + 2 * t][1], r[4 + 2 * t + 1][0], r[4 + 2 * t + 1][1]});
            bh[t] = __builtin_bit_cast(bf16x8, (u32x4){r[8 + 2 * t][0], r[8 + 2 * t][1], r[8 + 2 * t + 1][0], r[8 + 2 * t + 1][1]});
            bl[t] = __builtin_bit_cast(bf16x8, (u32x4){r[12 + 2 * t][0], r[12 + 2 * t][1], r[12 + 2 * t + 1][0], r[12 + 2 * t + 1][1]});
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
        stage = stage + 1 == S ? 0 : stage + 1; pre = pre + 1 == S ? 0 : pre + 1;
    }
    float* C = p.c + z * p.c_bs;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j0 + wn * 64 + j * 32 + (lane & 31);
                if (p.nsplit == 1) C[(long long)row * p.Cj + col] = acc[i][j][r];
                else unsafeAtomicAdd(C + (long long)row * p.Cj + col, acc[i][j][r]);
            }
}

__global__ void fill_bf16(u16* p, long long n, unsigned seed, int small)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // sign + exponent around 2^-2 .. 2^0 (small: 2^-10 .. 2^-8, a "lo" plane) + 7 random mantissa bits
        const unsigned e = (small ? 117u : 125u) + (h >> 8) % 3u;
        p[i] = (u16)(((h >> 31) << 15) | (e << 7) | (h & 0x7fu));
    }
}
// ---- host side ---------------------------------------------------------------------------------------------------------------------------
static u16 f2bf(float x) { unsigned u; memcpy(&u, &x, 4); const unsigned r = u + 0x7fffu + ((u >> 16) & 1u); return (u16)(r >> 16); }      // RNE (no NaNs here)
static float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; }
static void split(const std::vector<float>& x, std::vector<u16>& hi, std::vector<u16>& lo)
{
    hi.resize(x.size()); lo.resize(x.size());
    for (size_t i = 0; i < x.size(); ++i) { hi[i] = f2bf(x[i]); lo[i] = f2bf(x[i] - bf2f(hi[i])); }
}
template <class T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc((void**)&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
template <class F> static float timeit(F f, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
static float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    // ---- correctness, small: gemm 200 x 128 x 64 (ragged M), wgrad T = 96, 128 x 128
    {
        const int M = 200, N = 128, K = 64; unsigned s = 1;
        std::vector<float> A((size_t)256 * K), B((size_t)N * K);          // (rows padded to whole 128-row tiles)
        for (auto& v : A) v = rnd(s); for (auto& v : B) v = rnd(s);
        std::vector<u16> ah, al, bh, bl; split(A, ah, al); split(B, bh, bl);
        std::vector<float> Y((size_t)M * N, 0.f); float* dy = dev(Y);
        GemmArgs g{dev(ah), dev(al), dev(bh), dev(bl), dy, M, N, K, 0, 0, 0};
        hipLaunchKernelGGL((gemm_planes<3>), dim3((unsigned)((M + 127) / 128 * (N / 128)), 1, 1), dim3(256), 0, 0, g);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst32 = 0, scale = 0;
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            double ref = 0, ref32 = 0;
            for (int k = 0; k < K; ++k) {
                const double a0 = bf2f(ah[(size_t)m * K + k]), a1 = bf2f(al[(size_t)m * K + k]), b0 = bf2f(bh[(size_t)n * K + k]), b1 = bf2f(bl[(size_t)n * K + k]);
                ref += a1 * b0 + a0 * b1 + a0 * b0; ref32 += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
            }
            worst = fmax(worst, fabs(Y[(size_t)m * N + n] - ref)); worst32 = fmax(worst32, fabs(Y[(size_t)m * N + n] - ref32)); scale = fmax(scale, fabs(ref));
        }
        printf("gemm_planes  200x128x64: max |y - three-product float64| = %.3e, |y - exact fp32-operand product| = %.3e (scale %.2f)\n", worst, worst32, scale);
        // the interleaved layout of the same operands
        std::vector<u16> ai((size_t)256 * 2 * K), bi((size_t)N * 2 * K);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { ai[(size_t)m * 2 * K + (k / 16) * 32 + k % 16] = ah[(size_t)m * K + k]; ai[(size_t)m * 2 * K + (k / 16) * 32 + 16 + k % 16] = al[(size_t)m * K + k]; }
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { bi[(size_t)n * 2 * K + (k / 16) * 32 + k % 16] = bh[(size_t)n * K + k]; bi[(size_t)n * 2 * K + (k / 16) * 32 + 16 + k % 16] = bl[(size_t)n * K + k]; }
        std::vector<float> Y2((size_t)M * N, 0.f); float* dy2 = dev(Y2);
        GemmIArgs gi{dev(ai), dev(bi), dy2, M, N, K, 0, 0, 0};
        hipLaunchKernelGGL((gemm_inter<3>), dim3((unsigned)((M + 127) / 128 * (N / 128)), 1, 1), dim3(256), 0, 0, gi);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(Y2.data(), dy2, Y2.size() * 4, hipMemcpyDeviceToHost));
        double wd = 0; for (size_t i = 0; i < Y.size(); ++i) wd = fmax(wd, fabs((double)Y2[i] - Y[i]));
        printf("gemm_inter   200x128x64: max |y - gemm_planes y| = %.3e\n", wd);
    }
    {
        const int T = 96, Ci = 128, Cj = 256; unsigned s = 7;
        std::vector<float> A((size_t)T * Ci), B((size_t)T * Cj);
        for (auto& v : A) v = rnd(s); for (auto& v : B) v = rnd(s);
        std::vector<u16> ah, al, bh, bl; split(A, ah, al); split(B, bh, bl);
        std::vector<float> Cc((size_t)Ci * Cj, 0.f); float* dc = dev(Cc);
        WgradArgs g{dev(ah), dev(al), dev(bh), dev(bl), dc, T, Ci, Cj, 0, 0, 0, 48, 2};
        hipLaunchKernelGGL((wgrad_planes<3>), dim3((unsigned)((Ci / 128) * (Cj / 128) * 2), 1, 1), dim3(256), 0, 0, g);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(Cc.data(), dc, Cc.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int i = 0; i < Ci; ++i) for (int j = 0; j < Cj; ++j) {
            double ref = 0;
            for (int t = 0; t < T; ++t) {
                const double a0 = bf2f(ah[(size_t)t * Ci + i]), a1 = bf2f(al[(size_t)t * Ci + i]), b0 = bf2f(bh[(size_t)t * Cj + j]), b1 = bf2f(bl[(size_t)t * Cj + j]);
                ref += a1 * b0 + a0 * b1 + a0 * b0;
            }
            worst = fmax(worst, fabs(Cc[(size_t)i * Cj + j] - ref)); scale = fmax(scale, fabs(ref));
        }
        printf("wgrad_planes 96 rows, 128x256: max |c - three-product float64| = %.3e (scale %.2f)\n", worst, scale);
    }
    // ---- timing: batched shapes of the step (16 x 1024x512): planes filled with a pattern
    struct Shape { const char* name; int P; long long T; int K, N; } shapes[] = {
        {"conv3_2 (64 positions, T = 15136, 256 -> 256)", 64, 15136, 256, 256}, {"conv4_2 (64 positions, T = 3872, 512 -> 512)", 64, 3872, 512, 512},
        {"conv5_2 (64 positions, T = 1056, 512 -> 512)", 64, 1056, 512, 512}, {"fc6 (49 positions, T = 2048, 2048 -> 4096)", 49, 2048, 2048, 4096}};
    for (auto& sh : shapes) {
        const long long Tp = (sh.T + 127) / 128 * 128;
        const size_t ae = (size_t)sh.P * Tp * sh.K, be = (size_t)sh.P * sh.N * sh.K, ye = (size_t)sh.P * Tp * sh.N;
        u16 *ahi, *alo, *bhi, *blo, *dmh, *dml; float *y, *c;
        CK(hipMalloc((void**)&ahi, ae * 2)); CK(hipMalloc((void**)&alo, ae * 2)); CK(hipMalloc((void**)&bhi, be * 2)); CK(hipMalloc((void**)&blo, be * 2));
        CK(hipMalloc((void**)&dmh, ye * 2)); CK(hipMalloc((void**)&dml, ye * 2));
        CK(hipMalloc((void**)&y, ye * 4)); CK(hipMalloc((void**)&c, be * 4));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, ahi, (long long)ae, 1u, 0); hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, alo, (long long)ae, 2u, 1);
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, bhi, (long long)be, 3u, 0); hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, blo, (long long)be, 4u, 1);
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, dmh, (long long)ye, 5u, 0); hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, dml, (long long)ye, 6u, 1);
        CK(hipDeviceSynchronize());
        const double gf = 2.0 * sh.P * sh.T * sh.K * sh.N * 1e-9;
        GemmArgs g{ahi, alo, bhi, blo, y, sh.T, sh.N, sh.K, Tp * sh.K, (long long)sh.N * sh.K, Tp * sh.N};
        const float tg = timeit([&] { hipLaunchKernelGGL((gemm_planes<3>), dim3((unsigned)((sh.T + 127) / 128 * (sh.N / 128)), 1, (unsigned)sh.P), dim3(256), 0, 0, g); });
        // weight gradient: C[K][N] = sum_t V[t][K] dM[t][N]; rows split so that the launch has ~2 rounds of 768 blocks
        const int tiles = (sh.K / 128) * (sh.N / 128);
        int ns = (int)fmax(1.0, fmin((double)(Tp / 16), floor(1536.0 / ((double)tiles * sh.P))));
        int chunk = (int)(((Tp + ns - 1) / ns + 15) / 16 * 16); ns = (int)((Tp + chunk - 1) / chunk);
        WgradArgs w{ahi, alo, dmh, dml, c, Tp, sh.K, sh.N, Tp * sh.K, Tp * sh.N, (long long)sh.K * sh.N, chunk, ns};
        const float tw = timeit([&] { if (ns > 1) hipMemsetAsync(c, 0, be * 4, 0); hipLaunchKernelGGL((wgrad_planes<3>), dim3((unsigned)(tiles * ns), 1, (unsigned)sh.P), dim3(256), 0, 0, w); });
        GemmIArgs gi{ahi, bhi, y, sh.T, sh.N, sh.K / 2, Tp * sh.K, (long long)sh.N * sh.K, Tp * sh.N};      // (same buffers read as [rows][2 x K/2]: half the depth, timing scaled)
        float ti = timeit([&] { hipLaunchKernelGGL((gemm_inter<3>), dim3((unsigned)((sh.T + 127) / 128 * (sh.N / 128)), 1, (unsigned)sh.P), dim3(256), 0, 0, gi); });
        printf("%-52s gemm_planes %7.3f ms %7.1f TF/s   gemm_inter (half depth) %7.3f ms %7.1f TF/s   wgrad_planes %7.3f ms %7.1f TF/s  (row splits %d)\n", sh.name, tg, gf / tg, ti, gf / 2 / ti, tw, gf / tw, ns);
        CK(hipFree(ahi)); CK(hipFree(alo)); CK(hipFree(bhi)); CK(hipFree(blo)); CK(hipFree(dmh)); CK(hipFree(dml)); CK(hipFree(y)); CK(hipFree(c));
    }
    return 0;
}
