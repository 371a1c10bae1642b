// Lab for VERDICT round 3 item 3(d): what conv_bf16_256_kernel (gemm_bf16.hip) loses outside its K loop, and two ways to get it back.
//   C[M][N] (fp32) = A (bf16, row m starts at A + m * lda, K contiguous values) x Bt[N][K]^T     -- lda < K makes consecutive rows overlap
//   like the taps of a convolution over the padded bf16 copy (lda = Cin, K = 9 Cin: a 9-tap 1-D stencil with the same L2 reuse).
// Same main loop as tools/bf16_lab.hip (256 x 256 tile, 8 waves, five 32 KB LDS-DMA stages, two staggered wave groups).  Variants:
//   bit 0 (SWAP)    : the MFMA operands swapped (D = B A^T), so that a lane's four consecutive accumulator rows are four consecutive OUTPUT
//                     COLUMNS of one output row: the epilogue issues 32 16-byte stores per lane instead of 128 4-byte ones, no LDS transpose
//   bit 1 (PERSIST) : one block per CU walks tiles blockIdx.x, + gridDim.x, ...; the LDS-DMA of the next tile's first four K-tiles is issued
//                     BEFORE the epilogue of the current tile (the epilogue touches no LDS), so the pipeline fill overlaps the stores
//   bit 2 (NT)      : non-temporal stores
//
//   hipcc -O3 --offload-arch=gfx950 tools/bf16_persist_lab.hip -o scratch/bf16_persist_lab && scratch/bf16_persist_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory", "m0");
}
template <int N> static __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

static __device__ __forceinline__ unsigned xcd_run(unsigned p, unsigned total)
{
    const unsigned q = total >> 3, r = total & 7u, xcd = p & 7u, i = p >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

struct Args { const unsigned short* A; const unsigned short* Bt; float* C; int M, N, K, lda, ldb, ldc, m_fastest; };

constexpr int BM = 256, BN = 256, BK = 32, S = 5;
constexpr int ROWB = BK * 2;
constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;

template <int V>
__global__ __launch_bounds__(512, 1) void gemm_bf16_256(const Args p)
{
    constexpr bool SWAP = V & 1, PERSIST = V & 2, NT = V & 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const unsigned ntm = (unsigned)(p.M / BM), ntn = (unsigned)(p.N / BN), ntiles = ntm * ntn;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int nkt = p.K / BK;

    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + lane / 4, pc = lane % 4;
        const int lc = pc ^ ((row >> 2) & 3);
        a_voff[i] = (unsigned)(row * p.lda + lc * 8) * 2u;
        b_voff[i] = (unsigned)(row * p.ldb + lc * 8) * 2u;
    }
    int a_row[4], b_row[2];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) a_row[tm] = grp * 128 + tm * 32 + (lane & 31);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) b_row[tn] = wn * 64 + tn * 32 + (lane & 31);

    // tile walk: tiles are numbered so that the blocks of one XCD (blockIdx % 8) take a contiguous run, as in the one-tile-per-block form
    auto tile_of = [&](unsigned t, long long& m0, int& n0) {
        const unsigned lid = xcd_run(t, ntiles);
        const unsigned tmi = p.m_fastest ? lid % ntm : lid / ntn, tni = p.m_fastest ? lid / ntm : lid % ntn;
        m0 = (long long)tmi * BM; n0 = (int)tni * BN;
    };
    const unsigned short* a_base; const unsigned short* b_base;
    auto issue = [&](int kt, int stage) {
        const unsigned st = lds0 + (unsigned)(stage * STAGE_BYTES);
        const unsigned short* ga = a_base + (long long)kt * BK;
        const unsigned short* gb = b_base + (long long)kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(ga, a_voff[i], st + (wave * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(gb, b_voff[i], st + A_BYTES + (wave * 2 + i) * 1024);
    };
    bf16x8 af[2][4], bfr[2][2];
    auto load_frags = [&](int stage) {
        const unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const int r = a_row[tm], pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                af[ks][tm] = *reinterpret_cast<const bf16x8*>(st + r * ROWB + pc * 16);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int r = b_row[tn], pc = (2 * ks + (lane >> 5)) ^ ((r >> 2) & 3);
                bfr[ks][tn] = *reinterpret_cast<const bf16x8*>(st + A_BYTES + r * ROWB + pc * 16);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    f32x16 acc[4][2];
    auto mfma_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    if (SWAP) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][tn], af[ks][tm], acc[tm][tn], 0, 0, 0);
                    else      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][tm], bfr[ks][tn], acc[tm][tn], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    auto tick_end = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    auto wait_tile = [&](int kt) {
        if (kt + 3 < nkt) wait_vmcnt<12>(); else if (kt + 2 < nkt) wait_vmcnt<8>(); else if (kt + 1 < nkt) wait_vmcnt<4>(); else wait_vmcnt<0>();
    };

    // The stage of K-tile kt of the j-th tile a block works on is (base + kt) % S with base advancing by nkt per tile, so that the four tiles
    // prefetched across a tile boundary land in stages the finishing tile no longer reads.  (For the one-tile form base = 0.)
    unsigned t = PERSIST ? blockIdx.x : blockIdx.x;
    int base = 0;
    long long m0; int n0;
    tile_of(t, m0, n0);
    a_base = p.A + m0 * p.lda; b_base = p.Bt + (long long)n0 * p.ldb;
#pragma unroll
    for (int i = 0; i < S - 1; ++i) if (i < nkt) issue(i, i % S);
    for (;;) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (nkt > 3) wait_vmcnt<12>(); else if (nkt > 2) wait_vmcnt<8>(); else if (nkt > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int st = base;                                     // stage of K-tile kt
        auto adv = [&](int s, int d) { s += d; return s >= S ? s - S : s; };
        if (grp == 0) {
            for (int kt = 0; kt < nkt; ++kt) {
                if (kt + S - 1 < nkt) issue(kt + S - 1, adv(st, S - 1));
                load_frags(st);
                tick_end();
                mfma_phase();
                wait_tile(kt + 1);
                tick_end();
                st = adv(st, 1);
            }
            tick_end();
        } else {
            if (S - 1 < nkt) issue(S - 1, adv(st, S - 1));
            tick_end();
            for (int kt = 0; kt < nkt; ++kt) {
                load_frags(st);
                wait_tile(kt + 1);
                tick_end();
                if (kt + S < nkt) issue(kt + S, adv(st, S));       // == st: the stage both groups have just finished reading
                mfma_phase();
                tick_end();
                st = adv(st, 1);
            }
        }
        const long long em0 = m0; const int en0 = n0;
        bool more = false;
        if (PERSIST) {
            t += gridDim.x;
            more = t < ntiles;
            if (more) {
                // every wave is past its last fragment read (the trailing tick_end above): all five stages are free
                base = st;
                tile_of(t, m0, n0);
                a_base = p.A + m0 * p.lda; b_base = p.Bt + (long long)n0 * p.ldb;
#pragma unroll
                for (int i = 0; i < S - 1; ++i) if (i < nkt) issue(i, adv(base, i));
            }
        }
        // epilogue
        if (SWAP) {
            // D = B A^T: accumulator row index = output column (4 consecutive per register quad), accumulator column = output row (lane & 31)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const long long m = em0 + grp * 128 + tm * 32 + (lane & 31);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = en0 + wn * 64 + tn * 32 + 8 * q + 4 * (lane >> 5);
                        f32x4 v = {acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]};
                        f32x4* dst = reinterpret_cast<f32x4*>(p.C + m * p.ldc + col);
                        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
                    }
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = en0 + wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long m = em0 + grp * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (NT) __builtin_nontemporal_store(acc[tm][tn][r], p.C + m * p.ldc + col); else p.C[m * p.ldc + col] = acc[tm][tn][r];
                    }
            }
        }
        if (!more) break;
    }
}

static unsigned short f2bf(float f)
{
    unsigned u; memcpy(&u, &f, 4);
    const unsigned r = 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)((u + r) >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Buf { unsigned short *dA, *dB; float* dC; std::vector<unsigned short> hA, hB; };

template <int V>
static double run_variant(const char* name, Buf& b, int M, int N, int K, int lda, bool check, int reps)
{
    Args a{b.dA, b.dB, b.dC, M, N, K, lda, K, N, 0};
    a.m_fastest = (double)N * K > (double)M * lda;
    const unsigned ntiles = (unsigned)((M / BM) * (N / BN));
    const dim3 grid((V & 2) ? (ntiles < 256 ? ntiles : 256) : ntiles);
    CK(hipMemset(b.dC, 0xFF, (size_t)M * N * 4));
    hipLaunchKernelGGL(gemm_bf16_256<V>, grid, dim3(512), 0, 0, a);
    CK(hipDeviceSynchronize());
    double maxerr = 0;
    if (check) {
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), b.dC, hC.size() * 4, hipMemcpyDeviceToHost));
        const int step_m = M > 512 ? 997 : 1, step_n = N > 256 ? 89 : 7;
        for (int m = 0; m < M; m += step_m)
            for (int n = 0; n < N; n += step_n) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)bf2f(b.hA[(size_t)m * lda + k]) * bf2f(b.hB[(size_t)n * K + k]);
                const double e = fabs(ref - hC[(size_t)m * N + n]) / (1.0 + fabs(ref));
                if (!(e <= maxerr)) maxerr = e;
            }
        // every element written?
        size_t nan = 0; for (size_t i = 0; i < hC.size(); i += 13) if (hC[i] != hC[i]) ++nan;
        if (nan) maxerr = 1e9;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gemm_bf16_256<V>, grid, dim3(512), 0, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16_256<V>, grid, dim3(512), 0, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
    printf("   %-44s %8.3f ms  %7.1f TFLOP/s", name, ms, tf);
    if (check) printf("   max rel err %.2e %s", maxerr, maxerr < 2e-5 ? "ok" : "WRONG");
    printf("\n");
    return tf;
}

static void run_case(const char* what, int M, int N, int K, int lda, int reps)
{
    printf("%s: M %d N %d K %d lda %d (%d tiles)\n", what, M, N, K, lda, (M / BM) * (N / BN));
    Buf b;
    b.hA.resize((size_t)(M - 1) * lda + K); b.hB.resize((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : b.hA) v = f2bf(rnd());
    for (auto& v : b.hB) v = f2bf(rnd());
    CK(hipMalloc((void**)&b.dA, b.hA.size() * 2)); CK(hipMalloc((void**)&b.dB, b.hB.size() * 2)); CK(hipMalloc((void**)&b.dC, (size_t)M * N * 4));
    CK(hipMemcpy(b.dA, b.hA.data(), b.hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b.dB, b.hB.data(), b.hB.size() * 2, hipMemcpyHostToDevice));
    run_variant<0>("base (one tile per block, 4-byte stores)", b, M, N, K, lda, true, reps);
    run_variant<1>("swapped operands: 16-byte stores", b, M, N, K, lda, true, reps);
    run_variant<5>("swapped + non-temporal", b, M, N, K, lda, true, reps);
    run_variant<2>("persistent, 4-byte stores", b, M, N, K, lda, true, reps);
    run_variant<3>("persistent + swapped", b, M, N, K, lda, true, reps);
    run_variant<7>("persistent + swapped + non-temporal", b, M, N, K, lda, true, reps);
    run_variant<0>("base again", b, M, N, K, lda, false, reps);
    CK(hipFree(b.dA)); CK(hipFree(b.dB)); CK(hipFree(b.dC));
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    run_case("small check", 1024, 768, 416, 416, 2);
    run_case("small check, overlapping rows", 1024, 512, 9 * 64, 64, 2);
    // config 5's shapes (4 x 2048x1024) as 9-tap stencils over Cin-channel rows
    run_case("conv3_x (256 -> 256 at 512x256 x 4)", 524288, 256, 9 * 256, 256, 10);
    run_case("conv4_2 (512 -> 512 at 256x128 x 4)", 131072, 512, 9 * 512, 512, 10);
    run_case("conv5_x (512 -> 512 at 128x64 x 4)", 32768, 512, 9 * 512, 512, 10);
    run_case("fc6 (49 taps x 512 -> 4096, 8192 rows)", 8192, 4096, 49 * 512, 512, 5);
    run_case("fc7 (4096 -> 4096, 8192 rows)", 8192, 4096, 4096, 4096, 10);
    // the headline's shapes (16 x 1024x512)
    run_case("conv4_2 at 16 x 1024x512", 131072, 512, 9 * 512, 512, 10);
    return 0;
}
