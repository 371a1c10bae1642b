// Lab for the bf16 GEMM of the bf16_fc precision mode (fc6 / fc7 forward, BASELINE config 5):
//   C[M][N] (fp32) = A[M][K] (bf16, k contiguous) x Bt[N][K] (bf16, k contiguous)^T
// 256 x 256 tile, 8 wave64 (2 row groups x 4 column waves, 128 x 64 per wave = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16), BK = 32,
// four 32 KB LDS stages filled by LDS-DMA (global_load_lds_dwordx4, 64-byte rows, XOR-swizzled 16-byte chunks), and the two row
// groups STAGGERED by one barrier: while the waves of one group read their fragments of K-tile kt from LDS, the waves of the other
// group -- one per SIMD each -- issue the 16 MFMAs of their previous tile, so the matrix pipe of every SIMD always has a wave in its
// MFMA phase.  One barrier per tick, counted vmcnt (never 0 in the main loop).
//
//   hipcc -O3 --offload-arch=gfx950 tools/bf16_lab.hip -o scratch/bf16_lab && scratch/bf16_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
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

struct Args { const unsigned short* A; const unsigned short* Bt; float* C; int M, N, K, lda, ldb, ldc, m_fastest, mode; };   // mode 1: no LDS-DMA in the loop, 2: no MFMAs, 3: no fragment reads (diagnosis)

constexpr int BM = 256, BN = 256, BK = 32, S = 5;       // 5 x 32 KB = all 160 KB of LDS: four K-tiles (128 KB) in flight
constexpr int ROWB = BK * 2;                       // bytes per LDS row (64)
constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;

__global__ __launch_bounds__(512, 1) void gemm_bf16_256(const Args p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * STAGE_BYTES];       // 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const unsigned ntm = (unsigned)(p.M / BM), ntn = (unsigned)(p.N / BN);
    const unsigned lid = xcd_run(blockIdx.x, gridDim.x);
    const unsigned tmi = p.m_fastest ? lid % ntm : lid / ntn, tni = p.m_fastest ? lid / ntm : lid % ntn;
    const long long m0 = (long long)tmi * BM; const int n0 = (int)tni * BN;

    // LDS-DMA: wave w fills 16-row chunks 2w, 2w+1 of the A image and of the B image of a stage
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + lane / 4, pc = lane % 4;
        const int lc = pc ^ ((row >> 2) & 3);                                  // logical 16-byte chunk stored at physical chunk pc
        a_voff[i] = (unsigned)(row * p.lda + lc * 8) * 2u;
        b_voff[i] = (unsigned)(row * p.ldb + lc * 8) * 2u;
    }
    const unsigned short* a_base = p.A + m0 * p.lda;
    const unsigned short* b_base = p.Bt + (long long)n0 * p.ldb;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int kt) {
        if ((p.mode == 1 || p.mode == 4 || p.mode == 5) && kt >= S - 1) return;
        const unsigned st = lds0 + (unsigned)((kt % S) * STAGE_BYTES);
        const unsigned short* ga = a_base + (long long)kt * BK;
        const unsigned short* gb = b_base + (long long)kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(ga, a_voff[i], st + (wave * 2 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(gb, b_voff[i], st + A_BYTES + (wave * 2 + i) * 1024);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage
    int a_row[4], b_row[2];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) a_row[tm] = grp * 128 + tm * 32 + (lane & 31);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) b_row[tn] = wn * 64 + tn * 32 + (lane & 31);
    bf16x8 af[2][4], bfr[2][2];
    auto load_frags = [&](int kt) {
        if ((p.mode == 3 || p.mode == 4 || p.mode == 5) && kt > 0) return;
        const unsigned char* st = smem + (kt % S) * STAGE_BYTES;
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads are DONE in this tick: the stage may be refilled two ticks later
    };
    auto mfma_phase = [&]() {
        if (p.mode == 2) return;
        if (p.mode != 6) __builtin_amdgcn_s_setprio(1);      // (mode 6: without the priority)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][tm], bfr[ks][tn], acc[tm][tn], 0, 0, 0);
        if (p.mode != 6) __builtin_amdgcn_s_setprio(0);
    };

    const int nkt = p.K / BK;
    // prologue: tiles 0 .. 2 in flight, tile 0 landed everywhere
#pragma unroll
    for (int t = 0; t < S - 1; ++t) if (t < nkt) issue(t);
    if (nkt > 3) wait_vmcnt<12>(); else if (nkt > 2) wait_vmcnt<8>(); else if (nkt > 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // tick t: group g runs step t - g; even steps read the fragments of K-tile step / 2, odd steps multiply them.  Even ticks issue the
    // LDS-DMA of K-tile t / 2 + 3 (its stage was last read in tick t - 1); odd ticks t = 2 kt + 1 wait for K-tile kt + 1.  The two
    // groups run their own straight-line loops (one loop with per-tick branches made the compiler copy the accumulators around).
    // (sched_barrier: without it hipcc hoists a group's MFMAs above the barrier that opens its MFMA phase and threads them between its
    //  ds_reads -- legal, the MFMAs only depend on the reads -- which puts both groups' MFMAs into the same ticks)
    auto tick_end = [&]() { __builtin_amdgcn_sched_barrier(0); if (p.mode != 5) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    auto wait_tile = [&](int kt) {      // this wave's pieces of K-tile kt have landed; up to two newer tiles may still be in flight
        if (kt + 3 < nkt) wait_vmcnt<12>(); else if (kt + 2 < nkt) wait_vmcnt<8>(); else if (kt + 1 < nkt) wait_vmcnt<4>(); else wait_vmcnt<0>();
    };
    if (grp == 0) {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + S - 1 < nkt) issue(kt + S - 1);
            load_frags(kt);
            tick_end();               // end of tick 2 kt
            mfma_phase();
            wait_tile(kt + 1);
            tick_end();               // end of tick 2 kt + 1
        }
        tick_end();                   // tick 2 nkt: group 1's last MFMA phase
    } else {
        if (S - 1 < nkt) issue(S - 1);
        tick_end();                   // tick 0: group 0 reads tile 0
        for (int kt = 0; kt < nkt; ++kt) {
            load_frags(kt);
            wait_tile(kt + 1);
            tick_end();               // end of tick 2 kt + 1
            if (kt + S < nkt) issue(kt + S);
            mfma_phase();
            tick_end();               // end of tick 2 kt + 2
        }
    }

    // epilogue: fp32 stores (32 consecutive columns per lane half)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + grp * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                p.C[m * p.ldc + col] = acc[tm][tn][r];
            }
    }
}

static unsigned short f2bf(float f)
{
    unsigned u; memcpy(&u, &f, 4);
    const unsigned r = 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)((u + r) >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

static double run_case(int M, int N, int K, bool check, int reps, int mode = 0)
{
    std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc((void**)&dA, hA.size() * 2)); CK(hipMalloc((void**)&dB, hB.size() * 2)); CK(hipMalloc((void**)&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4));
    Args a{dA, dB, dC, M, N, K, K, K, N, 0, mode};
    a.m_fastest = (double)N * K > (double)M * K;
    const dim3 grid((unsigned)((M / BM) * (N / BN)));
    hipLaunchKernelGGL(gemm_bf16_256, grid, dim3(512), 0, 0, a);
    CK(hipDeviceSynchronize());
    double maxerr = 0;
    if (check) {
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        const int step_m = M > 512 ? 97 : 1, step_n = N > 512 ? 89 : 1;
        for (int m = 0; m < M; m += step_m)
            for (int n = 0; n < N; n += step_n) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
                const double e = fabs(ref - hC[(size_t)m * N + n]) / (1.0 + fabs(ref));
                if (!(e <= maxerr)) maxerr = e;
            }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16_256, grid, dim3(512), 0, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
    printf("M %5d N %5d K %5d: %8.3f ms  %7.1f TFLOP/s  (%.2f of 2500)%s", M, N, K, ms, tf, tf / 2500.0, check ? "" : "\n");
    if (check) printf("   max rel err %.2e %s\n", maxerr, maxerr < 2e-5 ? "ok" : "WRONG");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    return tf;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    run_case(256, 256, 32, true, 1);
    run_case(256, 256, 64, true, 1);
    run_case(256, 256, 96, true, 1);
    run_case(512, 512, 256, true, 1);
    run_case(1024, 768, 416, true, 2);
    run_case(8192, 4096, 4096, true, 10);        // fc7 (16 x 1024x512)
    run_case(8192, 4096, 25088, true, 3);        // fc6 as a plain GEMM of the same shape
    printf("diagnosis at the fc7 shape: no LDS-DMA in the loop / no MFMAs / no fragment reads\n");
    run_case(8192, 4096, 4096, false, 10, 1);
    run_case(8192, 4096, 4096, false, 10, 2);
    run_case(8192, 4096, 4096, false, 10, 3);
    printf("WITHOUT s_setprio(1) around the MFMA phase (fc7 / fc6 shapes)\n");
    run_case(8192, 4096, 4096, false, 10, 6);
    run_case(8192, 4096, 25088, false, 3, 6);
    printf("MFMAs + barriers only / MFMAs only\n");
    run_case(8192, 4096, 4096, false, 10, 4);
    run_case(8192, 4096, 4096, false, 10, 5);
    run_case(4096, 4096, 4096, false, 10);
    run_case(8192, 8192, 8192, false, 5);
    return 0;
}
