// Lab (round 4): the position GEMMs of the 64 -> 64 channel layers (conv1_2 forward, its data gradient) are HBM-bound: they read V and
// write M (3.86 GB each at 16 x 1024x512) only for the output transform to read M again.  Could one block own ALL 64 Winograd positions
// of a group of 16 tiles -- accumulators in registers (64 positions x 16 tiles x 64 channels = 128 registers per lane over 512 lanes),
// operands straight from global memory / L2 into v_mfma_f32_16x16x4_f32 -- and run the output transform from LDS, so that M never exists?
// This lab times that main loop (and checks it) before anything is built on it:
//   mode 0  the loop, M written to global memory (the GEMM as it is today: checked against a host evaluation of sampled entries)
//   mode 1  the loop alone (accumulators folded into a guard that never fires)
//   mode 2  the loop + the accumulators handed through LDS to 512 "output transform" lanes (two passes of 32 channels: 128 KB), which
//           read their 64 positions back and write one 6 x 6 tile each (a stand-in with the transform's memory pattern, not its arithmetic)
//   hipcc -O3 --offload-arch=gfx950 tools/fused64_lab.hip -o scratch/fused64_lab && scratch/fused64_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__host__ __device__ static inline float hashf(unsigned long long i, unsigned salt)
{
    unsigned long long z = (i + salt * 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
    return (float)((long long)(z & 0xFFFF) - 32768) / 32768.f;
}
__global__ void fill_v(float* v, long long slab, long long T)
{
    const long long per = T * 64;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < 64 * per; i += 256ll * gridDim.x) { const long long p = i / per, r = i - p * per; v[p * slab + r] = hashf(i, 1); }
}
// U[pos][ci][co] = hashf(idx, 2) * 0.125, stored in the kernel's order: [pos][kc][nb][lane][s] <- U[pos][16 kc + 4 (lane / 16) + s][16 nb + lane % 16]
__global__ void fill_u(float* up)
{
    const int i = blockIdx.x * 256 + threadIdx.x;          // 64 * 4096
    if (i >= 64 * 4096) return;
    const int s = i & 3, lane = (i >> 2) & 63, nb = (i >> 8) & 3, kc = (i >> 10) & 3, pos = i >> 12;
    const int ci = 16 * kc + 4 * (lane >> 4) + s, co = 16 * nb + (lane & 15);
    up[i] = hashf((unsigned long long)(pos * 64 + ci) * 64 + co, 2) * 0.125f;
}

template <int MODE, int D>
__global__ __launch_bounds__(512, 1) void fused64(const float* __restrict__ V, const float* __restrict__ Up, float* __restrict__ M, long long slab, float* __restrict__ Y)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long t0 = (long long)blockIdx.x * 16;
    const int ti = lane & 15, q = lane >> 4;
    f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    const float* vbase = V + (long long)(8 * wave) * slab + (t0 + ti) * 64 + q * 4;
    const float* ubase = Up + (long long)(8 * wave) * 4096 + lane * 4;
    // granule g = (position pi = g / 4, channel chunk kc = g % 4): one 16-byte A load and four 16-byte B loads per lane, 16 MFMAs.
    // Loads return in issue order, so the HBM-latency A loads and the L2-latency B loads share ONE prefetch distance D (granules).
    f4 ra[D + 1], rb[D + 1][4];
    auto issue = [&](int g) {
        const int npi = g >> 2, nkc = g & 3, sl = g % (D + 1);
        ra[sl] = *reinterpret_cast<const f4*>(vbase + npi * slab + nkc * 16);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) rb[sl][nb] = *reinterpret_cast<const f4*>(ubase + ((npi * 4 + nkc) * 4 + nb) * 256);
    };
#pragma unroll
    for (int g = 0; g < D; ++g) issue(g);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        if (g + D < 32) issue(g + D);
        __builtin_amdgcn_sched_barrier(0);            // (without it the compiler sinks the loads next to their use: every D compiles to the same code)
        const int pi = g >> 2, sl = g % (D + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[pi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[sl][s], rb[sl][nb][s], acc[pi][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 0) {
#pragma unroll
        for (int pi = 0; pi < 8; ++pi)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) M[(long long)(8 * wave + pi) * slab + (t0 + 4 * q + r) * 64 + 16 * nb + ti] = acc[pi][nb][r];
    } else if (MODE == 1) {
        float s = 0.f;
#pragma unroll
        for (int pi = 0; pi < 8; ++pi)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) s += acc[pi][nb][0] + acc[pi][nb][1] + acc[pi][nb][2] + acc[pi][nb][3];
        if (s == 1.2345678f) Y[0] = s;
    } else {
        // two passes of 32 channels: lds[pos 64][tile 16][33 (32 channels + 1 pad)]
        constexpr int LT = 33;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
#pragma unroll
            for (int pi = 0; pi < 8; ++pi)
#pragma unroll
                for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lds[((8 * wave + pi) * 16 + 4 * q + r) * LT + 16 * nb2 + ti] = acc[pi][2 * half + nb2][r];
            __syncthreads();
            // lane = (tile, channel): 16 x 32 = 512
            const int tl = tid >> 5, c = tid & 31;
            float m[64];
#pragma unroll
            for (int p = 0; p < 64; ++p) m[p] = lds[(p * 16 + tl) * LT + c];
            // stand-in for A^T m A: 36 outputs, each a sum over a different subset (keeps all 64 values live and costs ~ the transform's FMAs)
            float* yp = Y + ((t0 + tl) * 36) * 64 + 32 * half + c;
#pragma unroll
            for (int o = 0; o < 36; ++o) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) s = fmaf(m[(o + 4 * k) & 63], 0.5f + (float)(k & 3), s);
                yp[o * 64] = fmaxf(s, 0.f);
            }
        }
    }
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const long long T = 16ll * 86 * 171, slab = T * 64 + 1088;
    float *V, *M, *Up, *Y;
    CK(hipMalloc((void**)&V, 64 * slab * 4)); CK(hipMalloc((void**)&M, 64 * slab * 4)); CK(hipMalloc((void**)&Up, 64 * 4096 * 4)); CK(hipMalloc((void**)&Y, T * 36 * 64 * 4));
    hipLaunchKernelGGL(fill_v, dim3(16384), dim3(256), 0, 0, V, slab, T);
    hipLaunchKernelGGL(fill_u, dim3(64 * 4096 / 256), dim3(256), 0, 0, Up);
    CK(hipMemset(M, 0, 64 * slab * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(T / 16);
    const size_t lds2 = 64 * 16 * 33 * 4;
    auto time = [&](auto f, int reps) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    const double gflop = 2.0 * 64 * T * 64 * 64 * 1e-9;
    printf("T = %lld tiles, %u blocks of 16 tiles x 64 positions; %.1f GFLOP per launch; V = M = %.2f GB, 6 x 6 tiles %.2f GB\n", T, grid, gflop, 64 * slab * 4e-9, T * 36 * 64 * 4e-9);
    auto run = [&](auto d) {
        constexpr int D = decltype(d)::value;
        CK(hipFuncSetAttribute((const void*)fused64<2, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        CK(hipMemset(M, 0, 64 * slab * 4));
        const float t0 = time([&] { hipLaunchKernelGGL((fused64<0, D>), dim3(grid), dim3(512), 0, 0, V, Up, M, slab, Y); }, 10);
        double worst = 0;
        for (int k = 0; k < 64; ++k) {
            const int pos = (k * 37) & 63; const long long t = ((long long)k * 1234577) % T; const int co = (k * 29) & 63;
            double ref = 0;
            for (int ci = 0; ci < 64; ++ci) ref += (double)hashf(((long long)pos * T + t) * 64 + ci, 1) * (double)(hashf((unsigned long long)(pos * 64 + ci) * 64 + co, 2) * 0.125f);
            float got; CK(hipMemcpy(&got, M + pos * slab + t * 64 + co, 4, hipMemcpyDeviceToHost));
            worst = fmax(worst, fabs(got - ref));
        }
        const float t1 = time([&] { hipLaunchKernelGGL((fused64<1, D>), dim3(grid), dim3(512), 0, 0, V, Up, M, slab, Y); }, 10);
        const float t2 = time([&] { hipLaunchKernelGGL((fused64<2, D>), dim3(grid), dim3(512), lds2, 0, V, Up, M, slab, Y); }, 10);
        printf("prefetch distance %d granules: M written %.3f ms (%.1f TFLOP/s, largest error of 64 samples %.2g) | loop alone %.3f ms (%.1f) | loop + LDS hand-over + tile stores %.3f ms (%.1f)\n",
               D, t0, gflop / t0, worst, t1, gflop / t1, t2, gflop / t2);
    };
    run(std::integral_constant<int, 1>{}); run(std::integral_constant<int, 2>{}); run(std::integral_constant<int, 3>{}); run(std::integral_constant<int, 4>{});
    printf("(today: position GEMM 1.55 ms + output transform ~0.75 ms for conv1_2 forward)\n");
    return 0;
}
