// Does VALU work issue under a running bf16 / f32 MFMA on gfx950?  One wave per SIMD (256 threads/block, 1 block/CU via big LDS),
// or several, loop of: NM MFMAs + NV independent VALU FMAs, interleaved by sched_group_barrier or grouped.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int NV>   // MODE 0: mfma only, 1: valu only, 2: grouped (all VALU then all MFMA), 3: interleaved 1 MFMA : NV/8 VALU
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    f32x16 acc[2] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        }
        if (MODE != 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 1], 0, 0, 0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 8; ++m) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV / 8, 0); }
        } else if (MODE == 2) {
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += v[i] + acc[0][i] + acc[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NV> void run(const char* tag, float* d, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.f);
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s blocks %4d: %.3f ms  -> %.1f cycles/iter at 2.4 GHz\n", tag, blocks, ms, ms * 1e-3 * 2.4e9 / iters);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    for (int blocks : {256, 512, 768}) {
        run<0, 64>("8 bf16 MFMA only", d, blocks);
        run<1, 64>("64 VALU only", d, blocks);
        run<2, 64>("64 VALU then 8 MFMA (grouped)", d, blocks);
        run<3, 64>("1 MFMA : 8 VALU interleaved", d, blocks);
        run<1, 32>("32 VALU only", d, blocks);
        run<3, 32>("1 MFMA : 4 VALU interleaved", d, blocks);
    }
    return 0;
}
