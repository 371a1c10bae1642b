// Lab: can an HBM-bound kernel and an MFMA-bound kernel run side by side on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask) without
// slowing each other down?  (Sharing CUs is zero-sum on gfx950: profiles/r03_overlap_*.txt.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/cumask_lab.hip -o /tmp/cumask_lab && /tmp/cumask_lab
//
// 1. copy bandwidth of a streaming kernel restricted to n CUs, for two mask patterns: the first n bits, or n bits spread evenly;
// 2. rate of an MFMA spin kernel (v_mfma_f32_32x32x2_f32, no memory traffic) restricted to the complement;
// 3. both at once on disjoint masks: each one's time against its solo time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_spin_kernel(float* out, int iters)
{
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
    }
    float s = 0;
    for (int k = 0; k < 16; ++k) s += acc0[k] + acc1[k] + acc2[k] + acc3[k];
    if (s == 12345.678f) out[0] = s;
}

static hipStream_t masked_stream(const std::vector<int>& cus, int ncu)
{
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c : cus) mask[c / 32] |= 1u << (c % 32);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    return s;
}

static float time_ms(hipStream_t s, void (*launch)(hipStream_t, void*), void* ctx, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(s, ctx);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) launch(s, ctx);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct Ctx { float4 *a, *b; long long n; float* out; int iters; int blocks; };
static void launch_copy(hipStream_t s, void* c_) { Ctx* c = (Ctx*)c_; hipLaunchKernelGGL(copy_kernel, dim3(c->blocks), dim3(256), 0, s, c->a, c->b, c->n); }
static void launch_mfma(hipStream_t s, void* c_) { Ctx* c = (Ctx*)c_; hipLaunchKernelGGL(mfma_spin_kernel, dim3(c->blocks), dim3(256), 0, s, c->out, c->iters); }

int main(int argc, char** argv)
{
    // one (pattern, n) per process: `cumask_lab <pattern 0|1> <n>`; no arguments = only the unmasked figures and the shared-CU co-run
    const int arg_pattern = argc > 2 ? atoi(argv[1]) : -1, arg_n = argc > 2 ? atoi(argv[2]) : 0;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.name, ncu);
    Ctx c;
    const size_t bytes = 2ull << 30;
    c.n = bytes / 16;
    CK(hipMalloc((void**)&c.a, bytes)); CK(hipMalloc((void**)&c.b, bytes)); CK(hipMalloc((void**)&c.out, 64));
    CK(hipMemset(c.a, 1, bytes));
    c.blocks = ncu * 8; c.iters = 20000;
    // per block: 4 waves x iters x 4 MFMA x (32*32*2*2 flop)
    const double mfma_flop_per_block = 4.0 * c.iters * 4.0 * 4096.0;

    hipStream_t all; CK(hipStreamCreateWithFlags(&all, hipStreamNonBlocking));
    const float t_copy_all = time_ms(all, launch_copy, &c, 5);
    const float t_mfma_all = time_ms(all, launch_mfma, &c, 3);
    printf("unmasked: copy %.3f ms = %.2f TB/s (read+write);  mfma spin %.3f ms = %.1f TFLOP/s\n", t_copy_all, 2.0 * bytes / t_copy_all * 1e-9,
           t_mfma_all, mfma_flop_per_block * c.blocks / t_mfma_all * 1e-9);

    for (int pattern = 0; pattern < 2; ++pattern) {
        if (pattern != arg_pattern) continue;
        printf("mask pattern: %s\n", pattern == 0 ? "first n CU bits" : "n CU bits spread evenly over the mask");
        for (int n : {arg_n}) {
            std::vector<int> in, outc;
            std::vector<char> used(ncu, 0);
            if (pattern == 0) for (int i = 0; i < n; ++i) used[i] = 1;
            else for (int i = 0; i < n; ++i) used[(int)((long long)i * ncu / n)] = 1;
            for (int i = 0; i < ncu; ++i) (used[i] ? in : outc).push_back(i);
            hipStream_t sc = masked_stream(in, ncu), sm = masked_stream(outc, ncu);
            Ctx cc = c; cc.blocks = n * 8;
            Ctx cm = c; cm.blocks = (ncu - n) * 8;
            const float tc = time_ms(sc, launch_copy, &cc, 5);
            const float tm = time_ms(sm, launch_mfma, &cm, 3);
            // together: start both, time each on its own stream
            hipEvent_t a1, b1, a2, b2; hipEventCreate(&a1); hipEventCreate(&b1); hipEventCreate(&a2); hipEventCreate(&b2);
            const int rc = (int)(3.0f * tm / tc) + 1;           // copies that cover three spin launches
            CK(hipDeviceSynchronize());
            hipEventRecord(a1, sc); hipEventRecord(a2, sm);
            for (int i = 0; i < 3; ++i) launch_mfma(sm, &cm);
            for (int i = 0; i < rc; ++i) launch_copy(sc, &cc);
            hipEventRecord(b1, sc); hipEventRecord(b2, sm);
            CK(hipDeviceSynchronize());
            float tc2, tm2; hipEventElapsedTime(&tc2, a1, b1); hipEventElapsedTime(&tm2, a2, b2);
            tc2 /= rc; tm2 /= 3;
            printf("  copy on %3d CUs: %.3f ms = %.2f TB/s (%.0f%% of unmasked) | mfma on %3d CUs: %.1f TFLOP/s (%.0f%% of its share) | together: copy x%.2f, mfma x%.2f\n",
                   n, tc, 2.0 * bytes / tc * 1e-9, 100.0 * t_copy_all / tc, ncu - n, mfma_flop_per_block * cm.blocks / tm * 1e-9,
                   100.0 * (mfma_flop_per_block * cm.blocks / tm) / (mfma_flop_per_block * c.blocks / t_mfma_all * (ncu - n) / ncu), tc2 / tc, tm2 / tm);
            // (the masked streams are left to process exit: destroying one and creating the next hung on this stack)
        }
    }
    // shared CUs, for comparison: both unmasked at once
    if (arg_pattern < 0) {
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t a1, b1, a2, b2; hipEventCreate(&a1); hipEventCreate(&b1); hipEventCreate(&a2); hipEventCreate(&b2);
        const int rc = (int)(3.0f * t_mfma_all / t_copy_all) + 1;
        CK(hipDeviceSynchronize());
        hipEventRecord(a1, all); hipEventRecord(a2, s2);
        for (int i = 0; i < 3; ++i) launch_mfma(s2, &c);
        for (int i = 0; i < rc; ++i) launch_copy(all, &c);
        hipEventRecord(b1, all); hipEventRecord(b2, s2);
        CK(hipDeviceSynchronize());
        float tc2, tm2; hipEventElapsedTime(&tc2, a1, b1); hipEventElapsedTime(&tm2, a2, b2);
        printf("\nshared CUs (no masks), both at once: copy x%.2f, mfma x%.2f of their solo times\n", tc2 / rc / t_copy_all, tm2 / 3 / t_mfma_all);
    }
    return 0;
}
