// Lab (round 4): does a producer -> consumer pair of HBM-bound kernels run faster when the intermediate is small enough to stay in the
// 256 MB memory-side Infinity Cache?  W writes S bytes (16 bytes per lane, one contiguous chunk per block), R reads them back; the pair is
// timed for S from 16 MB to 2 GB.  If the cache serves it, (2 S / t) rises well above the ~5-6 TB/s of a DRAM stream for S below ~200 MB:
// then a layer pipeline that runs image group by image group (V -> GEMM -> M -> transform) keeps its intermediates off the DRAM bus.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_lab.hip -o scratch/mall_lab && scratch/mall_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void wk(float4* p, long long n4, float v)
{
    const long long per = (n4 + gridDim.x - 1) / gridDim.x, a = blockIdx.x * per, b = (a + per < n4) ? a + per : n4;
    for (long long i = a + threadIdx.x; i < b; i += 256) p[i] = make_float4(v, v + 1.f, v + 2.f, (float)i);
}
__global__ __launch_bounds__(256) void rk(const float4* p, long long n4, float* out)
{
    const long long per = (n4 + gridDim.x - 1) / gridDim.x, a = blockIdx.x * per, b = (a + per < n4) ? a + per : n4;
    float acc = 0.f;
    for (long long i = a + threadIdx.x; i < b; i += 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
// copy: reads src (S bytes), writes dst (S bytes)
__global__ __launch_bounds__(256) void ck(const float4* s, float4* d, long long n4)
{
    const long long per = (n4 + gridDim.x - 1) / gridDim.x, a = blockIdx.x * per, b = (a + per < n4) ? a + per : n4;
    for (long long i = a + threadIdx.x; i < b; i += 256) d[i] = s[i];
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t maxb = (size_t)2 << 30;
    float4 *buf, *buf2; float* out;
    CK(hipMalloc((void**)&buf, maxb)); CK(hipMalloc((void**)&buf2, maxb)); CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(buf, 0, maxb)); CK(hipMemset(buf2, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 8192;
    printf("%10s %28s %28s %28s %34s\n", "S", "write S then read S", "write only", "read only (same S again)", "copy chain A->B->A (2 reads 2 writes)");
    for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048}) {
        const long long n4 = (long long)mb * (1 << 20) / 16;
        const int reps = (int)(4096 / mb) + 2;
        auto time = [&](auto f) {
            f(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) f();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
        const float t_wr = time([&] { hipLaunchKernelGGL(wk, dim3(grid), dim3(256), 0, 0, buf, n4, 1.f); hipLaunchKernelGGL(rk, dim3(grid), dim3(256), 0, 0, buf, n4, out); });
        const float t_w = time([&] { hipLaunchKernelGGL(wk, dim3(grid), dim3(256), 0, 0, buf, n4, 1.f); });
        const float t_r = time([&] { hipLaunchKernelGGL(rk, dim3(grid), dim3(256), 0, 0, buf, n4, out); });
        const float t_c = time([&] { hipLaunchKernelGGL(ck, dim3(grid), dim3(256), 0, 0, buf, buf2, n4); hipLaunchKernelGGL(ck, dim3(grid), dim3(256), 0, 0, buf2, buf, n4); });
        const double S = (double)mb * (1 << 20);
        printf("%7zu MB   %8.3f ms %8.2f TB/s      %8.3f ms %8.2f TB/s      %8.3f ms %8.2f TB/s      %8.3f ms %8.2f TB/s\n", mb,
               t_wr, 2 * S / t_wr * 1e-9, t_w, S / t_w * 1e-9, t_r, S / t_r * 1e-9, t_c, 4 * S / t_c * 1e-9);
    }
    return 0;
}
