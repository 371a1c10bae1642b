// Lab: why does hipMemset write 6.5 TB/s when a plain grid-stride float4 fill writes 4.2-4.7 TB/s on the same box?  Write-only variants.
//   hipcc -O3 --offload-arch=gfx950 tools/fill_lab.hip -o scratch/fill_lab && scratch/fill_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride, one float4 per thread per iteration
template <bool NT> __global__ __launch_bounds__(256) void fill_gs(f4* b, long long n)
{
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) st<NT>(b + i, v);
}
// contiguous chunk per block, lanes interleaved (each wave store covers 1 KB contiguous), U stores per iteration
template <bool NT, int U> __global__ __launch_bounds__(256) void fill_chunk(f4* b, long long n, long long chunk)
{
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const long long lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (long long i = lo + threadIdx.x; i < hi; i += 256LL * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < hi) st<NT>(b + i + u * 256, v);
    }
}
// each thread owns 64 contiguous bytes (4 float4 in a row) -> a wave covers 4 KB, but each store instruction is strided by 64 B per lane
template <bool NT> __global__ __launch_bounds__(256) void fill_thread64(f4* b, long long n)
{
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = (blockIdx.x * (long long)blockDim.x + threadIdx.x) * 4; i + 3 < n; i += stride) {
        st<NT>(b + i, v); st<NT>(b + i + 1, v); st<NT>(b + i + 2, v); st<NT>(b + i + 3, v);
    }
}
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
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t bytes = 2ull << 30; const long long n = bytes / 16;
    f4* b; CK(hipMalloc((void**)&b, bytes));
    auto rep = [&](const char* name, float ms) { printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms * 1e-9); };
    rep("hipMemset", timeit([&] { hipMemsetAsync(b, 0, bytes, 0); }));
    char nm[128];
    for (int g : {512, 1024, 2048, 8192}) {
        snprintf(nm, sizeof nm, "grid-stride fill, grid %5d", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_gs<false>), dim3(g), dim3(256), 0, 0, b, n); }));
        snprintf(nm, sizeof nm, "grid-stride fill, grid %5d, nt", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_gs<true>), dim3(g), dim3(256), 0, 0, b, n); }));
    }
    for (int g : {256, 512, 1024, 2048, 8192, 65536}) {
        const long long chunk = (n + g - 1) / g;
        snprintf(nm, sizeof nm, "chunk-per-block fill, grid %5d U1", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_chunk<false, 1>), dim3(g), dim3(256), 0, 0, b, n, chunk); }));
        snprintf(nm, sizeof nm, "chunk-per-block fill, grid %5d U4", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_chunk<false, 4>), dim3(g), dim3(256), 0, 0, b, n, chunk); }));
        snprintf(nm, sizeof nm, "chunk-per-block fill, grid %5d U4 nt", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_chunk<true, 4>), dim3(g), dim3(256), 0, 0, b, n, chunk); }));
    }
    for (int g : {1024, 4096}) {
        snprintf(nm, sizeof nm, "64 contiguous bytes per thread, grid %5d", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_thread64<false>), dim3(g), dim3(256), 0, 0, b, n); }));
        snprintf(nm, sizeof nm, "64 contiguous bytes per thread, grid %5d, nt", g); rep(nm, timeit([&] { hipLaunchKernelGGL((fill_thread64<true>), dim3(g), dim3(256), 0, 0, b, n); }));
    }
    return 0;
}
