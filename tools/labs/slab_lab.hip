// Lab: does the [P][T][C] Winograd image layout (64 position slabs far apart: every block keeps 64 write streams going, one per
// slab, 60 MB apart for conv1_2) cost bandwidth against a chunked layout [T/Tb][P][Tb][C] in which the 64 sub-slabs of a chunk of
// Tb tiles are contiguous (2 MB for C = 64, Tb = 128)?  One thread = one tile x 2 channels, 64 8-byte stores (or loads), as in
// winograd.hip.   hipcc -O3 --offload-arch=gfx950 tools/slab_lab.hip -o scratch/slab_lab && scratch/slab_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// layout 0: slab   addr = p * S + t * C2 + c          (float2 units; S = T * C2 + pad)
// layout 1: chunk  addr = ((t / Tb) * 64 + p) * Tb * C2 + (t % Tb) * C2 + c
template <int LAYOUT, bool READ>
__global__ __launch_bounds__(256) void k(float2* __restrict__ buf, float* out, long long T, int C2, long long S, int Tb)
{
    const long long gid = blockIdx.x * 256LL + threadIdx.x;
    const long long t = gid / C2; const int c = (int)(gid % C2);
    if (t >= T) return;
    float2 acc = {0.f, 0.f};
    long long base, pstride;
    if (LAYOUT == 0) { base = t * C2 + c; pstride = S; }
    else { base = (t / Tb) * 64LL * Tb * C2 + (t % Tb) * (long long)C2 + c; pstride = (long long)Tb * C2; }
#pragma unroll
    for (int p = 0; p < 64; ++p) {
        if (READ) { float2 v = buf[base + p * pstride]; acc.x += v.x; acc.y += v.y; }
        else buf[base + p * pstride] = make_float2((float)p, (float)c);
    }
    if (READ && acc.x == 1.2345f) out[0] = acc.y;
}
template <class F> static float timeit(F f, int reps = 5)
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
    float* o; CK(hipMalloc((void**)&o, 64));
    struct Case { const char* name; long long T; int C; } cases[] = {{"conv1_2 (16 x 512x1024, C = 64)", 16LL * 86 * 171, 64}, {"conv2_2 (C = 128)", 16LL * 43 * 86, 128},
                                                                     {"conv3_2 (C = 256)", 16LL * 22 * 43, 256}, {"conv4_2 (C = 512)", 16LL * 11 * 22, 512}};
    for (auto& cs : cases) {
        const int C2 = cs.C / 2;
        const long long T = cs.T, Tpad = (T + 511) / 512 * 512;
        const long long S = T * C2 + 544;                         // (+1088 floats, as wino_slab)
        const size_t bytes = (size_t)64 * (Tpad * C2 + 544) * 8;
        float2* buf; CK(hipMalloc((void**)&buf, bytes)); CK(hipMemset(buf, 0, bytes));
        const unsigned grid = (unsigned)((T * C2 + 255) / 256);
        const double gb = 64.0 * T * C2 * 8;
        printf("%s: T = %lld tiles, image %.2f GB\n", cs.name, T, gb * 1e-9);
        auto rep = [&](const char* nm, float ms) { printf("   %-44s %8.3f ms  %6.2f TB/s\n", nm, ms, gb / ms * 1e-9); };
        rep("write, slab layout [P][T][C]", timeit([&] { hipLaunchKernelGGL((k<0, false>), dim3(grid), dim3(256), 0, 0, buf, o, T, C2, S, 128); }));
        for (int Tb : {32, 128, 512}) {
            char nm[96]; snprintf(nm, sizeof nm, "write, chunk layout Tb = %d", Tb);
            rep(nm, timeit([&] { hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(256), 0, 0, buf, o, T, C2, S, Tb); }));
        }
        rep("read,  slab layout [P][T][C]", timeit([&] { hipLaunchKernelGGL((k<0, true>), dim3(grid), dim3(256), 0, 0, buf, o, T, C2, S, 128); }));
        for (int Tb : {32, 128, 512}) {
            char nm[96]; snprintf(nm, sizeof nm, "read,  chunk layout Tb = %d", Tb);
            rep(nm, timeit([&] { hipLaunchKernelGGL((k<1, true>), dim3(grid), dim3(256), 0, 0, buf, o, T, C2, S, Tb); }));
        }
        CK(hipFree(buf));
    }
    return 0;
}
