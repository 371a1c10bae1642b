// Lab (round 4): what holds the Winograd transforms at 4.8 TB/s when a plain stream reaches 6.2-6.8 on this box?
// A stand-alone F(6x6,3x3) input transform (one thread = one tile x 2 channels, 64 8-byte loads of the 8x8 patch, V = B^T d B, 64 8-byte
// stores into the [P][T][C] image -- the production kernel's arithmetic and addressing) with one thing changed at a time:
//   base      the production form
//   ro / wo   the same kernel with its stores / its loads removed (what each stream costs alone, in this access pattern)
//   tpb N     a block walks N consecutive 256-thread chunks of a tile row (longer contiguous runs per slab and per block)
//   st16      16-byte stores: lanes (c, c+1) swap halves so that the even lane writes position 2j for both, the odd lane position 2j+1
//   ldst16    ... and 16-byte loads the same way (even lane fetches pixel 2j for both, odd lane pixel 2j+1)
//   hipcc -O3 --offload-arch=gfx950 tools/xform_lab.hip -o scratch/xform_lab && scratch/xform_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct alignas(8) V2 { float d[2]; };
struct alignas(16) V4 { float d[4]; };
static __device__ __forceinline__ float bt(int i, int j)
{
    constexpr float m[8][8] = {{4, 0, -21, 0, 21, 0, -4, 0}, {0, -4, -4, 17, 17, -4, -4, 0}, {0, 4, -4, -17, 17, 4, -4, 0}, {0, 2, 1, -10, -5, 8, 4, 0},
                               {0, -2, 1, 10, -5, -8, 4, 0}, {0, 4, 8, -5, -10, 1, 2, 0}, {0, -4, 8, 5, -10, -1, 2, 0}, {0, -4, 0, 21, 0, -21, 0, 4}};
    return m[i][j];
}
static __device__ __forceinline__ V2 fma2(float s, const V2& a, V2 acc) { acc.d[0] = fmaf(s, a.d[0], acc.d[0]); acc.d[1] = fmaf(s, a.d[1], acc.d[1]); return acc; }
static __device__ __forceinline__ V2 xchg(const V2& v) { V2 r; r.d[0] = __shfl_xor(v.d[0], 1); r.d[1] = __shfl_xor(v.d[1], 1); return r; }

// MODE: 0 base, 1 read-only, 2 write-only, 3 16-byte stores, 4 16-byte loads and stores
template <int MODE>
__global__ __launch_bounds__(256) void xin(const V2* __restrict__ x, V2* __restrict__ v, int N, int H, int W, int C4, long long slab, int tpb, float* sink, int order = 0, int strip = 4)
{
    constexpr int A = 8, M = 6;
    const int th = (H + M - 1) / M, tw = (W + M - 1) / M;
    // block order.  0: as dispatched (x fastest; consecutive blocks go to consecutive XCDs).  1: XCD k (= dispatch id % 8) takes the k-th
    // eighth of the row-major block list.  2: the same over a list in which a strip of `strip` chunks walks down all tile rows before the
    // next strip starts (vertically adjacent tiles a few blocks apart on ONE XCD: their shared two pixel rows can meet in its L2).  3: strips, no XCD remap.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (order) {
        const unsigned gx = gridDim.x, gy = gridDim.y, B = gx * gy, L = by * gx + bx;
        unsigned Lp = L;
        if (order == 1 || order == 2) { const unsigned per = (B + 7) / 8; Lp = (L % 8) * per + L / 8; if (Lp >= B) return; }      // (ragged tail: a few blocks idle in the lab)
        if (order == 1) { by = Lp / gx; bx = Lp % gx; }
        else {
            const unsigned S = (unsigned)strip, full = gx / S * S;          // chunks beyond the last whole strip form a narrower one
            const unsigned in_full = full * gy;
            if (Lp < in_full) { const unsigned st = Lp / (S * gy), rem = Lp % (S * gy); by = rem / S; bx = st * S + rem % S; }
            else { const unsigned w = gx - full, rem = Lp - in_full; by = rem / w; bx = full + rem % w; }
        }
    }
    const int n = by / th, ty = by - n * th;
    for (int it = 0; it < tpb; ++it) {
        const int idx = (bx * tpb + it) * 256 + threadIdx.x;
        const int tx = idx / C4, c = idx - tx * C4;
        if (tx >= tw) break;
        const long long t = ((long long)n * th + ty) * tw + tx;
        const int y0 = M * ty - 1, x0 = M * tx - 1;
        const V2* xp = x + (((long long)n * H + y0) * W + x0) * C4 + c;
        bool rok[A], cok[A];
#pragma unroll
        for (int a = 0; a < A; ++a) { rok[a] = (unsigned)(y0 + a) < (unsigned)H; cok[a] = (unsigned)(x0 + a) < (unsigned)W; }
        const bool odd = c & 1;
        V2 q[A][A];
#pragma unroll
        for (int b = 0; b < A; ++b) {
            V2 d[A];
            if (MODE == 2) {
#pragma unroll
                for (int a = 0; a < A; ++a) { d[a].d[0] = (float)(threadIdx.x + a); d[a].d[1] = (float)(b + c); }
            } else if (MODE == 4) {
                // columns in pairs (b, b+1): the even lane fetches 16 bytes of pixel column b (its own two channels and its partner's), the odd
                // lane 16 bytes of column b+1; one exchange gives each lane both columns of its own channels
                if ((b & 1) == 0) {
                    V4 w[A];
#pragma unroll
                    for (int a = 0; a < A; ++a) {
                        const int col = b + (odd ? 1 : 0);
                        const bool ok = rok[a] && (odd ? cok[b + 1] : cok[b]);
                        const V4* p4 = reinterpret_cast<const V4*>(xp + (a * W + col) * C4 - (odd ? 1 : 0));
                        w[a] = ok ? *p4 : V4{{0.f, 0.f, 0.f, 0.f}};
                    }
#pragma unroll
                    for (int a = 0; a < A; ++a) {
                        V2 mine, give;                         // even: mine = column b (w.xy), give = partner's column b (w.zw); odd: mine = column b+1 (w.zw), give = w.xy
                        mine.d[0] = odd ? w[a].d[2] : w[a].d[0]; mine.d[1] = odd ? w[a].d[3] : w[a].d[1];
                        give.d[0] = odd ? w[a].d[0] : w[a].d[2]; give.d[1] = odd ? w[a].d[1] : w[a].d[3];
                        const V2 got = xchg(give);             // even receives its column b+1, odd its column b
                        d[a] = odd ? got : mine;               // column b
                        q[a][b + 1] = odd ? mine : got;        // column b+1 parked in q (overwritten below after use)
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < A; ++a) d[a] = q[a][b];
                }
            } else {
#pragma unroll
                for (int a = 0; a < A; ++a) d[a] = (rok[a] && cok[b]) ? xp[(a * W + b) * C4] : V2{{0.f, 0.f}};
            }
#pragma unroll
            for (int a = 0; a < A; ++a) {
                V2 s{{0.f, 0.f}};
#pragma unroll
                for (int k = 0; k < A; ++k) if (bt(a, k) != 0.f) s = fma2(bt(a, k), d[k], s);
                q[a][b] = s;
            }
        }
        V2* vp = v + t * C4 + c;
        if (MODE == 1) {
            V2 acc{{0.f, 0.f}};
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = 0; b < A; ++b) { acc.d[0] += q[a][b].d[0]; acc.d[1] += q[a][b].d[1]; }
            if (acc.d[0] == 1.2345f) sink[0] = acc.d[1];
            continue;
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            V2 row[A];
#pragma unroll
            for (int b = 0; b < A; ++b) {
                V2 s{{0.f, 0.f}};
#pragma unroll
                for (int k = 0; k < A; ++k) if (bt(b, k) != 0.f) s = fma2(bt(b, k), q[a][k], s);
                row[b] = s;
            }
            if (MODE == 3 || MODE == 4) {
#pragma unroll
                for (int b = 0; b < A; b += 2) {
                    const V2 got = xchg(odd ? row[b] : row[b + 1]);        // even sends its position b+1, odd its position b
                    V4 o;
                    if (!odd) { o.d[0] = row[b].d[0]; o.d[1] = row[b].d[1]; o.d[2] = got.d[0]; o.d[3] = got.d[1]; }
                    else { o.d[0] = got.d[0]; o.d[1] = got.d[1]; o.d[2] = row[b + 1].d[0]; o.d[3] = row[b + 1].d[1]; }
                    *reinterpret_cast<V4*>(vp - (odd ? 1 : 0) + (a * A + b + (odd ? 1 : 0)) * slab) = o;
                }
            } else {
#pragma unroll
                for (int b = 0; b < A; ++b) vp[(a * A + b) * slab] = row[b];
            }
        }
    }
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

__global__ void sumk(const float* p, long long n, double* out)
{
    double acc = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) acc += (double)p[i] * (double)((i % 7) + 1);
    atomicAdd(out, acc);
}
__global__ void fillk(float* p, long long n) { for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) p[i] = (float)((i * 2654435761u) >> 20 & 1023) * 0.001f - 0.3f; }

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* sink; CK(hipMalloc((void**)&sink, 64));
    struct Case { const char* name; int N, H, W, C; } cases[] = {{"conv1_2 input (16 x 512x1024, C = 64)", 16, 512, 1024, 64}, {"conv2_2 input (C = 128)", 16, 256, 512, 128},
                                                                 {"conv3_2 input (C = 256)", 16, 128, 256, 256}, {"conv4_2 input (C = 512)", 16, 64, 128, 512}};
    for (auto& cs : cases) {
        const int C4 = cs.C / 2, th = (cs.H + 5) / 6, tw = (cs.W + 5) / 6;
        const long long T = (long long)cs.N * th * tw;
        const long long slabf = T * cs.C + 1088;                  // floats, as wino_slab
        const size_t xb = (size_t)cs.N * cs.H * cs.W * cs.C * 4, vb = (size_t)64 * slabf * 4;
        float *x, *v; CK(hipMalloc((void**)&x, xb)); CK(hipMalloc((void**)&v, vb));
        hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, x, (long long)(xb / 4)); CK(hipMemset(v, 0, vb));
        const double rd = (double)xb, wr = 64.0 * T * cs.C * 4;
        printf("%s: T = %lld tiles, x %.2f GB, V %.2f GB\n", cs.name, T, rd * 1e-9, wr * 1e-9);
        auto grid = [&](int tpb) { const int chunks = (tw * C4 + 255) / 256; return dim3((unsigned)((chunks + tpb - 1) / tpb), (unsigned)(cs.N * th)); };
        double* dsum; CK(hipMalloc((void**)&dsum, 8));
        auto rep = [&](const char* nm, float ms, double bytes) {
            double h = 0; CK(hipMemset(dsum, 0, 8));
            hipLaunchKernelGGL(sumk, dim3(2048), dim3(256), 0, 0, v, (long long)(vb / 4), dsum);
            CK(hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost));
            printf("   %-52s %8.3f ms  %6.2f TB/s   checksum %.9e\n", nm, ms, bytes / ms * 1e-9, h); };
#define RUN(MODE, tpb) timeit([&] { hipLaunchKernelGGL((xin<MODE>), grid(tpb), dim3(256), 0, 0, (const V2*)x, (V2*)v, cs.N, cs.H, cs.W, C4, slabf / 2, tpb, sink); })
        rep("base (8-byte lanes, one chunk per block)", RUN(0, 1), rd + wr);
        rep("read-only  (same loads, no stores)", RUN(1, 1), rd);
        rep("write-only (no loads, same stores)", RUN(2, 1), wr);
        rep("tpb 2", RUN(0, 2), rd + wr);
        rep("tpb 4", RUN(0, 4), rd + wr);
        rep("tpb 8", RUN(0, 8), rd + wr);
        rep("tpb 32 (a block walks a whole tile row)", RUN(0, 32), rd + wr);
        rep("16-byte stores (lane pairs swap halves)", RUN(3, 1), rd + wr);
        rep("16-byte loads and stores", RUN(4, 1), rd + wr);
        rep("16-byte loads and stores, tpb 4", RUN(4, 4), rd + wr);
#define RUNO(MODE, order, strip) timeit([&] { hipLaunchKernelGGL((xin<MODE>), grid(1), dim3(256), 0, 0, (const V2*)x, (V2*)v, cs.N, cs.H, cs.W, C4, slabf / 2, 1, sink, order, strip); })
        rep("order: XCD-banded, row-major", RUNO(0, 1, 4), rd + wr);
        rep("order: XCD-banded strips of 2 chunks", RUNO(0, 2, 2), rd + wr);
        rep("order: XCD-banded strips of 4 chunks", RUNO(0, 2, 4), rd + wr);
        rep("order: XCD-banded strips of 8 chunks", RUNO(0, 2, 8), rd + wr);
        rep("order: strips of 4 chunks, no XCD remap", RUNO(0, 3, 4), rd + wr);
        rep("read-only, XCD-banded strips of 4", RUNO(1, 2, 4), rd);
        rep("16-byte lanes + XCD-banded strips of 4", RUNO(4, 2, 4), rd + wr);
        rep("base again", RUN(0, 1), rd + wr);
        CK(hipGetLastError());
        CK(hipFree(x)); CK(hipFree(v));
    }
    return 0;
}
