// Winograd F(2x2, 3x3) transforms (Lavin & Gray 2015, correlation form) around 16 batched GEMMs that run
// on igemm_fwd_kernel (MFMA).
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],
//   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
// 16 multiplies per 2x2 output tile and (ci, co) pair instead of 36.  The transforms are HBM-bound
// element-wise kernels (16-byte accesses along the channel axis); they pay for themselves only where the
// activations are small next to the arithmetic, i.e. in the deep, wide layers (conv3 .. conv5).
#include "fcn8s_internal.h"

namespace fcn8s {

static inline int wcap(long long work)
{
    long long b = (work + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

__global__ void wino_filter_kernel(const float* w, float* u, int Cin, int Cout)
{
    const long long cc = (long long)Cin * Cout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < cc; i += (long long)gridDim.x * blockDim.x) {
        float g[3][3], t[4][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = w[(a * 3 + b) * cc + i];
#pragma unroll
        for (int b = 0; b < 3; ++b) {            // t = G g
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {            // u = t G^T
            u[(a * 4 + 0) * cc + i] = t[a][0];
            u[(a * 4 + 1) * cc + i] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
            u[(a * 4 + 2) * cc + i] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
            u[(a * 4 + 3) * cc + i] = t[a][2];
        }
    }
}
void launch_wino_filter(const float* w, float* u, int Cin, int Cout, hipStream_t s)
{
    hipLaunchKernelGGL(wino_filter_kernel, dim3(wcap((long long)Cin * Cout)), dim3(256), 0, s, w, u, Cin, Cout);
}

static __device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static __device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// one thread = one 2x2 output tile x 4 channels: loads the 4x4 input patch (zero outside), V = B^T d B
__global__ __launch_bounds__(256) void wino_input_kernel(const float4* x, float4* v, int N, int H, int W, int C4)
{
    const int th = H / 2, tw = W / 2;
    const long long T = (long long)N * th * tw, total = T * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const long long t = i / C4;
        const int tx = (int)(t % tw); const long long r = t / tw;
        const int ty = (int)(r % th); const int n = (int)(r / th);
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int iy = 2 * ty - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ix = 2 * tx - 1 + b;
                d[a][b] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                              ? x[(((long long)n * H + iy) * W + ix) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float4 q[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {            // q = B^T d
            q[0][b] = f4sub(d[0][b], d[2][b]);
            q[1][b] = f4add(d[1][b], d[2][b]);
            q[2][b] = f4sub(d[2][b], d[1][b]);
            q[3][b] = f4sub(d[1][b], d[3][b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {            // V = q B
            v[((long long)(a * 4 + 0) * T + t) * C4 + c] = f4sub(q[a][0], q[a][2]);
            v[((long long)(a * 4 + 1) * T + t) * C4 + c] = f4add(q[a][1], q[a][2]);
            v[((long long)(a * 4 + 2) * T + t) * C4 + c] = f4sub(q[a][2], q[a][1]);
            v[((long long)(a * 4 + 3) * T + t) * C4 + c] = f4sub(q[a][1], q[a][3]);
        }
    }
}
void launch_wino_input(const float* x, float* v, int N, int H, int W, int C, hipStream_t s)
{
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3(wcap(total)), dim3(256), 0, s, (const float4*)x, (float4*)v, N, H, W, C / 4);
}

// one thread = one tile x 4 channels: Y = A^T M A, then the conv epilogue (bias, skip add, ReLU / ReLU-mask)
__global__ __launch_bounds__(256) void wino_output_kernel(const float4* m, const float4* bias, const float4* addend, const float4* mask,
                                                          float mask_scale, int relu, float4* y, int N, int H, int W, int C4)
{
    const int th = H / 2, tw = W / 2;
    const long long T = (long long)N * th * tw, total = T * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const long long t = i / C4;
        const int tx = (int)(t % tw); const long long r = t / tw;
        const int ty = (int)(r % th); const int n = (int)(r / th);
        float4 q[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {            // q = A^T M
            const float4 m0 = m[((long long)(0 * 4 + b) * T + t) * C4 + c], m1 = m[((long long)(1 * 4 + b) * T + t) * C4 + c];
            const float4 m2 = m[((long long)(2 * 4 + b) * T + t) * C4 + c], m3 = m[((long long)(3 * 4 + b) * T + t) * C4 + c];
            q[0][b] = f4add(f4add(m0, m1), m2);
            q[1][b] = f4sub(f4sub(m1, m2), m3);
        }
        const float4 bv = bias ? bias[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float4 o[2];
            o[0] = f4add(f4add(q[a][0], q[a][1]), q[a][2]);
            o[1] = f4sub(f4sub(q[a][1], q[a][2]), q[a][3]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const long long off = (((long long)n * H + 2 * ty + a) * W + 2 * tx + b) * C4 + c;
                float4 v = f4add(o[b], bv);
                if (addend) v = f4add(v, addend[off]);
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (mask) {
                    const float4 k = mask[off];
                    v.x = k.x > 0.f ? v.x * mask_scale : 0.f; v.y = k.y > 0.f ? v.y * mask_scale : 0.f;
                    v.z = k.z > 0.f ? v.z * mask_scale : 0.f; v.w = k.w > 0.f ? v.w * mask_scale : 0.f;
                }
                y[off] = v;
            }
        }
    }
}
void launch_wino_output(const float* m, const float* bias, const float* addend, const float* mask, float mask_scale,
                        int relu, float* y, int N, int H, int W, int C, hipStream_t s)
{
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3(wcap(total)), dim3(256), 0, s, (const float4*)m, (const float4*)bias,
                       (const float4*)addend, (const float4*)mask, mask_scale, relu, (float4*)y, N, H, W, C / 4);
}

// ---- weight gradient in the Winograd domain --------------------------------------------------------------------
//   Y = A^T M A  =>  dM = A dY A^T (4x4 from the 2x2 output-gradient tile);   dU[xi] = V[xi]^T dM[xi] (16 batched GEMMs
//   over the tiles, V = the input transform kept from the forward pass);   U = G g G^T  =>  dg = G^T dU G.
__global__ __launch_bounds__(256) void wino_dout_kernel(const float4* dy, float4* dm, int N, int H, int W, int C4)
{
    const int th = H / 2, tw = W / 2;
    const long long T = (long long)N * th * tw, total = T * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const long long t = i / C4;
        const int tx = (int)(t % tw); const long long r = t / tw;
        const int ty = (int)(r % th); const int n = (int)(r / th);
        const long long base = (((long long)n * H + 2 * ty) * W + 2 * tx) * C4 + c;
        const float4 y00 = dy[base], y01 = dy[base + C4], y10 = dy[base + (long long)W * C4], y11 = dy[base + (long long)W * C4 + C4];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q[4][2];                            // q = A dY,  A = [1 0; 1 1; 1 -1; 0 -1]
        q[0][0] = y00;             q[0][1] = y01;
        q[1][0] = f4add(y00, y10); q[1][1] = f4add(y01, y11);
        q[2][0] = f4sub(y00, y10); q[2][1] = f4sub(y01, y11);
        q[3][0] = f4sub(z, y10);   q[3][1] = f4sub(z, y11);
#pragma unroll
        for (int a = 0; a < 4; ++a) {              // dM = q A^T
            dm[((long long)(a * 4 + 0) * T + t) * C4 + c] = q[a][0];
            dm[((long long)(a * 4 + 1) * T + t) * C4 + c] = f4add(q[a][0], q[a][1]);
            dm[((long long)(a * 4 + 2) * T + t) * C4 + c] = f4sub(q[a][0], q[a][1]);
            dm[((long long)(a * 4 + 3) * T + t) * C4 + c] = f4sub(z, q[a][1]);
        }
    }
}
void launch_wino_dout(const float* dy, float* dm, int N, int H, int W, int C, hipStream_t s)
{
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(wino_dout_kernel, dim3(wcap(total)), dim3(256), 0, s, (const float4*)dy, (float4*)dm, N, H, W, C / 4);
}

__global__ void wino_dfilter_kernel(const float* du, float* dw, int Cin, int Cout)
{
    const long long cc = (long long)Cin * Cout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < cc; i += (long long)gridDim.x * blockDim.x) {
        float u[4][4], t[3][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) u[a][b] = du[(a * 4 + b) * cc + i];
#pragma unroll
        for (int b = 0; b < 4; ++b) {              // t = G^T dU
            t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
            t[1][b] = 0.5f * (u[1][b] - u[2][b]);
            t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {              // dg = t G
            dw[(a * 3 + 0) * cc + i] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
            dw[(a * 3 + 1) * cc + i] = 0.5f * (t[a][1] - t[a][2]);
            dw[(a * 3 + 2) * cc + i] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
        }
    }
}
void launch_wino_dfilter(const float* du, float* dw, int Cin, int Cout, hipStream_t s)
{
    hipLaunchKernelGGL(wino_dfilter_kernel, dim3(wcap((long long)Cin * Cout)), dim3(256), 0, s, du, dw, Cin, Cout);
}

}  // namespace fcn8s
