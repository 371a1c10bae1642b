// The three 1x1 score heads of the decoder (fcn8s_tensorflow.py:171-200): GEMMs with one skinny dimension (C = num_classes
// <= 32) and a long one (256 / 512 / 4096 channels).  They move 134 + 67 + 134 MB per pass at bs16 and do almost no
// arithmetic, so the general 128-row tile kernels are latency-bound on them (10-18 TFLOP/s, 0.76 ms per training step, and
// 0.30 of the 2.95 ms bs1 inference).  Here one wave owns a 32 x 32 v_mfma_f32_32x32x2_f32 accumulator, operands go from
// global memory straight into the MFMA registers (x / dy / dx rows are touched exactly once, the 80 KB..320 KB weight stays
// in L1/L2), and the reduction dimension is split over the waves of a block (forward, weight gradient) so that even the
// 512-row bs1 problem fills the chip.  Summation order is fixed (LDS reduction in wave order) in the forward kernel.
#include "fcn8s_internal.h"

namespace fcn8s {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Accumulator element i of lane l is D[row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5)][col = l & 31].

// y[r][c] = alpha * sum_k x[r][k] w[k][c] + bias[c];  NW waves split K.
// Lane half h supplies k = kk + 4h + j to MFMA step j (same for both operands).  The loads of the next 32-deep slab are in
// flight while the current one is multiplied (two register stages).
template <int NW, int C>
__global__ __launch_bounds__(NW * 64) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, long long M, int K, float alpha, float* __restrict__ part)
{
    // part (few rows, long K -- fc7's head at batch 1: 16 row blocks on 256 CUs): gridDim.y blocks split K once more; each writes its
    // alpha-scaled partial sums to part[blockIdx.y][M][C], head_fwd_reduce_kernel adds them in a fixed order and the bias
    __shared__ float red[NW][16 * 2 * 20 + 16 * 2 * 12 * (NW <= 8)];      // [wave][i][h][col]: 32 columns for <= 8 waves, 20 (C <= 20) for 16
    constexpr int RC = NW <= 8 ? 32 : 20;
    __shared__ __attribute__((aligned(16))) float outt[32 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long long row0 = (long long)blockIdx.x * 32;
    long long row = row0 + r;
    if (row >= M) row = M - 1;                                  // rows / columns past the edge compute garbage that is never stored
    const int Kb = K / (int)gridDim.y, k0 = (int)blockIdx.y * Kb, ks = Kb / NW;
    const float* xp = x + row * K + k0 + wave * ks + 4 * h;
    const float* wp = w + (long long)(k0 + wave * ks + 4 * h) * C + (r < C ? r : C - 1);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load = [&](int kk, float4 (&a)[4], float (&b)[4][4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(xp + kk + 8 * u);
            const float* wq = wp + (long long)(kk + 8 * u) * C;
            b[u][0] = wq[0]; b[u][1] = wq[C]; b[u][2] = wq[2 * C]; b[u][3] = wq[3 * C];
        }
    };
    auto mul = [&](const float4 (&a)[4], const float (&b)[4][4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u][3], acc, 0, 0, 0);
        }
    };
    float4 a0[4], a1[4]; float b0[4][4], b1[4][4];
    const int kfull = ks & ~31;
    if (kfull) load(0, a0, b0);
#pragma unroll 1
    for (int kk = 0; kk < kfull; kk += 64) {
        if (kk + 32 < kfull) load(kk + 32, a1, b1);
        mul(a0, b0);
        if (kk + 32 < kfull) {
            if (kk + 64 < kfull) load(kk + 64, a0, b0);
            mul(a1, b1);
        }
    }
    for (int kk = kfull; kk < ks; kk += 8) {
        const float4 a = *reinterpret_cast<const float4*>(xp + kk);
        const float* wq = wp + (long long)kk * C;
        const float c0 = wq[0], c1 = wq[C], c2 = wq[2 * C], c3 = wq[3 * C];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, c0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, c1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, c2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, c3, acc, 0, 0, 0);
    }
    if (r < RC) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][(i * 2 + h) * RC + r] = acc[i];
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * RC; idx += NW * 64) {       // idx = (i * 2 + h) * RC + col
        float s = red[0][idx];
#pragma unroll
        for (int v = 1; v < NW; ++v) s += red[v][idx];
        const int ih = idx / RC, ocol = idx - ih * RC;
        const int i = ih >> 1, hh = ih & 1;
        const int orow = (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (ocol < C) outt[orow * C + ocol] = s * alpha + ((bias && !part) ? bias[ocol] : 0.f);
    }
    __syncthreads();
    const long long left = M - row0;
    const int n4 = (int)(left < 32 ? left : 32) * C / 4;        // the block's 32 rows x C floats are contiguous in y
    float4* dst = reinterpret_cast<float4*>((part ? part + (long long)blockIdx.y * M * C : y) + row0 * C);
    for (int idx = tid; idx < n4; idx += NW * 64) dst[idx] = reinterpret_cast<const float4*>(outt)[idx];
}

// dx[p][ch] = mask(alpha * sum_c dy[p][c] wt[c][ch]);  wt = [C][K].  MFMA rows = 32 pixels, columns = 32 channels: accumulator
// register i of a lane is one pixel row, the 32 lanes of a half are 32 consecutive channels -- every store (and mask load)
// instruction covers two full 128-byte lines.  (Channels as MFMA rows would give 16-byte stores, but 32 partial lines per
// instruction: twice the address-coalescer cycles and partial-line writes.)  One wave walks `tpw` channel tiles of its
// pixel group; the next tile's weights and this tile's mask are in flight during the MFMAs.
template <int C>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wt, const float* __restrict__ mask,
                                                         float* __restrict__ dx, long long M, int K, float alpha, float mask_scale, int tpw, int nct)
{
    constexpr int CH = C / 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long gw = (long long)blockIdx.x * 4 + wave;
    const long long pg = gw / nct;
    const int cc = (int)(gw - pg * nct);
    if (pg * 32 >= M) return;
    const int n = lane & 31, h = lane >> 5;
    const long long pa = pg * 32 + n < M ? pg * 32 + n : M - 1;
    float a[CH];
#pragma unroll
    for (int j = 0; j < CH; j += 2) { const float2 t = *reinterpret_cast<const float2*>(dy + pa * C + h * CH + j); a[j] = t.x; a[j + 1] = t.y; }
    const int t0 = cc * tpw, t1 = (cc + 1) * tpw < K / 32 ? (cc + 1) * tpw : K / 32;
    const float* wl = wt + (long long)h * CH * K + n;
    const long long prow0 = pg * 32 + 4 * h;                    // register i is pixel prow0 + 8 * (i >> 2) + (i & 3)
    const int nvalid = (int)(M - prow0 < 32 ? M - prow0 : 32);  // (rows of this lane half that exist: 8 * (i >> 2) + (i & 3) < nvalid)
    float b[CH], bn[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) b[j] = wl[(long long)j * K + t0 * 32];
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        const int ch0 = t * 32;
        const long long off0 = prow0 * K + ch0 + n;
        float mk[16];
        if (mask) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int pr = 8 * (i >> 2) + (i & 3); mk[i] = pr < nvalid ? mask[off0 + (long long)pr * K] : 0.f; }
        }
        if (t + 1 < t1) {
#pragma unroll
            for (int j = 0; j < CH; ++j) bn[j] = wl[(long long)j * K + ch0 + 32];
        }
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < CH; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int pr = 8 * (i >> 2) + (i & 3);
            if (pr >= nvalid) continue;
            float v = acc[i] * alpha;
            if (mask) v = mk[i] > 0.f ? v * mask_scale : 0.f;
            dx[off0 + (long long)pr * K] = v;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) b[j] = bn[j];
    }
}

// dw[ch][c] += alpha * sum_p x[p][ch] dy[p][c].  MFMA rows = channels, columns = classes, reduction = the block's pixel range,
// a quarter per wave.  A lane loads four consecutive channels of a pixel (16 bytes; a wave instruction covers two pixels x 512
// contiguous bytes) and feeds four accumulators: MFMA row m of accumulator t is channel ch0 + 4m + t.  Two register stages:
// the next eight pixels' loads are in flight during this batch's 32 MFMAs (16 KB per wave -- the launcher cannot buy
// parallelism with more blocks, because every block adding into the same dw tile costs ~0.1 us of serialised device-scope
// atomics).  The four waves' partial tiles meet in LDS, then one global atomic per element.
// (Tried: 16-wave blocks with ds_add_f32 into one LDS tile -- 4x slower, LDS float atomics run a few lanes per clock.)
template <int C>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                         long long M, int K, float alpha, int ppb, long long split_stride)
{
    dw += (long long)blockIdx.y * split_stride;                 // (deterministic mode: a zeroed slab per pixel range, summed in range order afterwards)
    __shared__ float red[4][4 * 32 * C];                        // [wave][t][(i * 2 + h) * C + class]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    // channel group = blockIdx.x (fastest): the blocks in flight together read whole rows of x rather than the same 512-byte
    // window of every row (measured: the same time either way)
    const int ch0 = blockIdx.x * 128;
    const int half = ppb / 8;                                   // pixels per (wave, lane half); a multiple of 8 (launcher)
    const long long p0 = (long long)blockIdx.y * ppb + (wave * 2 + h) * half;
    const float* xp = x + ch0 + 4 * m;
    const float* dp = dy + (m < C ? m : C - 1);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load = [&](int j0, float4 (&a)[8], float (&b)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long p = p0 + j0 + u;
            const bool ok = p < M;
            const long long pc = ok ? p : M - 1;
            a[u] = *reinterpret_cast<const float4*>(xp + pc * K);
            b[u] = ok ? dp[pc * C] : 0.f;                       // a zero class row contributes nothing whatever x holds
        }
    };
    auto mul = [&](const float4 (&a)[8], const float (&b)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u], acc[3], 0, 0, 0);
        }
    };
    float4 a0[8], a1[8]; float b0[8], b1[8];
    load(0, a0, b0);
#pragma unroll 1
    for (int j0 = 0; j0 < half; j0 += 16) {
        if (j0 + 8 < half) load(j0 + 8, a1, b1);
        mul(a0, b0);
        if (j0 + 8 < half) {
            if (j0 + 16 < half) load(j0 + 16, a0, b0);
            mul(a1, b1);
        }
    }
    if (m < C) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[wave][t * 32 * C + (i * 2 + h) * C + m] = acc[t][i];
    }
    __syncthreads();
    for (int idx = tid; idx < 4 * 32 * C; idx += 256) {
        const int t = idx / (32 * C), rem = idx - t * 32 * C;
        const int ih = rem / C, ocol = rem - ih * C;
        const int i = ih >> 1, hh = ih & 1;
        const int orow = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float v = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
        unsafeAtomicAdd(dw + (long long)(ch0 + 4 * orow + t) * C + ocol, v * alpha);
    }
}

__global__ __launch_bounds__(256) void head_fwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, long long n, int C, int splits)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int k = 1; k < splits; ++k) s += part[(long long)k * n + i];
    y[i] = s + (bias ? bias[i % C] : 0.f);
}
template <int C>
static bool launch_head_fwd_c(const float* x, const float* w, const float* bias, float* y, long long M, int K, float alpha, hipStream_t s, float* scratch, size_t scratch_floats)
{
    const unsigned blocks = (unsigned)((M + 31) / 32);
    // few row blocks and a long reduction: split K over 8 blocks as well (deterministic two-pass sum)
    constexpr int SPLITS = 8;
    if (blocks <= 64 && K % (SPLITS * 16 * 32) == 0 && scratch && scratch_floats >= (size_t)SPLITS * M * C && (M * C) % 4 == 0) {
        hipLaunchKernelGGL((head_fwd_kernel<16, C>), dim3(blocks, SPLITS), dim3(1024), 0, s, x, w, bias, y, M, K, alpha, scratch);
        hipLaunchKernelGGL(head_fwd_reduce_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, s, scratch, bias, y, M * C, C, SPLITS);
        return true;
    }
    if (K >= 2048 && K % 128 == 0) hipLaunchKernelGGL((head_fwd_kernel<16, C>), dim3(blocks), dim3(1024), 0, s, x, w, bias, y, M, K, alpha, (float*)nullptr);
    else if (K >= 2048 && K % 64 == 0) hipLaunchKernelGGL((head_fwd_kernel<8, C>), dim3(blocks), dim3(512), 0, s, x, w, bias, y, M, K, alpha, (float*)nullptr);
    else if (K % 32 == 0) hipLaunchKernelGGL((head_fwd_kernel<4, C>), dim3(blocks), dim3(256), 0, s, x, w, bias, y, M, K, alpha, (float*)nullptr);
    else return false;
    return true;
}
bool launch_head_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int K, int C, float alpha, hipStream_t s, float* scratch, size_t scratch_floats)
{
    if (M < 1) return false;
    if (C == 20) return launch_head_fwd_c<20>(x, w, bias, y, M, K, alpha, s, scratch, scratch_floats);
    if (C == 4) return launch_head_fwd_c<4>(x, w, bias, y, M, K, alpha, s, scratch, scratch_floats);
    return false;
}

bool launch_head_dgrad(const float* dy, const float* wt, const float* mask, float mask_scale, float* dx, long long M, int K, int C, float alpha,
                       hipStream_t s)
{
    if ((C != 20 && C != 4) || K % 32 || M < 1) return false;
    const int tiles = K / 32, tpw = tiles < 8 ? tiles : 8, nct = (tiles + tpw - 1) / tpw;
    const long long waves = ((M + 31) / 32) * nct;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (C == 20) hipLaunchKernelGGL(head_dgrad_kernel<20>, dim3(blocks), dim3(256), 0, s, dy, wt, mask, dx, M, K, alpha, mask_scale, tpw, nct);
    else hipLaunchKernelGGL(head_dgrad_kernel<4>, dim3(blocks), dim3(256), 0, s, dy, wt, mask, dx, M, K, alpha, mask_scale, tpw, nct);
    return true;
}

bool launch_head_wgrad(const float* x, const float* dy, float* dw, long long M, int K, int C, float alpha, hipStream_t s)
{
    if ((C != 20 && C != 4) || K % 128 || M < 1) return false;
    const int nct = K / 128;
    constexpr int bxmax = 128;        // blocks that add into the same dw tile: same-address float atomics serialise (512 blocks: 71 us, 128: 41)
    long long bx = 1024 / nct; if (bx < 1) bx = 1;
    if (bx > bxmax) bx = bxmax;
    if (bx > (M + 63) / 64) bx = (M + 63) / 64;
    int ppb = (int)((M + bx - 1) / bx);
    ppb = (ppb + 63) / 64 * 64;                                 // 8 pixel runs (4 waves x 2 lane halves), each a multiple of the 8-deep load batch
    bx = (M + ppb - 1) / ppb;
    float* out = dw; long long stride = 0;
    if (t_deterministic && bx > 1) {                            // the bx pixel ranges of a channel group meet in atomics: one slab each instead
        stride = (long long)K * C;
        out = det_scratch(s, (size_t)(bx * stride));
        if (!out) { defer_error(FCN8S_ERR_OOM, "deterministic mode: the score heads' weight-gradient scratch (%lld floats) cannot be allocated", (long long)bx * stride); return true; }
        hipMemsetAsync(out, 0, (size_t)(bx * stride) * sizeof(float), s);
    }
    if (C == 20) hipLaunchKernelGGL(head_wgrad_kernel<20>, dim3(nct, (unsigned)bx), dim3(256), 0, s, x, dy, out, M, K, alpha, ppb, stride);
    else hipLaunchKernelGGL(head_wgrad_kernel<4>, dim3(nct, (unsigned)bx), dim3(256), 0, s, x, dy, out, M, K, alpha, ppb, stride);
    if (stride) launch_det_reduce(dw, out, 1, (int)stride, (int)stride, stride, (int)bx, true, s);
    return true;
}

}  // namespace fcn8s
