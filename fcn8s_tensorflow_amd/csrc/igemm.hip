// Implicit-GEMM convolution kernels for gfx950 on the exact-f32 matrix core
// instruction v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD = the fp32 vector peak,
// bit-for-bit an fmaf chain).  Two kernels carry 99.6 % of FCN-8s' FLOPs:
//
//   igemm_fwd_kernel   : forward conv / data gradient / transposed conv (phases)
//   wgrad_kernel       : weight gradient (reduction over pixels, split-K + atomics)
//
// Data layout: NHWC activations, [tap][Cin][Cout] weights, so that
//   * the A operand (pixels x channels) is staged into LDS as [row][k] with k
//     contiguous (16-B global loads along channels, ds_read_b128 per lane row),
//   * the B operand (k x cout) is staged as [k][col] (ds_read_b32, lanes = columns).
// Block = 256 threads = 4 wave64; K is consumed 16 at a time (8 MFMA k-steps),
// global->register->LDS double-buffered with one barrier per K-tile.
#include "fcn8s_internal.h"
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace fcn8s {

thread_local const char* g_last_kernel = nullptr;
#define FCN8S_STR2(x) #x
#define FCN8S_STR(x) FCN8S_STR2(x)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK_ = 16;

static __device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// XCD-aware block order.  The dispatcher places workgroup p on XCD p % 8 (observed, speed only -- results
// never depend on it); each XCD has a private 4 MiB L2.  Give every XCD one CONTIGUOUS run of logical tile
// ids, so that tiles which share operand panels (the N-tiles of one M-tile; spatially adjacent M-tiles whose
// halos overlap) run concurrently behind the same L2 instead of being dealt round-robin across all eight.
// Bijective for any total (guide section 5.5 T1).
static __device__ __forceinline__ unsigned xcd_swizzle(unsigned p, unsigned total)
{
    const unsigned q = total >> 3, r = total & 7u, xcd = p & 7u, i = p >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

thread_local int t_deterministic = 0;
// Scratch for the slab reductions, the column-sum partials and the split-K slabs: one buffer per (device, stream) -- launches on one stream are ordered, two streams
// never share a buffer, and a stream handle (the NULL stream above all) means a different queue on every device.  Grown on demand; released by
// scratch_release when the model that owns the stream is destroyed (round 5 kept every buffer for the life of the process, keyed by the handle alone: a handle
// reused after fcn8s_destroy, or the NULL stream on a second device, got memory of the wrong device -- ADVICE round 5).
namespace {
struct ScratchBuf { float* p = nullptr; size_t cap = 0; };
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, ScratchBuf> g_scratch[2];      // [0] det_scratch, [1] scratch2 (column sums)
float* scratch_get(int which, hipStream_t s, size_t floats, size_t floor_)
{
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchBuf& b = g_scratch[which][std::make_pair(dev, s)];
    if (b.cap < floats) {
        if (b.p) hipFree(b.p);                     // (synchronises the device: nothing still reads the old buffer)
        b.p = nullptr; b.cap = 0;
        size_t want = floats + floats / 4 + (1u << 20);
        if (want < floor_) want = floor_;
        if (hipMalloc((void**)&b.p, want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        b.cap = want;
    }
    return b.p;
}
}
float* det_scratch(hipStream_t s, size_t floats) { return scratch_get(0, s, floats, 0); }
float* scratch2(hipStream_t s, size_t floats) { return scratch_get(1, s, floats, 1u << 18); }
// frees what the calling device's stream `s` holds (the caller has synchronised it)
void scratch_release(hipStream_t s)
{
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto& pool : g_scratch) {
        auto it = pool.find(std::make_pair(dev, s));
        if (it != pool.end()) { if (it->second.p) hipFree(it->second.p); pool.erase(it); }
    }
}
__global__ __launch_bounds__(256) void det_reduce_kernel(float* __restrict__ C, const float* __restrict__ ws, const long long n, const int cols, const int ldc,
                                                         const long long slab, const int nsplit, const int accumulate)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long off = (cols == ldc) ? i : (i / cols) * ldc + (i % cols);
        float v = accumulate ? C[off] : 0.f;
        const float* w = ws + off;
        for (int k = 0; k < nsplit; ++k) v += w[(long long)k * slab];          // fixed order: split 0, 1, 2, ...
        C[off] = v;
    }
}
void launch_det_reduce(float* C, const float* ws, long long rows, int cols, int ldc, long long slab, int nsplit, bool accumulate, hipStream_t s)
{
    const long long n = rows * cols;
    if (n <= 0) return;
    long long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, C, ws, n, cols, ldc, slab, nsplit, accumulate ? 1 : 0);
}
// ===========================================================================
// forward / dgrad / transposed-conv implicit GEMM
// ===========================================================================
// FAST = (Cin % 16 == 0, Cout % BN == 0, one phase): no K / N predicates, the (tap, ci) position of
// the next K-tile and the per-row source pointers are advanced incrementally (no integer division in
// the loop) and every global load is unconditional (out-of-image rows read a safe address and are
// zeroed by a select), which keeps the loop free of exec-mask branches.
static __device__ __attribute__((noinline)) float dropout_apply(float v, unsigned long long idx, unsigned long long seed,
                                                               unsigned int stream, float keep)
{
    return philox_uniform(idx, seed, stream) < keep ? v / keep : 0.f;
}

template <int BM, int BN, int WM, int WN, int MODE, int BK>
__global__ __launch_bounds__(256, (MODE != 0 && BM * BN <= 128 * 128) ? 4 : 1) void igemm_fwd_kernel(const IgemmArgs p)
{
    constexpr bool FAST = MODE != 0;
    // MODE 3 = plain batched GEMM (the Winograd positions): row m of slab z is row m of X / Y, no taps, no epilogue
    // ops -- the generic row -> pixel prologue (integer divisions) and the per-element epilogue branches cost a
    // K = 256 GEMM more than a tenth of its time.
    constexpr bool PLAIN = MODE == 3;
    constexpr int LDA = BK + 4, LDB = BN, F4R = BK / 4;     // F4R float4 per A row; LDA*4 B row stride keeps ds_read_b128 conflict-free
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LD = BM * F4R / 256;
    constexpr int B_F4 = BK * BN / 4;            // float4 per B tile
    constexpr int B_LD = (B_F4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDA + 2 * BK * LDB];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int pz = blockIdx.z;
    // gridDim.z: transposed-conv sub-pixel phases (weights + output offset per phase) or, when p.batched,
    // independent GEMMs with their own input / weight / output slabs (the 16 Winograd positions)
    const int pzy = p.batched ? 0 : pz / p.phases_x, pzx = p.batched ? 0 : pz - pzy * p.phases_x;
    const float* __restrict__ Wp = p.w + (long long)pz * p.w_phase_stride;
    const float* __restrict__ X = p.batched ? p.x + (long long)pz * p.x_batch_stride : p.x;
    float* __restrict__ Y = p.batched ? p.y + (long long)pz * p.y_batch_stride : p.y;
    const int offy = p.out_offy + pzy, offx = p.out_offx + pzx;
    const unsigned ntn = (unsigned)((p.Cout + BN - 1) / BN);
    // 1-D grid of M-tiles x N-tiles, N fastest: the N-tiles of one M-tile (same A panel) and the next
    // M-tiles (overlapping halos) are neighbours in the XCD's contiguous run of logical ids.  (M-fastest and
    // 32-M-tile panels were measured for the streamed fc6 filter bank: same time, same fabric traffic.)
    const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
    // p.m_fastest (launcher): when the B operand is the big one (fc6: a 33 MB filter slab per Winograd position against 4 MB of
    // activations), neighbouring ids share the B panel instead, so that each panel is fetched by ONE XCD's L2 and not by every
    // XCD that holds one of its M-tiles (rocprof FETCH_SIZE of the fc6 GEMMs was 2-3x their algorithmic bytes with N fastest).
    const unsigned ntm = gridDim.x / ntn;
    const long long m0 = (long long)(p.m_fastest ? lid % ntm : lid / ntn) * BM;
    const int n0 = (int)(p.m_fastest ? lid / ntm : lid % ntn) * BN;
    const int MaMb = p.Ma * p.Mb;

    long long a_base[A_LD];
    int a_iy[A_LD], a_ix[A_LD];
    bool a_ok[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int row = (tid + i * 256) / F4R;
        const long long m = m0 + row;
        a_ok[i] = m < p.M;
        if (PLAIN) { a_iy[i] = a_ix[i] = 0; a_base[i] = a_ok[i] ? m : 0; continue; }
        const long long mm = a_ok[i] ? m : 0;
        const int n = (int)(mm / MaMb);
        const int r = (int)(mm - (long long)n * MaMb);
        const int a = r / p.Mb, b = r - a * p.Mb;
        a_iy[i] = a * p.in_scale; a_ix[i] = b * p.in_scale;
        a_base[i] = (long long)n * p.Hi * p.Wi;
    }
    const int a_c4 = (tid % F4R) * 4;

    float4 ra[A_LD], rb[B_LD];
    // ---- FAST path state: position of the next tile to load.
    // MODE 2 (tap inner): K order = channel chunk OUTER, tap INNER -- the k*k taps of one 16-channel slice
    //   re-read (shifted) the same few KB per block, which stay in L1/L2, instead of sweeping the whole halo
    //   once per tap (that order missed L2 on every tap: rocprof FETCH_SIZE 4-6 GB per launch against
    //   ~0.5 GB algorithmic).  Used when the filter bank is small enough to stay cached (3x3 layers).
    // MODE 1 (tap outer): huge filter banks (fc6: 411 MB) are streamed contiguously instead.
    constexpr bool TAP_INNER = MODE == 2;
    int f_ci0 = 0, f_ty = 0, f_tx = 0, f_tap = 0;
    const int ntaps = FAST ? p.Ktot / p.Cin : 1;
    const float* a_ptr[A_LD];                // MODE 1: row pointer for the current tap; MODE 2: tap-offset-0 pointer
    unsigned a_mask[A_LD];                   // MODE 2: bit t set = tap t of this row lies inside the image (<= 32 taps)
    bool a_val[A_LD], a_ldok[A_LD];          // a_ldok: validity of the rows of the tile currently held in ra[]
    const float* b_ptr[B_LD];
    auto set_tap = [&]() {                   // MODE 1 (and 3: one tap, pixel = row)
        if (PLAIN) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) { a_val[i] = a_ok[i]; a_ptr[i] = X + a_base[i] * p.ldx + a_c4; }
            return;
        }
        const int dy = f_ty * p.tap_step + p.tap_off, dx = f_tx * p.tap_step + p.tap_off;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            a_val[i] = a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const long long pix = a_val[i] ? a_base[i] + (long long)iy * p.Wi + ix : 0;
            a_ptr[i] = X + pix * p.ldx + a_c4;
        }
    };
    if (FAST) {
        if (TAP_INNER) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                unsigned mk = 0;
                for (int t = 0; t < ntaps; ++t) {
                    const int ty = t / p.KW, tx = t - ty * p.KW;
                    const int iy = a_iy[i] + ty * p.tap_step + p.tap_off, ix = a_ix[i] + tx * p.tap_step + p.tap_off;
                    if (a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mk |= 1u << t;
                }
                a_mask[i] = mk;
                a_ptr[i] = X + (a_base[i] + (long long)a_iy[i] * p.Wi + a_ix[i]) * p.ldx + a_c4;
            }
        } else set_tap();
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            const int k = f / (BN / 4), j = (f - k * (BN / 4)) * 4;
            b_ptr[i] = Wp + (long long)k * p.Cout + n0 + j;
        }
    }
    auto gload_fast = [&]() {
        if (TAP_INNER) {
            // block-uniform tap offsets (scalar registers)
            const long long delta = ((long long)(f_ty * p.tap_step + p.tap_off) * p.Wi + (f_tx * p.tap_step + p.tap_off)) * p.ldx + f_ci0;
            const long long boff = ((long long)f_tap * p.Cin + f_ci0) * p.Cout;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const bool ok = (a_mask[i] >> f_tap) & 1u;
                ra[i] = ldg4(ok ? a_ptr[i] + delta : X);  // zeroing of out-of-image rows is deferred to sstore so
                a_ldok[i] = ok;                              // that the wave does not wait for the load before its MFMAs
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                if (B_F4 % 256 == 0 || tid + i * 256 < B_F4) rb[i] = ldg4(b_ptr[i] + boff);
            ++f_tap;
            if (++f_tx == p.KW) { f_tx = 0; ++f_ty; }
            if (f_tap == ntaps) { f_tap = 0; f_tx = 0; f_ty = 0; f_ci0 += BK; }
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                ra[i] = ldg4(a_ptr[i] + f_ci0);
                a_ldok[i] = a_val[i];
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                if (B_F4 % 256 == 0 || tid + i * 256 < B_F4) rb[i] = ldg4(b_ptr[i]);
                b_ptr[i] += (long long)BK * p.Cout;
            }
            f_ci0 += BK;
            if (f_ci0 == p.Cin) {
                f_ci0 = 0;
                if (++f_tx == p.KW) { f_tx = 0; ++f_ty; }
                set_tap();           // one tap past the end computes pointers that are never dereferenced
            }
        }
    };
    auto gload = [&](int kt) {
        if (FAST) { gload_fast(); return; }
        const int kg = kt * BK + a_c4;
        const bool kok = kg < p.Ktot;
        const int tap = kg / p.Cin, ci = kg - tap * p.Cin;
        const int ty = tap / p.KW, tx = tap - ty * p.KW;
        const int dy = ty * p.tap_step + p.tap_off, dx = tx * p.tap_step + p.tap_off;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            const bool ok = a_ok[i] && kok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            ra[i] = ok ? ldg4(X + (a_base[i] + (long long)iy * p.Wi + ix) * p.ldx + ci)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            const int k = f / (BN / 4), j = (f - k * (BN / 4)) * 4;
            const int kg2 = kt * BK + k, col = n0 + j;
            const bool ok = (B_F4 % 256 == 0 || f < B_F4) && kg2 < p.Ktot && col < p.Cout;
            rb[i] = ok ? ldg4(Wp + (long long)kg2 * p.Cout + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int row = (tid + i * 256) / F4R;
            if (FAST && !a_ldok[i]) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&As[buf * BM * LDA + row * LDA + a_c4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            if (B_F4 % 256 == 0 || f < B_F4) {
                const int k = f / (BN / 4), j = (f - k * (BN / 4)) * 4;
                *reinterpret_cast<float4*>(&Bs[buf * BK * LDB + k * LDB + j]) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The global loads of the NEXT K-tile are issued in the middle of this tile's MFMAs (after the first half has been handed to
    // the matrix pipe), not ahead of its LDS fragment reads: 131 vs 120 TFLOP/s on a bare main-loop microbenchmark at K = 512, +1-1.5 % in this kernel.
    auto compute = [&](int buf, int next_kt) {
        const float* A = As + buf * BM * LDA + (wm * TM * 32 + (lane & 31)) * LDA + (lane >> 5) * 4;
        const float* B = Bs + buf * BK * LDB + ((lane >> 5) * 4) * LDB + wn * TN * 32 + (lane & 31);
#pragma unroll
        for (int kk2 = 0; kk2 < BK / 8; ++kk2) {
            if (kk2 == BK / 16 && next_kt >= 0) gload(next_kt);
            float4 af[TM];
            float bf[TN][4];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[tm] = *reinterpret_cast<const float4*>(A + tm * 32 * LDA + kk2 * 8);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[tn][j] = B[(kk2 * 8 + j) * LDB + tn * 32];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const float av = j == 0 ? af[tm].x : j == 1 ? af[tm].y : j == 2 ? af[tm].z : af[tm].w;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[tn][j], acc[tm][tn], 0, 0, 0);
                }
            }
        }
    };

    // split-K (gridDim.y > 1, MODE 1 only, linear epilogue): this block reduces K-tiles [kt0, kt0 + nkt)
    const int nkt_all = (p.Ktot + BK - 1) / BK;
    const int kt0 = (int)((long long)nkt_all * blockIdx.y / gridDim.y);
    const int nkt = (int)((long long)nkt_all * (blockIdx.y + 1) / gridDim.y) - kt0;
    if ((MODE == 1 || MODE == 3) && kt0 > 0) {
        const int tap0 = kt0 * BK / p.Cin;
        f_ci0 = kt0 * BK - tap0 * p.Cin;
        f_ty = tap0 / p.KW; f_tx = tap0 - f_ty * p.KW;
        set_tap();
#pragma unroll
        for (int i = 0; i < B_LD; ++i) b_ptr[i] += (long long)kt0 * BK * p.Cout;
    }
    gload(kt0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        compute(cur, kt + 1 < nkt ? kt0 + kt + 1 : -1);
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    if (PLAIN) {
        // each wave transposes its 32x32 accumulator tiles through a private LDS patch so that the global stores are
        // 16 bytes per lane (8 rows x 128 B per instruction) instead of 64 dword stores per tile
        if (gridDim.y > 1) {                     // split-K partials
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (m < p.M) unsafeAtomicAdd(Y + m * p.ldy + n0 + wn * TN * 32 + tn * 32 + (lane & 31), acc[tm][tn][r]);
                    }
            return;
        }
        constexpr int LDT = 36;
        float* patch = smem + wave * 32 * LDT;           // 4 x 4.6 KB, inside the (now idle) A buffers
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + (lane & 31)] = acc[tm][tn][r];
                __builtin_amdgcn_wave_barrier();
                const long long mrow = m0 + wm * TM * 32 + tm * 32;
                float* yb = Y + n0 + wn * TN * 32 + tn * 32 + (lane & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = j * 8 + (lane >> 3);
                    const float4 v = *reinterpret_cast<const float4*>(&patch[row * LDT + (lane & 7) * 4]);
                    if (mrow + row < p.M) *reinterpret_cast<float4*>(yb + (mrow + row) * p.ldy) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }
    // ---- epilogue: row -> output offset table in LDS, then fused bias/add/relu/mask/dropout
    long long* rowoff = reinterpret_cast<long long*>(smem);
    if (tid < BM) {
        const long long m = m0 + tid;
        long long off = -1;
        if (m < p.M) {
            const int n = (int)(m / MaMb);
            const int r = (int)(m - (long long)n * MaMb);
            const int a = r / p.Mb, b = r - a * p.Mb;
            const int oy = a * p.out_scale + offy, ox = b * p.out_scale + offx;
            if ((unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo)
                off = (((long long)n * p.Ho + oy) * p.Wo + ox) * p.ldy;
        }
        rowoff[tid] = off;
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
        if (col >= p.Cout) continue;
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long long off = rowoff[row];
                if (off < 0) continue;
                float v = acc[tm][tn][r] * p.alpha + bv;
                if (MODE == 1 && gridDim.y > 1) { unsafeAtomicAdd(Y + off + col, v); continue; }   // split-K partial
                if (p.addend) v += p.addend[off + col];
                if (p.relu) v = v > 0.f ? v : 0.f;
                if (p.mask) v = p.mask[off + col] > 0.f ? v * p.mask_scale : 0.f;
                if (p.dropout) v = dropout_apply(v, (unsigned long long)(off + col), p.seed, p.stream_id, p.keep_prob);
                Y[off + col] = v;
            }
        }
    }
}

// ===========================================================================
// plain GEMM rows (the Winograd positions, 1x1 convs): LDS-DMA main loop
// ===========================================================================
// Y[z] (M x Cout) = epilogue( X[z] (M x K, row stride ldx) * W[z] (K x Cout) ), the MODE 3 / 1x1 cases of igemm_fwd_kernel with a
// different way of filling LDS: global_load_lds_dwordx4 moves 64 lanes x 16 B from global memory straight into LDS (no VGPR
// staging, no ds_write, no vmcnt wait in front of a store pass), S = 3 stages, and one K-tile stays in flight ACROSS the K-tile
// barrier (counted vmcnt + raw s_barrier).  Measured on the bare batched GEMM (tools/labs/gemm_lab.hip, random data, MI355X):
// 125-129 / 132-134 / 138-141 TFLOP/s at K = 256 / 512 / 2048 against 116-122 / 121-125 / 126-129 for the register-staged loop.
//   * the LDS destination of one instruction is M0 + lane * 16 B (wave-uniform base, lane-linear): the A image therefore has
//     unpadded 64-byte rows, and ds_read_b128 stays conflict-free through an XOR swizzle of the 16-byte chunk index with
//     (row / 4) & 3 -- applied to the per-lane SOURCE address when loading and to the LDS address when reading (guide 5.4 rule 21);
//   * inline asm, because hipcc's waitcnt pass makes every ds_read wait vmcnt(0) for any LDS-DMA it knows about, which would
//     drain the pipeline once per K-tile; the asm loads are invisible to it and the counted waits are written by hand;
//   * saddr form (block-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset): no 64-bit VALU address math in the loop.
// Rows >= M read row M - 1 (their products land in accumulator rows that are never stored).
static __device__ __forceinline__ void glds16(const float* sbase, unsigned voff, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_off) : "memory", "m0");
}
template <int N> static __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- f32x3: fp32-accurate products on the bf16 matrix pipe (optional precision mode FCN8S_PREC_F32X3) --------------------------
// Every fp32 operand x is split exactly into three bf16 pieces x = hi + mid + lo (each residual is representable, so the split has
// no error) and a*b is taken as hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; the three dropped terms are below 2^-23 |a b|, i.e. below the rounding of one fp32 multiply.  Against a float64
// reference the result is as close as the f32 MFMA's (tools/labs/gemm_lab.hip x3: max error 6.4e-6 vs 7.5e-6 on K = 128).  Six bf16 MFMAs
// (32 cycles, K = 16) replace eight f32 MFMAs (64 cycles, K = 2): 2.7x less matrix-pipe time, of which the ~190 VALU instructions
// of the split take half back -- VALU and MFMA issue did not overlap in any arrangement tried (interleaved by hand or by
// sched_group_barrier, or software-pipelined over K-tiles): 155-170 "TFLOP/s" against 122-136.  Not the default: the headline
// number is measured on the exact f32 MFMA.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// (which arithmetic a launch uses travels in IgemmArgs::split / WgradArgs::split, set by the caller from its model's precision)
// f32x2 (FCN8S_PREC_F32X2 / FCN8S_PREC_BF16_FWD_X2): the same with two pieces, x ~ hi + lo (16 significand bits kept: the residual after lo is
// below 2^-17 |x|) and a*b taken as lo*hi + hi*lo + hi*hi -- three MFMAs and two conversions per value instead of six and three.  A
// reduced-precision arithmetic (between TF32's 11 bits and fp32's 24), never used by the fp32 mode.
template <int NS> struct Bf16Pieces { bf16x8 p[NS]; };          // p[0] = hi, p[1] = mid (or lo for NS = 2), p[2] = lo
// (two values at a time, in the shapes of v_cvt_pk_bf16_f32 / v_pk_add_f32: one packed conversion per pair and piece, the pieces
//  widened again by a shift / a mask of the packed word -- 2.5 VALU instructions per value for two pieces; element by element hipcc
//  spent 3.3.  Same conversions (RNE), same bits.)
template <int NS>
static __device__ __forceinline__ void split_bf16(const float (&x)[8], Bf16Pieces<NS>& o)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w[NS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 r = {x[2 * i], x[2 * i + 1]};
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
            w[k][i] = pk;
            if (k + 1 < NS) {
                const f32x2 back = {__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
                r = r - back;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) o.p[k] = __builtin_bit_cast(bf16x8, w[k]);
}
template <int NS>
static __device__ __forceinline__ f32x16 mfma_split(const Bf16Pieces<NS>& a, const Bf16Pieces<NS>& b, f32x16 acc)
{
    if constexpr (NS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], acc, 0, 0, 0);          // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
}

template <int BM, int BN, int WM, int WN, int S, bool EPI, int NSPLIT, bool BT = false>
static __device__ __forceinline__ void gemm_glds_body(const IgemmArgs& p);
// (two entry points instead of one more template parameter, so that the f32 kernels keep the symbols the profiles are keyed by)
template <int BM, int BN, int WM, int WN, int S, bool EPI>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void gemm_glds_kernel(const IgemmArgs p) { gemm_glds_body<BM, BN, WM, WN, S, EPI, 0>(p); }
template <int BM, int BN, int WM, int WN, int S, bool EPI>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void gemm_glds_x3_kernel(const IgemmArgs p) { gemm_glds_body<BM, BN, WM, WN, S, EPI, 3>(p); }
// BT: B stored transposed (w[z][n][k], row stride p.ldw) -- the forward Winograd filter bank U[ci][co] read as the B operand of the
// adjoint data gradient dV = dM U^T.  Its tile is then an image like A's ([BN rows][16 k], 64-byte rows, same XOR swizzle, same
// 16-byte fragment reads), so no transposed copy of the bank is ever made.
template <int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void gemm_glds_nt_kernel(const IgemmArgs p) { gemm_glds_body<BM, BN, WM, WN, S, false, 0, true>(p); }
template <int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void gemm_glds_nt_x3_kernel(const IgemmArgs p) { gemm_glds_body<BM, BN, WM, WN, S, false, 3, true>(p); }
template <int BM, int BN, int WM, int WN, int S, bool EPI>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void gemm_glds_x2_kernel(const IgemmArgs p) { gemm_glds_body<BM, BN, WM, WN, S, EPI, 2>(p); }
template <int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void gemm_glds_nt_x2_kernel(const IgemmArgs p) { gemm_glds_body<BM, BN, WM, WN, S, false, 2, true>(p); }
template <int BM, int BN, int WM, int WN, int S, bool EPI, int NSPLIT, bool BT>
static __device__ __forceinline__ void gemm_glds_body(const IgemmArgs& p)
{
    constexpr int BK = 16, CH = 4, RPI = 16;             // 16-byte chunks per A row; A rows per wave-instruction (1 KiB)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_PW = BM / RPI / 4, B_PW = BT ? BN / RPI / 4 : BK * BN / 256 / 4;   // LDS-DMA instructions per wave and K-tile
    static_assert(WM * WN == 4 && A_PW >= 1 && B_PW >= 1, "tile / wave split");
    constexpr int L = A_PW + B_PW;
    constexpr int STAGE = BM * BK + BK * BN;             // floats per stage
    __shared__ __attribute__((aligned(16))) float smem[S * STAGE];
    static_assert(S * STAGE >= 4 * 32 * 36, "epilogue patches");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // At most 32 rows in a 64-row tile (fc6 at batch 1: 32 Winograd tiles per position, a 1.6 GB filter bank to stream): the waves of the
    // lower half would multiply padding only.  They skip their MFMAs (they still load and synchronise), and which two waves -- i.e. which
    // two SIMDs -- do the work alternates between the blocks that share a CU (256 apart in dispatch order: 8 XCDs x 32 CUs round-robin).
    int cwave = wave;
    bool idle = false;
    if constexpr (BM == 64 && WM == 2 && WN == 2) {
        if (p.M <= 32) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      // dispatch order
            cwave = wave ^ (int)(((lin >> 8) & 1u) << 1); idle = cwave >= 2;
        }
    }
    const int wm = cwave / WN, wn = cwave % WN;
    const int pz = blockIdx.z;
    const float* __restrict__ Wp = p.w + (long long)pz * p.w_phase_stride;
    const float* __restrict__ X = p.batched ? p.x + (long long)pz * p.x_batch_stride : p.x;
    float* __restrict__ Y = p.batched ? p.y + (long long)pz * p.y_batch_stride : p.y;
    const unsigned ntn = (unsigned)(p.Cout / BN);
    const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned ntm = gridDim.x / ntn;
    const long long m0 = (long long)(p.m_fastest ? lid % ntm : lid / ntn) * BM;
    const int n0 = (int)(p.m_fastest ? lid / ntm : lid % ntn) * BN;

    unsigned a_voff[A_PW], b_voff[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int row = (wave * A_PW + i) * RPI + lane / CH, pc = lane % CH;
        const int c = pc ^ ((row >> 2) & 3);             // logical chunk held by physical chunk pc of this row
        long long m = m0 + row; if (m >= p.M) m = p.M - 1;
        a_voff[i] = (unsigned)((m - m0) * p.ldx + c * 4) * 4u;
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        if (BT) {
            const int row = (wave * B_PW + i) * RPI + lane / CH, pc = lane % CH;
            const int c = pc ^ ((row >> 2) & 3);
            b_voff[i] = (unsigned)(row * p.ldw + c * 4) * 4u;
        } else {
            const int f = (wave * B_PW + i) * 64 + lane;
            const int k = f / (BN / 4), j = (f % (BN / 4)) * 4;
            b_voff[i] = (unsigned)(k * p.Cout + j) * 4u;
        }
    }
    // split-K (gridDim.y > 1, linear epilogue only): this block reduces K-tiles [kt0, kt0 + nkt)
    const int nkt_all = p.Ktot / BK;
    const int kt0 = (int)((long long)nkt_all * blockIdx.y / gridDim.y);
    const int nkt = (int)((long long)nkt_all * (blockIdx.y + 1) / gridDim.y) - kt0;
    const float* a_base = X + m0 * p.ldx + (long long)kt0 * BK;
    const float* b_base = BT ? Wp + (long long)n0 * p.ldw + (long long)kt0 * BK : Wp + n0 + (long long)kt0 * BK * p.Cout;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    auto issue = [&](int kt, int stage) {
        const float* ga = a_base + (long long)kt * BK;
        const float* gb = BT ? b_base + (long long)kt * BK : b_base + (long long)kt * BK * p.Cout;
        const unsigned la = lds0 + (unsigned)(stage * STAGE + wave * A_PW * 256) * 4u;
        const unsigned lb = lds0 + (unsigned)(stage * STAGE + BM * BK + wave * B_PW * 256) * 4u;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) glds16(ga, a_voff[i], la + i * 1024);
#pragma unroll
        for (int i = 0; i < B_PW; ++i) glds16(gb, b_voff[i], lb + i * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[TM], a_sw[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = wm * TM * 32 + tm * 32 + (lane & 31);
        a_off[tm] = row * BK;
        a_sw[tm] = (row >> 2) & 3;
    }
    const int b_off = ((lane >> 5) * 4) * BN + wn * TN * 32 + (lane & 31);
    int bt_off[TN], bt_sw[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int row = wn * TN * 32 + tn * 32 + (lane & 31);
        bt_off[tn] = row * BK;
        bt_sw[tn] = (row >> 2) & 3;
    }
    auto compute = [&](int stage) {
        if (idle) return;
        const float* sa = smem + stage * STAGE;
        const float* sb = sa + BM * BK;
        if constexpr (NSPLIT >= 2) {                   // one K = 16 step of the bf16 MFMA per K-tile: lane half h holds k = 8h .. 8h + 7
            Bf16Pieces<NSPLIT> ap[TM], bp[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const float4 u = *reinterpret_cast<const float4*>(sa + a_off[tm] + (((2 * (lane >> 5)) ^ a_sw[tm]) * 4));
                const float4 v = *reinterpret_cast<const float4*>(sa + a_off[tm] + (((2 * (lane >> 5) + 1) ^ a_sw[tm]) * 4));
                const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
                split_bf16<NSPLIT>(x, ap[tm]);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float x[8];
                if (BT) {
                    const float4 u = *reinterpret_cast<const float4*>(sb + bt_off[tn] + (((2 * (lane >> 5)) ^ bt_sw[tn]) * 4));
                    const float4 v = *reinterpret_cast<const float4*>(sb + bt_off[tn] + (((2 * (lane >> 5) + 1) ^ bt_sw[tn]) * 4));
                    x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = sb[((lane >> 5) * 8 + j) * BN + wn * TN * 32 + tn * 32 + (lane & 31)];
                }
                split_bf16<NSPLIT>(x, bp[tn]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_split<NSPLIT>(ap[tm], bp[tn], acc[tm][tn]);
            return;
        }
#pragma unroll
        for (int kk2 = 0; kk2 < BK / 8; ++kk2) {       // (loading all fragments of the K-tile ahead of its MFMAs measured 7 % slower here)
            float4 af[TM];
            float bf[TN][4];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                af[tm] = *reinterpret_cast<const float4*>(sa + a_off[tm] + (((kk2 * 2 + (lane >> 5)) ^ a_sw[tm]) * 4));
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (BT) {
                    const float4 t = *reinterpret_cast<const float4*>(sb + bt_off[tn] + (((kk2 * 2 + (lane >> 5)) ^ bt_sw[tn]) * 4));
                    bf[tn][0] = t.x; bf[tn][1] = t.y; bf[tn][2] = t.z; bf[tn][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[tn][j] = sb[b_off + (kk2 * 8 + j) * BN + tn * 32];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const float av = j == 0 ? af[tm].x : j == 1 ? af[tm].y : j == 2 ? af[tm].z : af[tm].w;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[tn][j], acc[tm][tn], 0, 0, 0);
                }
        }
    };

    // Iteration kt: wait until this wave's pieces of tile kt have landed (the S - 2 newer tiles may stay in flight), barrier (every
    // wave's pieces have landed AND every wave is done reading the stage of tile kt - 1), refill that stage with tile kt + S - 1, compute.
#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + S - 2 < nkt) wait_vmcnt<(S - 2) * L>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        compute(stage);
        stage = stage + 1 == S ? 0 : stage + 1;
        pre = pre + 1 == S ? 0 : pre + 1;
    }

    if (gridDim.y > 1) {                         // split-K partials (Y zeroed by the launcher)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M) unsafeAtomicAdd(Y + m * p.ldy + n0 + wn * TN * 32 + tn * 32 + (lane & 31), acc[tm][tn][r]);
                }
        return;
    }
    if (!EPI) {
        // each wave transposes its 32x32 accumulator tiles through a private LDS patch: 16-byte global stores
        __builtin_amdgcn_s_barrier();            // every wave is done reading the last stage
        constexpr int LDT = 36;
        float* patch = smem + wave * 32 * LDT;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + (lane & 31)] = acc[tm][tn][r];
                __builtin_amdgcn_wave_barrier();
                const long long mrow = m0 + wm * TM * 32 + tm * 32;
                float* yb = Y + n0 + wn * TN * 32 + tn * 32 + (lane & 7) * 4;
                const float4 bv = p.bias ? ldg4(p.bias + n0 + wn * TN * 32 + tn * 32 + (lane & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = j * 8 + (lane >> 3);
                    float4 v = *reinterpret_cast<const float4*>(&patch[row * LDT + (lane & 7) * 4]);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;           // (the epilogue ops this path knows: a per-column bias, or a ReLU/dropout mask)
                    if (mrow + row < p.M) {
                        if (p.mask) {
                            const float4 mk = ldg4(p.mask + (mrow + row) * p.ldy + n0 + wn * TN * 32 + tn * 32 + (lane & 7) * 4);
                            v.x = mk.x > 0.f ? v.x * p.mask_scale : 0.f; v.y = mk.y > 0.f ? v.y * p.mask_scale : 0.f;
                            v.z = mk.z > 0.f ? v.z * p.mask_scale : 0.f; v.w = mk.w > 0.f ? v.w * p.mask_scale : 0.f;
                        }
                        *reinterpret_cast<float4*>(yb + (mrow + row) * p.ldy) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }
    // fused 1x1-conv epilogue (row m = output pixel m)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                const long long off = m * p.ldy;
                float v = acc[tm][tn][r] * p.alpha + bv;
                if (p.addend) v += p.addend[off + col];
                if (p.relu) v = v > 0.f ? v : 0.f;
                if (p.mask) v = p.mask[off + col] > 0.f ? v * p.mask_scale : 0.f;
                if (p.dropout) v = dropout_apply(v, (unsigned long long)(off + col), p.seed, p.stream_id, p.keep_prob);
                Y[off + col] = v;
            }
    }
}

// ===========================================================================
// conv1_1 forward (3 -> 64 channels on the 4-channel padded input): a write-bound layer (2.15 GB out, 0.13 GB in at 16 x 1024x512)
// ===========================================================================
// One 16-byte LDS-DMA chunk of the A image is exactly one tap of one pixel (b, g, r, 0), so the implicit-GEMM A tile
// [128 pixels][taps 4kt .. 4kt+3] is gathered straight from the image by per-lane addresses (taps outside the image read a zero
// pixel); K = 9 taps x 4 = 36, padded to 48 = three K-tiles, all in flight at once; bias + ReLU in the 16-byte-store epilogue.
// The generic predicated kernel needed 0.98 ms for this layer, this one is bound by the output stream.
static __device__ __forceinline__ void glds16v(const float* vaddr, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds_byte_off) : "memory", "m0");
}
struct Conv1Args { const float* x4; const float* w48; const float* bias; float* y; const float* zero16; int N, H, W; long long M;
                   unsigned short* yb16; long long yb16_ps; };        // (tile kernel) the consumer's padded bf16 copy [N][H + 2][W + 2][64], padded pixel 0; y may then be null

__global__ __launch_bounds__(256, 4) void conv1_glds_kernel(const Conv1Args p)
{
    constexpr int BM = 128, BN = 64, BK = 16, NKT = 3, WN = 2, TM = 2, TN = 1, CH = 4, RPI = 16;
    constexpr int A_PW = BM / RPI / 4;               // 2 LDS-DMA instructions per wave and K-tile for A, 1 for B
    constexpr int STAGE = BM * BK + BK * BN;
    __shared__ __attribute__((aligned(16))) float smem[NKT * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const long long m0 = (long long)blockIdx.x * BM;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int row = (wave * A_PW + i) * RPI + lane / CH, pc = lane % CH;
        const int c = pc ^ ((row >> 2) & 3);          // logical chunk (= tap within the K-tile) held by physical chunk pc
        long long m = m0 + row; if (m >= p.M) m = p.M - 1;
        const int x = (int)(m % p.W); const long long r = m / p.W; const int y = (int)(r % p.H);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int t = kt * 4 + c, dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = t < 9 && (unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W;
            const float* src = ok ? p.x4 + (m + (long long)dy * p.W + dx) * 4 : p.zero16;
            glds16v(src, lds0 + (unsigned)(kt * STAGE + (wave * A_PW + i) * 256) * 4u);
        }
    }
    {
        const int f = wave * 64 + lane;               // B: one instruction per wave and K-tile (16 x 64 floats = 4 KB)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            glds16v(p.w48 + (kt * BK + f / (BN / 4)) * BN + (f % (BN / 4)) * 4, lds0 + (unsigned)(kt * STAGE + BM * BK + wave * 256) * 4u);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int a_off[TM], a_sw[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = wm * TM * 32 + tm * 32 + (lane & 31);
        a_off[tm] = row * BK; a_sw[tm] = (row >> 2) & 3;
    }
    const int b_off = ((lane >> 5) * 4) * BN + wn * TN * 32 + (lane & 31);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const float* sa = smem + kt * STAGE;
        const float* sb = sa + BM * BK;
#pragma unroll
        for (int kk2 = 0; kk2 < BK / 8; ++kk2) {
            if (kt == NKT - 1 && kk2 == 1) continue;        // taps 10, 11: padding (all-zero rows of w48)
            float4 af[TM]; float bf[3];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[tm] = *reinterpret_cast<const float4*>(sa + a_off[tm] + (((kk2 * 2 + (lane >> 5)) ^ a_sw[tm]) * 4));
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = sb[b_off + (kk2 * 8 + j) * BN];
#pragma unroll
            for (int j = 0; j < 3; ++j)                     // (the fourth channel of every tap is the zero pad of the 4-channel image: 15 MFMA steps instead of 24)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const float av = j == 0 ? af[tm].x : j == 1 ? af[tm].y : af[tm].z;
                    acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[j], acc[tm][0], 0, 0, 0);
                }
        }
    }
    __builtin_amdgcn_s_barrier();
    constexpr int LDT = 36;
    float* patch = smem + wave * 32 * LDT;
    const float4 bv = ldg4(p.bias + wn * 32 + (lane & 7) * 4);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + (lane & 31)] = acc[tm][0][r];
        __builtin_amdgcn_wave_barrier();
        const long long mrow = m0 + wm * TM * 32 + tm * 32;
        float* yb = p.y + wn * 32 + (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 8 + (lane >> 3);
            float4 v = *reinterpret_cast<const float4*>(&patch[row * LDT + (lane & 7) * 4]);
            v.x = fmaxf(v.x + bv.x, 0.f); v.y = fmaxf(v.y + bv.y, 0.f); v.z = fmaxf(v.z + bv.z, 0.f); v.w = fmaxf(v.w + bv.w, 0.f);
            if (mrow + row < p.M) *reinterpret_cast<float4*>(yb + (mrow + row) * BN) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// Round 3: the same layer with the input staged as a SPATIAL tile.  The gather above fetches nine scattered 16-byte taps per pixel (1.2 GB of
// 16-byte requests for a 134 MB image: its time went into address processing, not into the 2.15 GB it writes).  Here a block owns 8 x 16
// pixels of one image, loads the 10 x 18 halo tile once (row-contiguous 16-byte loads, zero outside the image) and every MFMA A fragment is
// one ds_read_b128 straight from it: lane (pixel p, k-half h) of K-step (kt, kk2) needs the four channels of tap 4 kt + 2 kk2 + h at
// pixel p, i.e. halo[(py + dy + 1)][(px + dx + 1)] -- 16 consecutive pixels are 256 contiguous bytes, conflict-free.  Same products, same
// K order and the same epilogue as conv1_glds_kernel, so the results are bit-identical.
template <int TPB>        // tiles per block, consecutive along x: the kernel (12 KB) is staged once per block
__global__ __launch_bounds__(256, 4) void conv1_tile_kernel(const Conv1Args p)
{
    constexpr int TH = 8, TW = 16, BN = 64, BK = 16, NKT = 3, WN = 2, TM = 2, HW_ = TW + 2, HH_ = TH + 2;
    constexpr int LDT = 36;
    __shared__ __attribute__((aligned(16))) float sB[NKT * BK * BN];           // the 48 x 64 kernel (rows 36..47 zero)
    __shared__ __attribute__((aligned(16))) float4 halo[HH_ * HW_];
    __shared__ __attribute__((aligned(16))) float patches[4 * 32 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_x = p.W / TW / TPB, tiles_y = p.H / TH;
    const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y, n = blockIdx.x / (tiles_x * tiles_y);
    const int y0 = by * TH;
    // B: 48 x 64 floats = 768 float4, three per thread
#pragma unroll
    for (int i = 0; i < 3; ++i) reinterpret_cast<float4*>(sB)[tid + i * 256] = ldg4(p.w48 + (tid + i * 256) * 4);
    const float4 bv = ldg4(p.bias + wn * 32 + (lane & 7) * 4);
    float* patch = patches + wave * 32 * LDT;
#pragma unroll 1
  for (int it = 0; it < TPB; ++it) {
    const int x0 = (bx * TPB + it) * TW;
    if (it) __syncthreads();                               // every wave is done with the previous halo tile
    if (tid < HH_ * HW_) {
        const int hy = tid / HW_, hx = tid - hy * HW_, yy = y0 + hy - 1, xx = x0 + hx - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) v = ldg4(p.x4 + (((long long)n * p.H + yy) * p.W + xx) * 4);
        halo[tid] = v;
    }
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    // pixel of this lane in pixel group tm of the wave: group g = wm * TM + tm covers tile rows 2 g, 2 g + 1
    int hbase[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int g = wm * TM + tm, py = 2 * g + ((lane & 31) >> 4), px = lane & 15;
        hbase[tm] = (py + 1) * HW_ + (px + 1);
    }
    const int b_off = ((lane >> 5) * 4) * BN + wn * 32 + (lane & 31);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int kk2 = 0; kk2 < BK / 8; ++kk2) {
            if (kt == NKT - 1 && kk2 == 1) continue;        // taps 10, 11: padding (all-zero rows of w48)
            // this lane half's tap; tap 9 (kt 2, kk2 0, upper half) is padding too: it reads a valid pixel against zero kernel rows
            const int t = kt * 4 + kk2 * 2 + (lane >> 5), tt = t < 9 ? t : 4, dy = tt / 3 - 1, dx = tt % 3 - 1;
            float4 af[TM]; float bf[3];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[tm] = halo[hbase[tm] + dy * HW_ + dx];
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = sB[kt * BK * BN + b_off + (kk2 * 8 + j) * BN];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const float av = j == 0 ? af[tm].x : j == 1 ? af[tm].y : af[tm].z;
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[j], acc[tm], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + (lane & 31)] = acc[tm][r];
        __builtin_amdgcn_wave_barrier();
        const int g = wm * TM + tm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = j * 8 + (lane >> 3);              // pixel within the group: tile row 2 g + row / 16, column row % 16
            float4 v = *reinterpret_cast<const float4*>(&patch[row * LDT + (lane & 7) * 4]);
            v.x = fmaxf(v.x + bv.x, 0.f); v.y = fmaxf(v.y + bv.y, 0.f); v.z = fmaxf(v.z + bv.z, 0.f); v.w = fmaxf(v.w + bv.w, 0.f);
            const long long pix = ((long long)n * p.H + y0 + 2 * g + (row >> 4)) * p.W + x0 + (row & 15);
            if (p.y) *reinterpret_cast<float4*>(p.y + pix * BN + wn * 32 + (lane & 7) * 4) = v;
        }
        if (p.yb16) {
            // bf16_train: conv1_2 reads this layer as its padded bf16 input -- 8 channels (16 bytes) per lane, 16 pixels per pass
            const float4 b0 = ldg4(p.bias + wn * 32 + (lane & 3) * 8), b1 = ldg4(p.bias + wn * 32 + (lane & 3) * 8 + 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = j * 16 + (lane >> 2);
                const float4 u = *reinterpret_cast<const float4*>(&patch[row * LDT + (lane & 3) * 8]);
                const float4 w_ = *reinterpret_cast<const float4*>(&patch[row * LDT + (lane & 3) * 8 + 4]);
                __attribute__((ext_vector_type(8))) __bf16 o;
                o[0] = (__bf16)fmaxf(u.x + b0.x, 0.f); o[1] = (__bf16)fmaxf(u.y + b0.y, 0.f); o[2] = (__bf16)fmaxf(u.z + b0.z, 0.f); o[3] = (__bf16)fmaxf(u.w + b0.w, 0.f);
                o[4] = (__bf16)fmaxf(w_.x + b1.x, 0.f); o[5] = (__bf16)fmaxf(w_.y + b1.y, 0.f); o[6] = (__bf16)fmaxf(w_.z + b1.z, 0.f); o[7] = (__bf16)fmaxf(w_.w + b1.w, 0.f);
                const long long q = ((long long)n * (p.H + 2) + y0 + 2 * g + (row >> 4) + 1) * (p.W + 2) + x0 + (row & 15) + 1;
                // (channel-chunk planes: this wave's 32 channels are plane wn)
                *reinterpret_cast<__attribute__((ext_vector_type(8))) __bf16*>(p.yb16 + (p.yb16_ps ? (long long)wn * p.yb16_ps + q * 32 + (lane & 3) * 8 : q * BN + wn * 32 + (lane & 3) * 8)) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
  }
}

// x4: [N,H,W,4] (b, g, r, 0); w48: [48][64] = taps 0..8 x 4 channels, rows 36..47 zero; y = relu(conv + bias), [N,H,W,64]
// tiled: 1 = the spatial-tile kernel, 0 = the LDS-DMA gather kernel (model option "conv1_tiled"; bit-identical results)
// yb16 (tile kernel only): also -- or, with y == nullptr, only -- the padded bf16 copy of the output, [N][H + 2][W + 2][64] from padded pixel 0 (border kept zero by its owner)
bool launch_conv1_fwd(const float* x4, const float* w48, const float* bias, float* y, const float* zero16, int N, int H, int W, int Cout, int tiled, hipStream_t s, unsigned short* yb16, long long yb16_ps)
{
    if (Cout != 64 || !bias || !zero16) return false;
    if ((yb16 || !y) && !(tiled && H % 8 == 0 && W % 16 == 0)) return false;
    Conv1Args a{x4, w48, bias, y, zero16, N, H, W, (long long)N * H * W, yb16, yb16_ps};
    if (tiled && H % 8 == 0 && W % 16 == 0) {
        g_last_kernel = "conv1_tile_kernel";
        const long long tiles = (long long)N * (H / 8) * (W / 16);
        // (four tiles per block once there are enough of them: the 12 KB kernel is staged once per block -- 0.53 -> 0.46 ms; eight: the same)
        if (W % 64 == 0 && tiles >= 4 * 4096) hipLaunchKernelGGL(conv1_tile_kernel<4>, dim3((unsigned)(tiles / 4)), dim3(256), 0, s, a);
        else                                  hipLaunchKernelGGL(conv1_tile_kernel<1>, dim3((unsigned)tiles), dim3(256), 0, s, a);
        return true;
    }
    g_last_kernel = "conv1_glds_kernel";
    hipLaunchKernelGGL(conv1_glds_kernel, dim3((unsigned)((a.M + 127) / 128)), dim3(256), 0, s, a);
    return true;
}

template <int BM, int BN, int WM, int WN, int BKF = 16>
static void launch_igemm_cfg(const IgemmArgs& a0, int phases, hipStream_t s)
{
    IgemmArgs a = a0;
    a.m_fastest = (double)a.Ktot * a.Cout > (double)a.M * a.Cin;         // B (filters) larger than A (activations)
    dim3 grid((unsigned)(((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN)), 1, (unsigned)phases);
    const bool fast = (phases == 1 || a.batched) && a.Cin % BKF == 0 && a.Cout % BN == 0 && a.Ktot % a.Cin == 0;
    // filter bank small enough to live in L2 / MALL -> tap-inner K order (and at most 32 taps for the bit mask)
    const bool tap_inner = fast && (double)a.Ktot * a.Cout * 4.0 <= 64e6 && a.Ktot / a.Cin <= 32;
    const bool plain = fast && a.batched && a.Ktot == a.Cin && !a.bias && !a.addend && !a.mask && !a.relu && !a.dropout && a.alpha == 1.f &&
                       a.out_scale == 1 && a.ldy >= a.Cout && a.ldy % 4 == 0 && a.in_scale == 1 && a.tap_off == 0;
    const int mode = !fast ? 0 : (plain ? 3 : (tap_inner ? 2 : 1));
    static const std::string base = "igemm_fwd_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", ";
    static const std::string tags[4] = {base + "0, 16>", base + "1, " + std::to_string(BKF) + ">", base + "2, " + std::to_string(BKF) + ">",
                                        base + "3, " + std::to_string(BKF) + ">"};
    g_last_kernel = tags[mode].c_str();
    // few output tiles but a very long reduction (fc6 data gradient: 256 tiles, K = 200704): split K over
    // gridDim.y and combine with fp32 atomics -- only when the epilogue is linear
    // ... or few blocks altogether (batch-1 inference: fc6 has 32 tiles per Winograd position and must still stream a 1.6 GB
    // filter bank at HBM speed)
    const bool linear = (mode == 1 || mode == 3) && !a.relu && !a.mask && !a.dropout && !a.addend && a.out_scale == 1 && a.ldy == a.Cout;
    constexpr int splitk_min_kt = 64;        // fewest K-tiles a few-block launch must have before its reduction is split (32 measured slower at batch 1)
    const unsigned nblocks = grid.x * (unsigned)phases, nkt_all = (unsigned)(a.Ktot / BKF);
    // (deterministic mode: no split -- the partial sums would meet in atomics)
    if (linear && !t_deterministic && ((grid.x < 512 && nkt_all >= 512) || (nblocks < 2048 && nkt_all >= (unsigned)splitk_min_kt))) {
        unsigned ks = nkt_all >= 512 && grid.x < 512 ? 1024 / grid.x : 4096 / nblocks;
        if (ks > 8) ks = 8;
        if (ks > nkt_all / 16) ks = nkt_all / 16;
        if (ks >= 2) {
            grid.y = ks;
            const size_t nfl = a.batched ? (size_t)(phases - 1) * a.y_batch_stride + (size_t)a.M * a.Cout : (size_t)a.M * a.Cout;
            hipMemsetAsync(a.y, 0, nfl * sizeof(float), s);        // every slab of a batched launch
        }
    }
    // plain-GEMM rows with K % 16 == 0 and whole N tiles: the LDS-DMA main loop (gemm_glds_kernel)
    const bool rows1x1 = fast && a.Ktot == a.Cin && a.in_scale == 1 && a.tap_off == 0 && a.out_scale == 1 && a.out_offy == 0 && a.out_offx == 0 &&
                         (a.batched || (phases == 1 && a.Hi == a.Ma && a.Wi == a.Mb && a.Ho == a.Ma && a.Wo == a.Mb)) && a.ldy >= a.Cout;
    // nothing but a per-column bias (and no split-K): the glds kernel's 16-byte-store epilogue adds it
    const bool bias_only = fast && a.bias && !a.addend && !a.mask && !a.relu && !a.dropout && a.alpha == 1.f && a.ldy % 4 == 0 && a.Cout % 4 == 0 && grid.y == 1;
    if constexpr (BN >= 64 && BKF == 16) {
        if (rows1x1 && (mode == 3 || grid.y == 1) && (long long)(BM - 1) * a.ldx < (1LL << 29) && (long long)16 * a.Cout < (1LL << 29)) {
            // (the kernel symbols as rocprofv3 prints them, so that bench.py can look the PMC traffic of the dominant kernel up by name)
            static const std::string gbase = "gemm_glds_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3, ";
            static const std::string xbase = "gemm_glds_x3_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3, ";
            static const std::string gt[4] = {gbase + "false>", gbase + "true>", xbase + "false>", xbase + "true>"};
            // ... or nothing but a mask (the fc7 data gradient: ReLU + dropout of fc6)
            const bool mask_only = fast && a.mask && !a.bias && !a.addend && !a.relu && !a.dropout && a.alpha == 1.f && a.ldy % 4 == 0 && a.Cout % 4 == 0 && grid.y == 1;
            const bool epi = mode != 3 && !bias_only && !mask_only;
            if (a.bt) {                                         // transposed B: plain batched GEMMs only (the adjoint Winograd data gradient)
                static const std::string nb = "gemm_glds_nt_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3>";
                static const std::string nbx = "gemm_glds_nt_x3_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3>";
                if (mode != 3 || (long long)(BN - 1) * a.ldw >= (1LL << 29)) { defer_error(FCN8S_ERR_STATE, "transposed-B GEMM needs the plain batched form"); return; }
                static const std::string nbx2 = "gemm_glds_nt_x2_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3>";
                g_last_kernel = (a.split == 3 ? nbx : a.split == 2 ? nbx2 : nb).c_str();
                if (a.split == 3) hipLaunchKernelGGL((gemm_glds_nt_x3_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, s, a);
                else if (a.split == 2) hipLaunchKernelGGL((gemm_glds_nt_x2_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, s, a);
                else                   hipLaunchKernelGGL((gemm_glds_nt_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, s, a);
                return;
            }
            g_last_kernel = gt[(epi ? 1 : 0) + (a.split == 3 ? 2 : 0)].c_str();
            if (a.split == 3) {
                if (epi) hipLaunchKernelGGL((gemm_glds_x3_kernel<BM, BN, WM, WN, 3, true>), grid, dim3(256), 0, s, a);
                else     hipLaunchKernelGGL((gemm_glds_x3_kernel<BM, BN, WM, WN, 3, false>), grid, dim3(256), 0, s, a);
                return;
            }
            if (a.split == 2) {
                static const std::string x2base = "gemm_glds_x2_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3, ";
                static const std::string x2t[2] = {x2base + "false>", x2base + "true>"};
                g_last_kernel = x2t[epi ? 1 : 0].c_str();
                if (epi) hipLaunchKernelGGL((gemm_glds_x2_kernel<BM, BN, WM, WN, 3, true>), grid, dim3(256), 0, s, a);
                else     hipLaunchKernelGGL((gemm_glds_x2_kernel<BM, BN, WM, WN, 3, false>), grid, dim3(256), 0, s, a);
                return;
            }
            if (epi) hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, 3, true>), grid, dim3(256), 0, s, a);
            else     hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, 3, false>), grid, dim3(256), 0, s, a);
            return;
        }
    }
    if (a.bt) {      // igemm_fwd_kernel would read w[z][n][k] as if it were [k][n]: callers must only set bt for launches the branch above takes
        defer_error(FCN8S_ERR_STATE, "transposed-B GEMM (M %lld, K %d, N %d, tile %dx%d) does not fit the LDS-DMA kernel", a.M, a.Ktot, a.Cout, BM, BN);
        return;
    }
    if (mode == 3)      hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, 3, BKF>), grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, 2, BKF>), grid, dim3(256), 0, s, a);
    else if (mode == 1) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, 1, BKF>), grid, dim3(256), 0, s, a);
    else                hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, 0, 16>), grid, dim3(256), 0, s, a);
}

void launch_igemm(const IgemmArgs& a, int phases, hipStream_t s)
{
    // Tile choice.  With many blocks per CU the 128 x 128 tile wins (most MFMAs per LDS byte).  Small batches / deep layers have few row
    // tiles (1024x512 at batch 1: conv3_x 946 Winograd tiles per position, conv5_x 128), and then what matters is how the blocks fall onto the
    // 256 CUs: n = blocks per CU, of which `bpc` are resident at a time; a last, partial group of co-resident blocks runs below the CU's
    // rate (one 4-wave block alone reaches less than half of it).  cost = (full groups + penalised remainder) x tile area / tile efficiency,
    // fitted to batch-1 measurements (conv3_2: 0.094 / 0.078 / 0.076 ms for 128x128 / 64x128 / 64x64, conv4_2: 0.075 / 0.074 / 0.079,
    // conv5_x: - / 0.039 / 0.032); at training sizes (n >= 9) it keeps 128 x 128 everywhere.
    auto cost = [&](int bm, int bn, int bpc, double eff) {
        const long long blocks = ((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn) * phases;
        const long long n = (blocks + 255) / 256, rem = n % bpc;
        const double tail = rem == 0 ? 0.0 : (rem == 1 ? 2.0 : (rem == 2 ? 2.4 : 3.0));
        return ((double)(n / bpc) * bpc + tail) * bm * bn / eff;
    };
    if (a.Cout <= 32)      launch_igemm_cfg<128, 32, 4, 1>(a, phases, s);
    else if (a.Cout <= 64) {
        if (cost(128, 64, 4, 1.0) <= cost(64, 64, 4, 0.9)) launch_igemm_cfg<128, 64, 2, 2>(a, phases, s);
        else                                               launch_igemm_cfg<64, 64, 2, 2>(a, phases, s);
    } else {
        const double c128 = cost(128, 128, 3, 1.0), c64x128 = cost(64, 128, 4, 0.9), c64 = cost(64, 64, 4, 0.85);
        // (at most 32 rows -- fc6 at batch 1 -- the 64-row kernels idle the waves of the padded half, see gemm_glds_body)
        if (a.M > 64 && c128 <= c64x128 && c128 <= c64) launch_igemm_cfg<128, 128, 2, 2>(a, phases, s);   // (K-tile depth 32 and 256x64 tiles measured slower: fewer resident waves)
        else if (c64x128 <= c64)                        launch_igemm_cfg<64, 128, 2, 2>(a, phases, s);
        else                                            launch_igemm_cfg<64, 64, 2, 2>(a, phases, s);
    }
}

// ===========================================================================
// weight gradient
// ===========================================================================
// FAST = (Adim % BM == 0, Bdim % BN == 0, Pb >= 16): pixel coordinates of each thread's load slots
// are advanced incrementally (16 pixels per K-tile), loads are unconditional + select.
template <int BM, int BN, int WM, int WN, int WK, bool FAST, bool COLSUM>
__global__ __launch_bounds__(256, FAST ? 4 : 1) void wgrad_kernel(const WgradArgs p, const int chunk)
{
    constexpr int BK = 16, LDA = BM, LDB = BN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_F4 = BK * BM / 4, B_F4 = BK * BN / 4;
    constexpr int A_LD = (A_F4 + 255) / 256, B_LD = (B_F4 + 255) / 256;
    static_assert(WM * WN * WK == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (WM * WN), wmn = wave % (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;
    const int ntj = (p.Bdim + BN - 1) / BN;
    const int ti = blockIdx.x / ntj, tj = blockIdx.x - ti * ntj;
    const int i0 = ti * BM, j0 = tj * BN;
    const int tap = blockIdx.z;
    const int ty = tap / p.KW, tx = tap - ty * p.KW;
    // p.batched: gridDim.z enumerates independent GEMMs (Winograd positions) with their own A / B / C slabs, no tap shift
    const int dy = p.batched ? 0 : ty + p.tap_off, dx = p.batched ? 0 : tx + p.tap_off;
    const float* __restrict__ Ap = p.batched ? p.A + (long long)tap * p.a_batch_stride : p.A;
    const float* __restrict__ Bp = p.batched ? p.B + (long long)tap * p.b_batch_stride : p.B;
    const long long pbeg = (long long)blockIdx.y * chunk;
    const long long pend = (pbeg + chunk < p.P) ? pbeg + chunk : p.P;
    const int PaPb = p.Pa * p.Pb;
    const bool do_colsum = COLSUM && p.colsum != nullptr && tap == 0 && ti == 0 && wm == 0;

    float4 ra[A_LD], rb[B_LD];
    // ---- FAST path state (position of the next tile to load)
    int s_n[A_LD], s_a[A_LD], s_b[A_LD], s_left[A_LD];   // A slots: pixel coords + pixels left before pend
    const float* b_ptr[B_LD]; int b_left[B_LD];
    bool a_ldok[A_LD], b_ldok[B_LD];
    if (FAST) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int f = tid + i * 256;
            const int pix = f / (BM / 4);
            const long long pp = pbeg + pix;
            const long long q = pp < p.P ? pp : 0;
            s_n[i] = (int)(q / PaPb);
            const int r = (int)(q - (long long)s_n[i] * PaPb);
            s_a[i] = r / p.Pb; s_b[i] = r - s_a[i] * p.Pb;
            s_left[i] = (int)(pend - pp);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            const int pix = f / (BN / 4), c = (f - pix * (BN / 4)) * 4;
            b_ptr[i] = Bp + (pbeg + pix) * p.ldb + j0 + c;
            b_left[i] = (int)(pend - (pbeg + pix));
        }
    }
    auto gload_fast = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int f = tid + i * 256;
            const int c = (f % (BM / 4)) * 4;
            const int iy = s_a[i] * p.a_scale + dy, ix = s_b[i] * p.a_scale + dx;
            const bool ok = s_left[i] > 0 && (unsigned)iy < (unsigned)p.Ha && (unsigned)ix < (unsigned)p.Wa;
            const long long pixi = ok ? ((long long)s_n[i] * p.Ha + iy) * p.Wa + ix : 0;
            ra[i] = ldg4(Ap + pixi * p.lda + i0 + c);
            a_ldok[i] = ok;                        // select deferred to sstore
            s_left[i] -= BK;
            s_b[i] += BK;
            if (s_b[i] >= p.Pb) { s_b[i] -= p.Pb; if (++s_a[i] >= p.Pa) { s_a[i] = 0; ++s_n[i]; } }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const bool ok = b_left[i] > 0;
            rb[i] = ldg4(ok ? b_ptr[i] : Bp);
            b_ldok[i] = ok;
            b_ptr[i] += (long long)BK * p.ldb;
            b_left[i] -= BK;
        }
    };
    auto gload = [&](long long pk) {
        if (FAST) { gload_fast(); return; }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int f = tid + i * 256;
            const int pix = f / (BM / 4), c = (f - pix * (BM / 4)) * 4;
            const long long pp = pk + pix;
            bool ok = (A_F4 % 256 == 0 || f < A_F4) && pp < pend && (i0 + c) < p.Adim;
            long long off = 0;
            if (ok) {
                const int n = (int)(pp / PaPb);
                const int r = (int)(pp - (long long)n * PaPb);
                const int a = r / p.Pb, b = r - a * p.Pb;
                const int iy = a * p.a_scale + dy, ix = b * p.a_scale + dx;
                ok = (unsigned)iy < (unsigned)p.Ha && (unsigned)ix < (unsigned)p.Wa;
                off = (((long long)n * p.Ha + iy) * p.Wa + ix) * p.lda + i0 + c;
            }
            ra[i] = ok ? ldg4(Ap + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            const int pix = f / (BN / 4), c = (f - pix * (BN / 4)) * 4;
            const long long pp = pk + pix;
            const bool ok = (B_F4 % 256 == 0 || f < B_F4) && pp < pend && (j0 + c) < p.Bdim;
            rb[i] = ok ? ldg4(Bp + pp * p.ldb + j0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int f = tid + i * 256;
            if (A_F4 % 256 == 0 || f < A_F4) {
                const int pix = f / (BM / 4), c = (f - pix * (BM / 4)) * 4;
                if (FAST && !a_ldok[i]) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&As[buf * BK * LDA + pix * LDA + c]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int f = tid + i * 256;
            if (B_F4 % 256 == 0 || f < B_F4) {
                const int pix = f / (BN / 4), c = (f - pix * (BN / 4)) * 4;
                if (FAST && !b_ldok[i]) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Bs[buf * BK * LDB + pix * LDB + c]) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
    f32x16 accs[COLSUM ? TN : 1];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) if (COLSUM) accs[j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    auto compute = [&](int buf) {
        const float* A = As + buf * BK * LDA + (lane >> 5) * LDA + wm * TM * 32 + (lane & 31);
        const float* B = Bs + buf * BK * LDB + (lane >> 5) * LDB + wn * TN * 32 + (lane & 31);
        constexpr int KK = 8 / WK;
#pragma unroll
        for (int q = 0; q < KK; ++q) {
            const int kk = wk * KK + q;
            float af[TM], bf[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[tm] = A[kk * 2 * LDA + tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = B[kk * 2 * LDB + tn * 32];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
            if (COLSUM && do_colsum) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    accs[COLSUM ? tn : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, bf[tn], accs[COLSUM ? tn : 0], 0, 0, 0);
            }
        }
    };

    const int nkt = (int)((pend - pbeg + BK - 1) / BK);
    if (nkt <= 0) return;
    gload(pbeg);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(pbeg + (long long)(kt + 1) * BK);      // (issuing these mid-compute, as igemm_fwd_kernel does, measured 3 % slower here)
        compute(cur);
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    // (deterministic mode: reduction split (blockIdx.y, wk) owns the slab C + its index * split_stride and stores into it)
    float* Ct = p.C + (long long)tap * p.Areal * p.ldc + (long long)(blockIdx.y * WK + wk) * p.split_stride;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = j0 + wn * TN * 32 + tn * 32 + (lane & 31);
        if (col >= p.Bdim) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.Areal) {
                    if (p.plain_store) Ct[(long long)row * p.ldc + col] = acc[tm][tn][r] * p.alpha;     // sole writer of this tile
                    else unsafeAtomicAdd(Ct + (long long)row * p.ldc + col, acc[tm][tn][r] * p.alpha);
                }
            }
        if (COLSUM && do_colsum && lane < 32) unsafeAtomicAdd(p.colsum + col, accs[COLSUM ? tn : 0][0]);
    }
}


// ===========================================================================
// k x k weight gradient, several taps per block (3x3: all nine; 7x7: one filter row)
// ===========================================================================
// dW[ky][kx][ci][co] += sum_p X[p + (ky-1, kx-1)][ci] * dZ[p][co] for a 64(ci) x 64(co) tile and (3x3) ALL
// nine taps: the K-tile is a run of 16 consecutive pixels of one image row; its 3 x 18 input halo and
// the 16 dZ rows are staged in LDS once and feed 9 x 8 MFMA k-steps per wave (each wave: 32x32 tile,
// nine accumulators).  Compared with one tap per block this reads X and dZ ~4x less often
// (66 flop per staged byte instead of 16-32) and amortises the per-K-tile bookkeeping over 72 MFMAs.
// The bias gradient (column sums of dZ) is accumulated from the LDS copy by wave 0 of the ci-tile-0 blocks.
struct Wgrad9Args {
    const float* X; const float* dZ; float* dW; float* db;
    int N, H, W, Cin, Cout;
    long long nseg; int segs_per_block;
};

// KH x KW taps per block (3x3: all nine; 7x7: one filter row of seven, the row index comes from the grid).
template <int K, int KH, int KW>
__global__ __launch_bounds__(256, 2) void wgrad_taps_kernel(const Wgrad9Args p)
{
    constexpr int PAD = (K - 1) / 2, HCOLS = 16 + KW - 1, XROWS = KH * HCOLS, NT = KH * KW;
    constexpr int XF4 = XROWS * 16, X_LD = (XF4 + 255) / 256, BUF = XROWS * 64 + 16 * 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;
    const int nco = p.Cout / 64;
    const unsigned ntiles = (unsigned)(p.Cin / 64) * nco;
    constexpr unsigned NKY = K / KH;                               // filter-row groups handled by different blocks
    const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);       // 1-D grid: (pixel split, filter row, (ci, co) tile), tile fastest
    const unsigned tile = lid % ntiles, rest = lid / ntiles;
    const unsigned kyb = rest % NKY, split = rest / NKY;
    const int ky0 = (int)kyb * KH;
    const int ti = tile / nco, tj = tile - ti * nco;
    const int i0 = ti * 64, j0 = tj * 64;
    const int wsegs = p.W / 16;
    const long long s0 = (long long)split * p.segs_per_block;
    long long s1 = s0 + p.segs_per_block; if (s1 > p.nseg) s1 = p.nseg;
    const int nkt = (int)(s1 - s0);
    if (nkt <= 0) return;
    // position of the next segment to load (block-uniform)
    int n, h, ws;
    {
        const long long per_img = (long long)p.H * wsegs;
        n = (int)(s0 / per_img);
        const int r = (int)(s0 - (long long)n * per_img);
        h = r / wsegs; ws = r - h * wsegs;
    }
    int x_dy[X_LD], x_col[X_LD], x_c[X_LD];
#pragma unroll
    for (int i = 0; i < X_LD; ++i) {
        const int f = tid + i * 256;
        const int prow = f / 16;
        x_dy[i] = ky0 + prow / HCOLS - PAD; x_col[i] = prow % HCOLS - PAD; x_c[i] = (f % 16) * 4;
    }
    const int d_px = tid / 16, d_c = (tid % 16) * 4;

    float4 rx[X_LD], rd;
    bool okx[X_LD];
    auto gload = [&]() {
        const int w0 = ws * 16;
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            const int iy = h + x_dy[i], ix = w0 + x_col[i];
            const bool ok = (XF4 % 256 == 0 || tid + i * 256 < XF4) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const long long pix = ok ? ((long long)n * p.H + iy) * p.W + ix : 0;
            rx[i] = ldg4(p.X + pix * p.Cin + i0 + x_c[i]);
            okx[i] = ok;                           // zeroing deferred to sstore (no wait before the MFMAs)
        }
        rd = ldg4(p.dZ + (((long long)n * p.H + h) * p.W + w0 + d_px) * p.Cout + j0 + d_c);
        if (++ws == wsegs) { ws = 0; if (++h == p.H) { h = 0; ++n; } }
    };
    auto sstore = [&](int buf) {
        float* Xs = smem + buf * BUF;
        float* Ds = Xs + XROWS * 64;
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            const int f = tid + i * 256;
            if (XF4 % 256 == 0 || f < XF4)
                *reinterpret_cast<float4*>(&Xs[(f / 16) * 64 + x_c[i]]) = okx[i] ? rx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *reinterpret_cast<float4*>(&Ds[d_px * 64 + d_c]) = rd;
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = p.db != nullptr && ti == 0 && kyb == 0 && wave == 0;

    auto compute = [&](int buf) {
        const float* Xs = smem + buf * BUF;
        const float* Ds = Xs + XROWS * 64;
        const float* Xb = Xs + half * 64 + wm * 32 + (lane & 31);
        const float* Db = Ds + half * 64 + wn * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b = Db[kk * 2 * 64];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const float a = Xb[(ky * HCOLS + kk * 2 + kx) * 64];
                    acc[ky * KW + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ky * KW + kx], 0, 0, 0);
                }
        }
        if (do_bias) {
#pragma unroll
            for (int px = 0; px < 16; ++px) bsum += Ds[px * 64 + lane];
        }
    };

    gload();
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload();
        compute(cur);
        if (kt + 1 < nkt) sstore(cur ^ 1);
        __syncthreads();
    }

    const int col = j0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* Ct = p.dW + (long long)((ky0 + t / KW) * K + t % KW) * p.Cin * p.Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            unsafeAtomicAdd(Ct + (long long)row * p.Cout + col, acc[t][r]);
        }
    }
    if (do_bias) unsafeAtomicAdd(p.db + j0 + lane, bsum);
}

bool launch_wgrad_taps(const float* X, const float* dZ, float* dW, float* db, int N, int H, int W, int Cin, int Cout,
                       int K, hipStream_t s)
{
    if ((K != 3 && K != 7) || Cin % 64 || Cout % 64 || W % 16) return false;
    Wgrad9Args a{X, dZ, dW, db, N, H, W, Cin, Cout, (long long)N * H * (W / 16), 0};
    const long long work = (long long)(Cin / 64) * (Cout / 64) * (K == 3 ? 1 : 7);
    long long splits = 4096 / work;                    // ~16 blocks per CU in total
    if (t_deterministic) splits = 1;                   // (this kernel's splits meet in atomics; the default path does not use it)
    if (splits < 1) splits = 1;
    if (splits > (a.nseg + 7) / 8) splits = (a.nseg + 7) / 8;      // at least 8 K-tiles per block
    if (splits < 1) splits = 1;
    a.segs_per_block = (int)((a.nseg + splits - 1) / splits);
    splits = (a.nseg + a.segs_per_block - 1) / a.segs_per_block;
    if (K == 3) { g_last_kernel = "wgrad_taps_kernel<3, 3, 3>"; hipLaunchKernelGGL((wgrad_taps_kernel<3, 3, 3>), dim3((unsigned)(work * splits)), dim3(256), 0, s, a); }
    else        { g_last_kernel = "wgrad_taps_kernel<7, 1, 7>"; hipLaunchKernelGGL((wgrad_taps_kernel<7, 1, 7>), dim3((unsigned)(work * splits)), dim3(256), 0, s, a); }
    return true;
}

// ===========================================================================
// conv1_1 weight gradient (Cin = 3 padded to 4, Cout = 64): HBM-bound (reads the 134 MB/image dZ once),
// far too skinny for the matrix core (M = 3).  VALU kernel: a block walks 64-pixel row segments; thread
// (cq, pg) owns couts 4cq..4cq+3 and pixels {pg, pg+16, pg+32, pg+48} of the segment and keeps all
// 27 x 4 partial sums in registers; the 3 x 66 input halo sits in LDS as float4 (b, g, r, 0) and is
// read with broadcast ds_read_b128.  One cross-thread reduction + atomics per block at the very end.
// ===========================================================================
struct Conv1WgradArgs { const float4* X4; const float* dZ; float* dW; float* db; int N, H, W; long long nseg; int segs_per_block;
                        long long blk_stride; };      // != 0 (deterministic mode): block b adds into dW + b * blk_stride / db + b * blk_stride, summed in block order afterwards

__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const Conv1WgradArgs p)
{
    __shared__ __attribute__((aligned(16))) float4 xs[3 * 66];
    __shared__ float red[4 * 16 * 112];
    const int tid = threadIdx.x, cq = tid & 15, pg = tid >> 4;
    const int wsegs = p.W / 64;
    const long long s0 = (long long)blockIdx.x * p.segs_per_block;
    long long s1 = s0 + p.segs_per_block; if (s1 > p.nseg) s1 = p.nseg;
    float acc[27][4];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long s = s0; s < s1; ++s) {
        const long long per_img = (long long)p.H * wsegs;
        const int n = (int)(s / per_img);
        const int r = (int)(s - (long long)n * per_img);
        const int h = r / wsegs, w0 = (r - h * wsegs) * 64;
        // dZ for this thread's four pixels (issued first: longest latency)
        float4 dz[4];
        const float* dzrow = p.dZ + (((long long)n * p.H + h) * p.W + w0) * 64 + cq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) dz[j] = *reinterpret_cast<const float4*>(dzrow + (long long)(j * 16 + pg) * 64);
        __syncthreads();                      // previous segment's readers are done with xs
        if (tid < 198) {
            const int ky = tid / 66, col = tid - ky * 66;
            const int iy = h + ky - 1, ix = w0 + col - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = p.X4[((long long)n * p.H + iy) * p.W + ix];
            xs[tid] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int px = j * 16 + pg;
            const float4 d = dz[j];
            bs[0] += d.x; bs[1] += d.y; bs[2] += d.z; bs[3] += d.w;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 x = xs[ky * 66 + px + kx];
                    const int t = (ky * 3 + kx) * 3;
                    acc[t][0] += x.x * d.x; acc[t][1] += x.x * d.y; acc[t][2] += x.x * d.z; acc[t][3] += x.x * d.w;
                    acc[t + 1][0] += x.y * d.x; acc[t + 1][1] += x.y * d.y; acc[t + 1][2] += x.y * d.z; acc[t + 1][3] += x.y * d.w;
                    acc[t + 2][0] += x.z * d.x; acc[t + 2][1] += x.z * d.y; acc[t + 2][2] += x.z * d.z; acc[t + 2][3] += x.z * d.w;
                }
        }
    }
    // reduce over the 16 pixel groups: lanes with equal cq inside a wave (xor 16, 32), then across waves via LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[t][q];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][q] = v;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) { float v = bs[q]; v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); bs[q] = v; }
    if (lane < 16) {
        float* dst = red + (wave * 16 + lane) * 112;
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[t * 4 + q] = acc[t][q];
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[108 + q] = bs[q];
    }
    __syncthreads();
    // 16 cq x 112 values, summed over the 4 waves
    for (int i = tid; i < 16 * 112; i += 256) {
        const int c = i / 112, v = i - c * 112;
        const float sum = red[(0 * 16 + c) * 112 + v] + red[(1 * 16 + c) * 112 + v] + red[(2 * 16 + c) * 112 + v] + red[(3 * 16 + c) * 112 + v];
        if (v < 108) unsafeAtomicAdd(p.dW + blockIdx.x * p.blk_stride + (v >> 2) * 64 + c * 4 + (v & 3), sum);     // dW[tap*3+ci][co]
        else if (p.db) unsafeAtomicAdd(p.db + blockIdx.x * p.blk_stride + c * 4 + (v - 108), sum);
    }
}

// Round 3: the same gradient on the matrix core.  dW[(tap, ci)][co] = sum over pixels of x[pixel + tap][ci] * dZ[pixel][co] is a
// (27 -> 32) x 64 product with the pixels as the reduction: rows = (tap, ci), ROW 27 = all ones (its result is the bias gradient), K = two
// pixels per v_mfma_f32_32x32x2_f32, columns = two tiles of 32 couts.  A block walks 8 x 16 pixel tiles; the 10 x 18 input halo of a tile
// sits in LDS (as in conv1_tile_kernel) and a lane's A value is one ds_read_b32 of it; the B values come straight from dZ (lanes 0..31: 32
// consecutive couts of one pixel, lanes 32..63: of the pixel below) -- all 32 loads of a tile are in flight before the halo is staged.
// Each wave owns two rows of the tile and keeps its 2 x 16 accumulator registers over all tiles of the block; one LDS reduction over the
// four waves and one set of atomics per block at the end.  8.4 M MFMAs are 0.22 ms of matrix-pipe time: the kernel is bound by reading dZ.
__global__ __launch_bounds__(256, 4) void conv1_wgrad_mfma_kernel(const Conv1WgradArgs p, const int tiles_per_block, const long long ntiles)
{
    constexpr int TH = 8, TW = 16, HW_ = TW + 2, HH_ = TH + 2;
    __shared__ __attribute__((aligned(16))) float halo[HH_ * HW_ * 4];
    __shared__ float red[4 * 32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    const int i = lane & 31, h = lane >> 5;                      // MFMA row (tap, ci) of this lane's A values; which pixel of the K pair
    const int tap = i < 27 ? i / 3 : 4, ci = i < 27 ? i % 3 : 0, dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int a_off = ((2 * wave + h + dy + 1) * HW_ + (dx + 1)) * 4 + ci;       // + 4 kp: halo float of pixel (2 wave + h, kp) at this lane's tap
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long long t0 = (long long)blockIdx.x * tiles_per_block;
    long long t1 = t0 + tiles_per_block; if (t1 > ntiles) t1 = ntiles;
    // software pipeline: the dZ values and the halo pixel of tile t + 1 are loaded (into registers) before the MFMAs of tile t are issued
    float bn[16][2];
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](long long t) {
        const int bx = (int)(t % tiles_x), by = (int)((t / tiles_x) % tiles_y), n = (int)(t / ((long long)tiles_x * tiles_y));
        const int y0 = by * TH, x0 = bx * TW;
        // K pair kp = pixels (2 wave + h, kp); MFMA column i of N-tile t is cout 2 i + t: one 8-byte load per lane and pixel
        const float2* dzp = reinterpret_cast<const float2*>(p.dZ + (((long long)n * p.H + y0 + 2 * wave + h) * p.W + x0) * 64) + i;
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) { const float2 v = dzp[kp * 32]; bn[kp][0] = v.x; bn[kp][1] = v.y; }
        if (tid < HH_ * HW_) {
            const int hy = tid / HW_, hx = tid - hy * HW_, yy = y0 + hy - 1, xx = x0 + hx - 1;
            hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) hv = p.X4[((long long)n * p.H + yy) * p.W + xx];
        }
    };
    if (t0 < t1) fetch(t0);
    for (long long t = t0; t < t1; ++t) {
        float bz[16][2];
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) { bz[kp][0] = bn[kp][0]; bz[kp][1] = bn[kp][1]; }
        __syncthreads();                          // the previous tile's readers are done with the halo
        if (tid < HH_ * HW_) reinterpret_cast<float4*>(halo)[tid] = hv;
        __syncthreads();
        if (t + 1 < t1) fetch(t + 1);
#pragma unroll
        for (int kp = 0; kp < 16; ++kp) {
            float a = halo[a_off + 4 * kp];
            a = i < 27 ? a : (i == 27 ? 1.f : 0.f);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bz[kp][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bz[kp][1], acc[1], 0, 0, 0);
        }
    }
    // sum over the four waves, then one atomic per element and block
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 2 * i + t] = acc[t][r];
    __syncthreads();
    for (int e = tid; e < 28 * 64; e += 256) {
        const float sum = red[e] + red[32 * 64 + e] + red[2 * 32 * 64 + e] + red[3 * 32 * 64 + e];
        if (e < 27 * 64) unsafeAtomicAdd(p.dW + blockIdx.x * p.blk_stride + e, sum);                 // dW[tap * 3 + ci][co]
        else if (p.db) unsafeAtomicAdd(p.db + blockIdx.x * p.blk_stride + (e - 27 * 64), sum);
    }
}

// mfma: 1 = the matrix-core kernel, 0 = the VALU kernel above (model option "conv1_wgrad_mfma")
bool launch_conv1_wgrad(const float* X4, const float* dZ, float* dW, float* db, int N, int H, int W, int Cout, int mfma, hipStream_t s)
{
    // deterministic mode: every block adds into a zeroed slab of its own (27 x 64 weights + 64 biases), the slabs are summed in block order
    constexpr long long kSlab = 27 * 64 + 64;
    auto det_conv1 = [&](Conv1WgradArgs& a, long long blocks) -> float* {
        if (!t_deterministic || blocks <= 1) return nullptr;
        float* ws = det_scratch(s, (size_t)(blocks * kSlab));
        if (!ws) { defer_error(FCN8S_ERR_OOM, "deterministic mode: conv1_1's weight-gradient scratch (%lld floats) cannot be allocated", blocks * kSlab); return nullptr; }
        hipMemsetAsync(ws, 0, (size_t)(blocks * kSlab) * sizeof(float), s);
        a.dW = ws; a.db = db ? ws + 27 * 64 : nullptr; a.blk_stride = kSlab;
        return ws;
    };
    auto det_conv1_finish = [&](float* ws, long long blocks) {
        if (!ws) return;
        launch_det_reduce(dW, ws, 1, 27 * 64, 27 * 64, kSlab, (int)blocks, true, s);
        if (db) launch_det_reduce(db, ws + 27 * 64, 1, 64, 64, kSlab, (int)blocks, true, s);
    };
    if (Cout == 64 && mfma && H % 8 == 0 && W % 16 == 0) {
        Conv1WgradArgs a{(const float4*)X4, dZ, dW, db, N, H, W, 0, 0, 0};
        const long long ntiles = (long long)N * (H / 8) * (W / 16);
        long long blocks = ntiles < 1024 ? ntiles : 1024;               // 4 per CU, all resident; measured flat from 1024 to 2048, slower below (0.47 ms at 512)
        const int tpb = (int)((ntiles + blocks - 1) / blocks);
        blocks = (ntiles + tpb - 1) / tpb;
        g_last_kernel = "conv1_wgrad_mfma_kernel";
        float* ws = det_conv1(a, blocks);
        hipLaunchKernelGGL(conv1_wgrad_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, tpb, ntiles);
        det_conv1_finish(ws, blocks);
        return true;
    }
    if (Cout != 64 || W % 64) return false;
    Conv1WgradArgs a{(const float4*)X4, dZ, dW, db, N, H, W, (long long)N * H * (W / 64), 0, 0};
    // 512 blocks: every block ends in 1792 atomics on the same addresses, and those serialise (2048 blocks: 0.67 ms, 512: 0.60 ms;
    // a register-staged prefetch of the next segment needed 256 VGPRs and was slower)
    constexpr int maxb = 512;
    long long blocks = a.nseg < maxb ? a.nseg : maxb;
    a.segs_per_block = (int)((a.nseg + blocks - 1) / blocks);
    blocks = (a.nseg + a.segs_per_block - 1) / a.segs_per_block;
    g_last_kernel = "conv1_wgrad_kernel";
    float* ws = det_conv1(a, blocks);
    hipLaunchKernelGGL(conv1_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    det_conv1_finish(ws, blocks);
    return true;
}

// ===========================================================================
// weight gradient of the stride-S, K = 2S transposed conv (the 16x16 / stride-8 upsampler):
//   dW[ky][kx][co][ci] = sum_{n,i,j} dY[n, S*i + ky - S/2, S*j + kx - S/2, co] * X[n, i, j, ci]
// C = 20 channels: a 20 x 20 product per tap would waste 61 % of a 32 x 32 MFMA tile, and the generic kernel
// re-read dY once per tap (256 taps).  VALU instead: a block owns one filter row ky and a 64-pixel segment of input
// rows; the matching dY row segment (S*64 + S pixels x C) and the X segment sit in LDS, thread (kx mod 8, co)
// keeps the partial sums of two taps dW[ky][kx + 8q][co][:] in registers (packed fp32 FMAs; X via LDS broadcast reads).  dY is read twice in
// total (each output row serves two ky), atomics once per block at the end.
// ===========================================================================
struct TconvWgradArgs { const float* X; const float* dY; float* dW; int N, Hi, Wi, rows_per_block; };

template <int C, int K, int S>
__global__ __launch_bounds__(K / 2 * C) void tconv_wgrad_kernel(const TconvWgradArgs p)
{
    constexpr int JSEG = 64, PAD = (K - S) / 2, SEGPIX = JSEG * S + (K - S);      // dY pixels one segment touches
    constexpr int NT = K / 2 * C, TQ = 2, KQ = K / TQ, HALVES = NT / (KQ * C);     // thread = (kx mod KQ, co); owns the TQ taps kx + KQ*q
    // (TQ = 4 with the segment split between two thread halves was measured: 196 VGPRs, 2x slower)
    __shared__ __attribute__((aligned(16))) float dys[SEGPIX * C];
    __shared__ __attribute__((aligned(16))) float xs[JSEG * C];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, half = tid / (KQ * C), tt = tid - half * (KQ * C), kx = tt / C, co = tt - kx * C;
    const int ky = blockIdx.x, j0 = blockIdx.y * JSEG;
    const int Ho = p.Hi * S, Wo = p.Wi * S;
    const int ox0 = j0 * S - PAD;                                                 // first dY column of the segment (may be < 0)
    const int jn = p.Wi - j0 < JSEG ? p.Wi - j0 : JSEG;
    f32x2 acc[TQ][C / 2];                                                         // packed fp32 FMA: two ci per instruction
#pragma unroll
    for (int q = 0; q < TQ; ++q)
#pragma unroll
        for (int c = 0; c < C / 2; ++c) acc[q][c] = f32x2{0.f, 0.f};
    const long long row0 = (long long)blockIdx.z * p.rows_per_block, nrows = (long long)p.N * p.Hi;
    for (long long r = row0; r < row0 + p.rows_per_block && r < nrows; ++r) {
        const int n = (int)(r / p.Hi), i = (int)(r - (long long)n * p.Hi);
        const int oy = i * S + ky - PAD;
        if ((unsigned)oy >= (unsigned)Ho) continue;                               // block-uniform
        const float* dyrow = p.dY + ((long long)n * Ho + oy) * Wo * C;
        const float* xrow = p.X + (((long long)n * p.Hi + i) * p.Wi + j0) * C;
        constexpr int NF4 = SEGPIX * C / 4, FILL = (NF4 + NT - 1) / NT, XF4 = JSEG * C / 4, XFILL = (XF4 + NT - 1) / NT;
        float4 stage[FILL], xst[XFILL];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {                                       // all loads in flight before the first LDS store
            const int f = tid + it * NT;
            const int px = ox0 + (f * 4) / C;                                     // C % 4 == 0: a float4 never straddles two pixels
            stage[it] = (f < NF4 && (unsigned)px < (unsigned)Wo) ? *reinterpret_cast<const float4*>(dyrow + (long long)ox0 * C + f * 4)
                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < XFILL; ++it) {
            const int f = tid + it * NT;
            xst[it] = (f < jn * C / 4) ? *reinterpret_cast<const float4*>(xrow + f * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();                                                          // previous row's readers are done
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int f = tid + it * NT;
            if (f < NF4) *reinterpret_cast<float4*>(&dys[f * 4]) = stage[it];
        }
#pragma unroll
        for (int it = 0; it < XFILL; ++it) {
            const int f = tid + it * NT;
            if (f < XF4) *reinterpret_cast<float4*>(&xs[f * 4]) = xst[it];
        }
        __syncthreads();
#pragma unroll 2
        for (int jj = 0; jj < JSEG / HALVES; ++jj) {                              // columns >= jn hold zeros
            const int j = half * (JSEG / HALVES) + jj;
            f32x2 d[TQ];
#pragma unroll
            for (int q = 0; q < TQ; ++q) { const float v = dys[(j * S + kx + KQ * q) * C + co]; d[q] = f32x2{v, v}; }   // zero outside the image
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
                const float4 xv = *reinterpret_cast<const float4*>(&xs[j * C + c4 * 4]);             // one address per half-block: LDS broadcast
                const f32x2 xa = {xv.x, xv.y}, xb = {xv.z, xv.w};
#pragma unroll
                for (int q = 0; q < TQ; ++q) {
                    acc[q][2 * c4] = __builtin_elementwise_fma(d[q], xa, acc[q][2 * c4]);
                    acc[q][2 * c4 + 1] = __builtin_elementwise_fma(d[q], xb, acc[q][2 * c4 + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        float* out = p.dW + ((long long)(ky * K + kx + KQ * q) * C + co) * C;
#pragma unroll
        for (int c = 0; c < C / 2; ++c) { unsafeAtomicAdd(out + 2 * c, acc[q][c][0]); unsafeAtomicAdd(out + 2 * c + 1, acc[q][c][1]); }
    }
}

bool launch_tconv_wgrad(const float* X, const float* dY, float* dW, int N, int Hi, int Wi, int C, int K, int S, hipStream_t s)
{
    if (!(C == 20 && K == 16 && S == 8) || t_deterministic) return false;      // (its blocks meet in atomics: deterministic mode takes the generic weight gradient)
    TconvWgradArgs a{X, dY, dW, N, Hi, Wi, 0};
    const long long nrows = (long long)N * Hi;
    const int jsegs = (Wi + 63) / 64;
    long long zb = 2048 / (16 * jsegs); if (zb < 1) zb = 1; if (zb > nrows) zb = nrows;
    a.rows_per_block = (int)((nrows + zb - 1) / zb);
    zb = (nrows + a.rows_per_block - 1) / a.rows_per_block;
    g_last_kernel = "tconv_wgrad_kernel<20, 16, 8>";
    hipLaunchKernelGGL((tconv_wgrad_kernel<20, 16, 8>), dim3(16, (unsigned)jsegs, (unsigned)zb), dim3(160), 0, s, a);
    return true;
}

// ===========================================================================
// weight gradient over plain rows (Winograd positions, 1x1 convs): LDS-DMA main loop
// ===========================================================================
// C[z][i][j] (+)= alpha * sum_{rows of this block's chunk} A[z][row][i0 + i] * B[z][row][j0 + j].  Both operands are [row][channel],
// so both LDS images ([k][column]) are lane-linear as they come (no swizzle); same S = 3 stage pipeline as gemm_glds_kernel.  A
// chunk's last, partial K-tile (rows % 16, only when the row count is not a multiple of 16) goes through registers with zero fill.
// 1-D grid, XCD-aware: the (ci, co) tiles of one (position, row chunk) share their A / B row panels and are neighbours in the id
// order, i.e. run behind the same L2 (the 3-D grid of wgrad_kernel deals them round-robin over all eight).
template <int BM, int BN, int WM, int WN, int S, bool PRELOAD, int NSPLIT>
static __device__ __forceinline__ void wgrad_glds_body(const WgradArgs& p, const int chunk, const int nsplit);
template <int BM, int BN, int WM, int WN, int S, bool PRELOAD>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void wgrad_glds_kernel(const WgradArgs p, const int chunk, const int nsplit) { wgrad_glds_body<BM, BN, WM, WN, S, PRELOAD, 0>(p, chunk, nsplit); }
template <int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void wgrad_glds_x3_kernel(const WgradArgs p, const int chunk, const int nsplit) { wgrad_glds_body<BM, BN, WM, WN, S, true, 3>(p, chunk, nsplit); }
template <int BM, int BN, int WM, int WN, int S>
__global__ __launch_bounds__(256, (S * (BM + BN) * 64 <= 40960) ? 4 : 3) void wgrad_glds_x2_kernel(const WgradArgs p, const int chunk, const int nsplit) { wgrad_glds_body<BM, BN, WM, WN, S, true, 2>(p, chunk, nsplit); }
template <int BM, int BN, int WM, int WN, int S, bool PRELOAD, int NSPLIT>
static __device__ __forceinline__ void wgrad_glds_body(const WgradArgs& p, const int chunk, const int nsplit)
{
    constexpr int BK = 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_PW = BK * BM / 256 / 4, B_PW = BK * BN / 256 / 4;
    static_assert(WM * WN == 4 && A_PW >= 1 && B_PW >= 1, "tile / wave split");
    constexpr int L = A_PW + B_PW;
    constexpr int STAGE = BK * (BM + BN);
    __shared__ __attribute__((aligned(16))) float smem[S * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const unsigned ntj = (unsigned)(p.Bdim / BN), ntiles = (unsigned)(p.Adim / BM) * ntj;
    const unsigned lid = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned tile = lid % ntiles, rest = lid / ntiles;
    const unsigned ys = rest % (unsigned)nsplit, z = rest / (unsigned)nsplit;
    const int i0 = (int)(tile / ntj) * BM, j0 = (int)(tile % ntj) * BN;
    const float* __restrict__ Ap = p.batched ? p.A + (long long)z * p.a_batch_stride : p.A;
    const float* __restrict__ Bp = p.batched ? p.B + (long long)z * p.b_batch_stride : p.B;
    const long long pbeg = (long long)ys * chunk;
    const long long pend = (pbeg + chunk < p.P) ? pbeg + chunk : p.P;
    const int nkt = (int)((pend - pbeg) / BK), rem = (int)((pend - pbeg) % BK);

    unsigned a_voff[A_PW], b_voff[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        const int f = (wave * A_PW + i) * 64 + lane;
        a_voff[i] = (unsigned)((f / (BM / 4)) * p.lda + (f % (BM / 4)) * 4) * 4u;
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        const int f = (wave * B_PW + i) * 64 + lane;
        b_voff[i] = (unsigned)((f / (BN / 4)) * p.ldb + (f % (BN / 4)) * 4) * 4u;
    }
    const float* a_base = Ap + pbeg * p.lda + i0;
    const float* b_base = Bp + pbeg * p.ldb + j0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    auto issue = [&](int kt, int stage) {
        const float* ga = a_base + (long long)kt * BK * p.lda;
        const float* gb = b_base + (long long)kt * BK * p.ldb;
        const unsigned la = lds0 + (unsigned)(stage * STAGE + wave * A_PW * 256) * 4u;
        const unsigned lb = lds0 + (unsigned)(stage * STAGE + BK * BM + wave * B_PW * 256) * 4u;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) glds16(ga, a_voff[i], la + i * 1024);
#pragma unroll
        for (int i = 0; i < B_PW; ++i) glds16(gb, b_voff[i], lb + i * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = (lane >> 5) * BM + wm * TM * 32 + (lane & 31);
    const int b_off = (lane >> 5) * BN + wn * TN * 32 + (lane & 31);
    auto compute = [&](int stage) {
        if constexpr (NSPLIT >= 2) {                   // f32x3 / f32x2 (see split_bf16): lane half h holds rows k = 8h .. 8h + 7 of the K-tile
            const float* sa3 = smem + stage * STAGE + (lane >> 5) * 8 * BM + wm * TM * 32 + (lane & 31);
            const float* sb3 = smem + stage * STAGE + BK * BM + (lane >> 5) * 8 * BN + wn * TN * 32 + (lane & 31);
            Bf16Pieces<NSPLIT> ap[TM], bp[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sa3[j * BM + tm * 32];
                split_bf16<NSPLIT>(x, ap[tm]);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = sb3[j * BN + tn * 32];
                split_bf16<NSPLIT>(x, bp[tn]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_split<NSPLIT>(ap[tm], bp[tn], acc[tm][tn]);
            return;
        }
        const float* sa = smem + stage * STAGE + a_off;
        const float* sb = smem + stage * STAGE + BK * BM + b_off;
        // all fragments of the K-tile first (8 x (TM + TN) VGPRs), then the MFMAs back to back: with the reads interleaved one
        // k-step at a time the wave sat at lgkmcnt(0) in front of every group of TM * TN MFMAs
        float af[BK / 2][TM], bf[BK / 2][TN];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[kk][tm] = sa[kk * 2 * BM + tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[kk][tn] = sb[kk * 2 * BN + tn * 32];
        }
        if (PRELOAD) __builtin_amdgcn_sched_barrier(0);         // (the machine scheduler otherwise sinks the reads back between the MFMAs)
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][tm], bf[kk][tn], acc[tm][tn], 0, 0, 0);
    };

#pragma unroll
    for (int t = 0; t < S - 1; ++t)
        if (t < nkt) issue(t, t);
    int stage = 0, pre = S - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + S - 2 < nkt) wait_vmcnt<(S - 2) * L>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, pre);
        compute(stage);
        stage = stage + 1 == S ? 0 : stage + 1;
        pre = pre + 1 == S ? 0 : pre + 1;
    }
    if (rem > 0) {                               // partial last K-tile: through registers, rows >= rem are zeros
        __syncthreads();
        const float* ga = a_base + (long long)nkt * BK * p.lda;
        const float* gb = b_base + (long long)nkt * BK * p.ldb;
#pragma unroll
        for (int i = 0; i < BK * BM / 4 / 256; ++i) {
            const int f = tid + i * 256, k = f / (BM / 4), c = (f % (BM / 4)) * 4;
            *reinterpret_cast<float4*>(&smem[k * BM + c]) = k < rem ? ldg4(ga + (long long)k * p.lda + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < BK * BN / 4 / 256; ++i) {
            const int f = tid + i * 256, k = f / (BN / 4), c = (f % (BN / 4)) * 4;
            *reinterpret_cast<float4*>(&smem[BK * BM + k * BN + c]) = k < rem ? ldg4(gb + (long long)k * p.ldb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        compute(0);
    }

    float* Ct = p.C + (long long)z * p.Areal * p.ldc + (long long)ys * p.split_stride;      // (split_stride != 0: deterministic mode, one slab per row chunk)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = j0 + wn * TN * 32 + tn * 32 + (lane & 31);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.Areal) {
                    if (p.plain_store) Ct[(long long)row * p.ldc + col] = acc[tm][tn][r] * p.alpha;     // sole writer of this tile
                    else unsafeAtomicAdd(Ct + (long long)row * p.ldc + col, acc[tm][tn][r] * p.alpha);
                }
            }
    }
}

template <int BM, int BN, int WM, int WN, int WK>
static void launch_wgrad_cfg(const WgradArgs& a, hipStream_t s)
{
    const int nti = (a.Adim + BM - 1) / BM, ntj = (a.Bdim + BN - 1) / BN;
    const long long tiles = (long long)nti * ntj * a.ntaps;
    // split the pixel reduction so that ~8 blocks per CU are in flight
    long long want = 2048 / tiles;                     // floor: stay just under a whole number of resident rounds
    long long maxsplit = (a.P + 255) / 256;            // at least 256 pixels per block
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    long long chunk = (a.P + want - 1) / want;
    chunk = (chunk + 15) / 16 * 16;
    const int splits = (int)((a.P + chunk - 1) / chunk);
    dim3 grid((unsigned)(nti * ntj), (unsigned)splits, (unsigned)a.ntaps);
    constexpr bool full_tiles = (BK_ * BM / 4) % 256 == 0 && (BK_ * BN / 4) % 256 == 0;
    const bool fast = full_tiles && a.Adim % BM == 0 && a.Bdim % BN == 0 && a.Pb >= 16;
    // big tiles: keep the bias-gradient accumulators out of the register budget (separate column-sum pass)
    constexpr bool fused_colsum = BM * BN < 128 * 128;
    WgradArgs b = a;
    const bool det = t_deterministic != 0;
    if ((!fused_colsum || det) && a.colsum) { launch_colsum(a.B, a.colsum, a.P, a.Bdim, s); b.colsum = nullptr; }     // (deterministic: never the fused atomics)
    b.plain_store = 0; b.split_stride = 0;
    // Deterministic mode: `nslabs` partial results -> scratch slabs (plain stores, every element of a slab has one writer), then added in slab
    // order into C by launch_det_reduce.  Returns false (and leaves b alone) when one slab is all there is or the scratch cannot be had.
    const long long slab = (long long)a.ntaps * a.Areal * a.ldc;
    auto det_slabs = [&](long long nslabs) -> bool {
        if (nslabs <= 1) { b.plain_store = a.c_uninitialized ? 1 : 0; return false; }      // a single writer per element: an atomic add onto C is reproducible
        float* ws = det_scratch(s, (size_t)(nslabs * slab));
        if (!ws) { defer_error(FCN8S_ERR_OOM, "deterministic mode: a weight gradient's slab scratch (%lld floats) cannot be allocated", nslabs * slab); b.plain_store = a.c_uninitialized ? 1 : 0; return false; }
        b.C = ws; b.split_stride = slab; b.plain_store = 1;
        return true;
    };
    auto det_finish = [&](long long nslabs) { launch_det_reduce(a.C, b.C, (long long)a.ntaps * a.Areal, a.Bdim, a.ldc, slab, (int)nslabs, !a.c_uninitialized, s); };
    if (a.c_uninitialized && !det) {            // the caller did not zero C: store directly when every tile has a single writer, else zero it here
        if (splits == 1 && WK == 1) b.plain_store = 1;
        else hipMemsetAsync(a.C, 0, (size_t)a.ntaps * a.Areal * a.ldc * sizeof(float), s);
    }
    static const std::string tag = "wgrad_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ">";
    g_last_kernel = tag.c_str();
    if constexpr (WK == 1 && BM >= 64 && BN >= 64) {
        const bool rows = a.batched || (a.ntaps == 1 && a.KW == 1 && a.a_scale == 1 && a.tap_off == 0 && a.Ha == a.Pa && a.Wa == a.Pb);
        if (full_tiles && a.Adim % BM == 0 && a.Bdim % BN == 0 && rows && !b.colsum && a.lda <= (1 << 20) && a.ldb <= (1 << 20)) {
            // All blocks of a launch do the same work, so the launch runs in whole rounds of (256 CUs x resident blocks): pick the row
            // split that wastes least of the last round (2048 blocks on 768 slots = 2.67 rounds idle a ninth of the chip), then the fewest splits
            // (every split adds one pass of atomics over C).
            constexpr int slots = 256 * ((3 * (BM + BN) * 64 <= 40960) ? 4 : 3);
            long long best = 1; double best_score = -1e9;
            for (long long w = 1; w <= maxsplit && (w == 1 || tiles * w <= 8LL * slots); ++w) {
                const double r = (double)tiles * w / slots;
                double score = r / std::ceil(r) - 0.004 * (double)w;
                if (r < 1.0) score -= 1.0;
                if (score > best_score) { best_score = score; best = w; }
            }
            long long gchunk = ((a.P + best - 1) / best + 15) / 16 * 16;
            const int gsplits = (int)((a.P + gchunk - 1) / gchunk);
            bool slabs = false;
            if (det) slabs = det_slabs(gsplits);
            else {
                b.plain_store = 0;
                if (a.c_uninitialized && gsplits == 1) b.plain_store = 1;
                else if (a.c_uninitialized && splits == 1) hipMemsetAsync(a.C, 0, (size_t)a.ntaps * a.Areal * a.ldc * sizeof(float), s);   // (zeroed above otherwise)
            }
            static const std::string gbase = "wgrad_glds_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3, ";
            static const std::string gtag = gbase + "true>";     // (fragments of a whole K-tile preloaded before its MFMAs: +3 % over the interleaved order)
            g_last_kernel = gtag.c_str();
            const dim3 ggrid((unsigned)(nti * ntj) * (unsigned)gsplits * (unsigned)a.ntaps);
            if (a.split == 3) { static const std::string xtag = "wgrad_glds_x3_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3>"; g_last_kernel = xtag.c_str();
                                     hipLaunchKernelGGL((wgrad_glds_x3_kernel<BM, BN, WM, WN, 3>), ggrid, dim3(256), 0, s, b, (int)gchunk, gsplits); }
            else if (a.split == 2) { static const std::string x2tag = "wgrad_glds_x2_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", 3>"; g_last_kernel = x2tag.c_str();
                                     hipLaunchKernelGGL((wgrad_glds_x2_kernel<BM, BN, WM, WN, 3>), ggrid, dim3(256), 0, s, b, (int)gchunk, gsplits); }
            else hipLaunchKernelGGL((wgrad_glds_kernel<BM, BN, WM, WN, 3, true>), ggrid, dim3(256), 0, s, b, (int)gchunk, gsplits);
            if (slabs) det_finish(gsplits);
            return;
        }
    }
    if (det && det_slabs((long long)splits * WK)) {
        if (fast) hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, WK, true, false>), grid, dim3(256), 0, s, b, (int)chunk);
        else      hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, WK, false, false>), grid, dim3(256), 0, s, b, (int)chunk);
        det_finish((long long)splits * WK);
        return;
    }
    if (fast) hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, WK, true, fused_colsum>), grid, dim3(256), 0, s, b, (int)chunk);
    else      hipLaunchKernelGGL((wgrad_kernel<BM, BN, WM, WN, WK, false, fused_colsum>), grid, dim3(256), 0, s, b, (int)chunk);
}

void launch_wgrad(const WgradArgs& a, hipStream_t s)
{
    const bool small_i = a.Adim <= 32, small_j = a.Bdim <= 32;
    if (small_i && small_j)      launch_wgrad_cfg<32, 32, 1, 1, 4>(a, s);
    else if (small_i)            launch_wgrad_cfg<32, 64, 1, 2, 2>(a, s);
    else if (small_j)            launch_wgrad_cfg<64, 32, 2, 1, 2>(a, s);
    else if (a.Adim <= 64 || a.Bdim <= 64) launch_wgrad_cfg<64, 64, 2, 2, 1>(a, s);
    else                         launch_wgrad_cfg<128, 128, 2, 2, 1>(a, s);
}

}  // namespace fcn8s
