// Host side of libfcn8s_hip.so: the model handle (variables, HBM arena,
// optimizer slots, metric accumulators) and the launch sequences that replace
// the reference's three hot sess.run sites (fcn8s_tensorflow.py:554-572,
// 685-689, 764-770).  Everything is enqueued on one HIP stream; the only host
// synchronisations are the ones the reference's fetches imply (loss, metrics,
// predictions).
#include "../../include/fcn8s_hip.h"
#include "fcn8s_internal.h"
#include <cstdarg>

#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <tuple>
#include <mutex>
#include <map>
#include <set>
#include <string>
#include <vector>

using namespace fcn8s;

// The few RCCL (= NCCL API) types behind the entry points this file resolves with dlsym: declared here, with the values of rccl.h (RCCL 2.x; the
// NCCL API keeps them stable), so that building the library needs no RCCL development headers -- a single-GPU user never touches RCCL at all.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat = 7, ncclDouble = 8 } ncclDataType_t;
}
static_assert(sizeof(ncclUniqueId) == FCN8S_COMM_ID_BYTES, "fcn8s_comm_unique_id hands out ncclUniqueId bytes");

namespace {

thread_local std::string g_last_error;
thread_local int t_deferred_code = 0;
thread_local std::string t_deferred_text;

struct ParamInfo {
    std::string name;
    int ndim;
    int64_t shape[4];
    size_t offset, numel;
};

struct ProfGroup {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;          // owned
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_shared;   // borrowed from another group (per-kernel-symbol view)
    double flops = 0, bytes = 0;
    int64_t launches = 0;
};

struct Act { float* p = nullptr; size_t n = 0; int H = 0, W = 0, C = 0; };

const int kConvsPerBlock[5] = {2, 2, 3, 3, 3};

}  // namespace

namespace fcn8s {
void defer_error(int code, const char* fmt, ...)
{
    if (t_deferred_code) return;
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    t_deferred_code = code; t_deferred_text = buf;
}
int take_deferred_error(std::string* text)
{
    const int c = t_deferred_code;
    if (c && text) *text = t_deferred_text;
    t_deferred_code = 0; t_deferred_text.clear();
    return c;
}
}  // namespace fcn8s

struct fcn8s_model {
    int C = 20, fc6k = 7, device = 0;
    int widths[7] = {64, 128, 256, 512, 512, 4096, 4096};
    uint64_t seed = 0;
    std::vector<ParamInfo> params;
    std::map<std::string, int> index;
    size_t total = 0;
    size_t bucket_off[FCN8S_MAX_BUCKETS] = {0}, bucket_n[FCN8S_MAX_BUCKETS] = {0};
    hipEvent_t bucket_ev[FCN8S_MAX_BUCKETS] = {nullptr};                 // recorded right behind the last kernel that writes into the bucket
    bool bucket_final[FCN8S_MAX_BUCKETS] = {false};                      // ... this backward pass (fcn8s_bucket_wait)
    float* dz7_cur = nullptr; bool defer_fc_cur = false;                 // handed from backward phase 0 (decoder, fc7 weights) to phase 1 (fc6)
    // the library's own RCCL communicator (fcn8s_comm_*): one rank per model, collectives on a stream of its own behind the bucket events
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_world = 1;
    hipStream_t comm_stream = nullptr;
    // slot b < FCN8S_MAX_BUCKETS: the all-reduce of gradient bucket b; slot FCN8S_MAX_BUCKETS: the parameter broadcast / the metrics all-reduce
    static constexpr int kCommSlots = FCN8S_MAX_BUCKETS + 1, kCommMisc = FCN8S_MAX_BUCKETS;
    hipEvent_t comm_done[kCommSlots] = {nullptr}; bool comm_pending[FCN8S_MAX_BUCKETS] = {false};
    // ... and its watchdog (world > 1): a thread that polls ncclCommGetAsyncError and the age of every all-reduce still in flight; on an
    // asynchronous error or after comm_timeout_ms it calls ncclCommAbort (RCCL's kernels then leave the streams they block) and records
    // why -- every later fcn8s_comm_* / fcn8s_apply_update call returns FCN8S_ERR_RCCL with that text instead of waiting for a dead peer
    std::mutex comm_mu;                                                   // guards comm (enqueue vs. abort), comm_enq_ns, comm_error
    std::thread comm_watch; std::atomic<bool> comm_watch_stop{false}, comm_failed{false};
    std::atomic<bool> comm_inflight[kCommSlots] = {}; int64_t comm_enq_ns[kCommSlots] = {0};
    std::string comm_error; int64_t comm_timeout_ms = 600000;
    float *d_params = nullptr, *d_grads = nullptr, *d_m = nullptr, *d_v = nullptr, *d_wt = nullptr;
    bool own_params = false, own_grads = false;
    float *d_w1pad = nullptr, *d_tph[3] = {nullptr, nullptr, nullptr};
    float *d_wino_u = nullptr, *d_wino_v = nullptr, *d_wino_m = nullptr;   // Winograd scratch: filters, transformed input / output
    int wino_min_cin = 64;                                              // 3x3 layers with Cin >= this use Winograd; 0 = never
    int wino_tile = 6;                                                    // largest 3x3 output tile: F(6x6,3x3) / F(4x4,3x3) per layer by cost, F(2x2,3x3) fallback
    int wino_fc6 = 1;                                                     // fc6 7x7 as a 2x2 grid of 4x4 sub-filters in the Winograd domain
    int wino_tile_hires = 0, wino_hires_pixels = 0;                       // != 0: 3x3 layers on maps of at least wino_hires_pixels pixels (per image) use at most this tile
    int wino_force_tile = 0;                                              // != 0: every eligible 3x3 layer uses exactly this tile (op-level parity entry point)
    int precision = FCN8S_PREC_F32;                                       // FCN8S_PREC_BF16_FC: forward fc6 / fc7 on the bf16 MFMA
    bool pool_fused[5] = {false, false, false, false, false};            // forward wrote pool_b + argmax bytes from conv_b_last's output transform
    // fcn8s_freeze_params: the caller promises constant parameters; Winograd-transformed filters are then kept per layer
    bool frozen = false;
    unsigned long long frozen_fp = 0; unsigned long long* d_fp = nullptr;   // fingerprint of the parameter buffer the cached banks were built from
    unsigned long long* h_fp = nullptr; hipEvent_t fp_event = nullptr;      // pinned landing place of the guard's fingerprint, and the event behind its copy
    std::map<std::string, float*> u_cache;                               // layer -> transformed filter bank (hipMalloc'ed), valid while frozen
    std::map<std::string, float*> u_train;                               // layer -> forward filter bank of the current training step: the adjoint data
                                                                         // gradient reads it as a transposed B operand (no second, transposed bank)
    bool fwd_train = false;                                              // forward() is running a training pass
    std::set<std::string> rbits_ok;                                       // layers whose forward pass wrote a ReLU bit mask ("rb:<layer>") this step
    std::string fused_v_layer;                                            // layer whose data-gradient input transform already sits in d_wino_v
    std::string dm_layer;                                                 // layer whose dM = A dY A^T sits in d_wino_m, ready for the adjoint data gradient
    std::string dm_prefilled;                                             // layer whose dM the data gradient of the layer after it has already written into d_wino_m (fused transform)
    int deterministic = 0;                                                // option: reductions split over blocks are joined in a fixed order (slabs + ordered sum) instead of atomics
    int fuse_dgrad_dout = 1;                                              // option: allow that fusion
    int fuse_out_in = 1;                                                  // option: inside a block, conv L's output transform writes conv L+1's V directly (Y is never written): 0 never, 1 unless the row ranges would get too short, 2 always
    std::string fwd_v_layer;                                              // forward: layer whose V the previous layer's fused output transform has already written
    bool pool_routed[5] = {false, false, false, false, false};           // bf16_train: the forward pool of block b kept its routing bytes (pidx<b>) for maxpool_bwd_bf16_route_kernel
    std::set<std::string> dy_bf16_only;                                   // ... and layers whose fp32 OUTPUT GRADIENT was not written by this backward pass (their padded bf16 copy + the bias gradient were)
    std::set<std::string> in_bf16_only;                                   // bf16_train, option bf16_acts: layers whose fp32 INPUT was not written by the last training forward pass (their padded bf16 copy is all there is)
    std::set<std::string> y_unwritten;                                    // layers whose activation tensor was not materialised by the last forward pass
    int conv1_tiled = 1, conv1_wgrad_mfma = 1;                            // options: conv1_1 forward on the spatial-tile kernel / its weight gradient on the matrix core
    int conv1_in_transform = 1;                                           // option: conv1_1 is evaluated inside conv1_2's input transform (its activation tensor is never written)
    unsigned short* d_wbf16 = nullptr; size_t wbf16_elems = 0;            // bf16 copy of one layer's kernel at a time (K-tile-major or transposed)
    std::map<std::string, unsigned short*> wbf16_cache;                   // ... per layer, valid while frozen
    int bf16_gemm256 = 1;                                                 // bf16_fc mode: 256 x 256 LDS-DMA kernel -- 0 never, 1 when it fills the chip, 2 whenever shapes allow
    unsigned short* d_abf16 = nullptr; size_t abf16_elems = 0;            // bf16 copy of the layer's input activations
    // bf16 modes, training: zero-bordered padded bf16 copies of the inputs of conv3_1 .. conv5_3, one per layer (the border is written once, at
    // allocation; the interior every step by that layer's Winograd input transform, wino_input_kernel<.., XB>); keyed by layer, dropped with the workspace
    std::map<std::string, unsigned short*> xbf16;
    // FCN8S_PREC_BF16_TRAIN: per layer, the zero-bordered bf16 copy of its INPUT with zeroed guard rows in front and behind (bf16_guard_rows), written by
    // the forward pass and read again by the layer's weight gradient
    std::map<std::string, unsigned short*> xg16; std::map<std::string, size_t> xg16_elems;
    std::map<std::string, unsigned short*> dyg16; std::map<std::string, size_t> dyg16_elems;     // ... per layer: the same kind of copy of its output gradient dY
    std::set<std::string> db_taken;                                      // layers whose bias gradient the producer of their dY copy has already added (this backward pass)
    int bf16_fuse_pool = 1;                                               // option: bf16_train, the max-pool backward writes the last conv's bf16 dZ copy and bias gradient directly
    std::set<std::string> xg16_filled, dyg16_filled;                     // copies a producing kernel's epilogue has already written in this pass (no conversion pass)
    int bf16_infer_copies = 1;                                            // option: bf16_train's evaluation / prediction passes take the training pass's data flow (see forward())
    int bf16_rows_bn = 0;                                                 // option (A/B): 128 = the flat-position bf16 convolution takes its 128-column tile where it can (default: 64 columns)
    int bf16_acts = 1;                                                    // option: bf16_train training passes keep a conv -> conv activation only as the consumer's padded bf16 copy (the producer's epilogue writes it; no fp32 tensor, no conversion pass)
    int bf16_fuse_convert = 0;                                            // option: let the producing convolution write its consumer's bf16 copy (measured: the 2-byte epilogue stores cost more than the conversion passes they replace -- off)
    int saved_wino_min_cin = -1, saved_wino_fc6 = -1;                      // the options the mode overrides (the direct path carries it), restored on leaving
    int bf16_copy_by_transform = 1;                                       // option: 0 = every bf16 layer converts its input with a pass of its own (round 3's path)
    hipStream_t stream = nullptr;
    int64_t step = 0;
    // workspace for the current (N,H,W)
    int N = 0, H = 0, W = 0;
    int plan_N = 0;
    char* arena = nullptr; size_t arena_bytes = 0;
    std::map<std::string, Act> acts;
    // option "keep_output_gradients" (tests): every weighted layer's fp32 output gradient dY, copied as the backward pass hands it to the layer's weight gradient;
    // read back through fcn8s_get_activation("dy:<layer>")
    int keep_dy = 0; std::map<std::string, Act> kept_dy; std::set<std::string> dz_unwritten;      // dz_unwritten: layers whose fp32 dY the pool's backward kernel skipped
    // the last transposed conv (k = 2s = 16) as one GEMM over output blocks (PixMap, elementwise.hip): logits / dlogits live in that blocked layout
    int tconv_gemm = 1; PixMap pm{0, 0, 0, 0, 0, 0}; int tg_kp = 0;
    float *logits_b = nullptr, *dlogits_b = nullptr, *tg_A = nullptr, *tg_dA = nullptr;      // arena
    float *tg_b2 = nullptr, *tg_b2t = nullptr, *tg_bias = nullptr, *tg_db2 = nullptr;         // per model
    bool logits_nhwc_valid = false;
    float *dlogits = nullptr, *da3 = nullptr, *da4 = nullptr, *ds7 = nullptr, *gskip3 = nullptr, *gskip4 = nullptr;
    float *gbuf[2] = {nullptr, nullptr};
    int gcur = 0;
    void* d_images = nullptr; uint8_t* d_labels = nullptr;
    float *d_loss = nullptr, *d_regsum = nullptr, *d_softmax = nullptr;
    float* h_loss = nullptr; hipEvent_t loss_ev = nullptr; bool loss_copied = false;   // pinned copy of d_loss, queued right after the loss kernels
    float* d_lastbias = nullptr;       // column sums of dlogits (gradient of the last transposed conv's bias), produced by the loss kernel
    double* d_partials = nullptr; long long* d_pred = nullptr;
    unsigned long long* d_conf = nullptr;
    double loss_sum = 0; int64_t loss_cnt = 0;
    float keep_prob = 1.f, l2_rate = 0.f;
    uint32_t drop_stream = 0;
    bool have_forward = false, have_loss = false, train_mode = false;
    int next_bucket = 0;
    const uint8_t* cur_labels = nullptr;
    bool profile = false, profile_detail = false;
    // pinned host staging + device slots filled on a copy stream (fcn8s_stage_inputs): the next batch's H2D overlaps this step's kernels
    struct StageSlot { void* h_img = nullptr; uint8_t* h_lab = nullptr; void* d_img = nullptr; uint8_t* d_lab = nullptr;
                       size_t cap_img = 0, cap_lab = 0; hipEvent_t ready = nullptr, consumed = nullptr; bool used = false; };
    StageSlot slots[FCN8S_NUM_STAGE_SLOTS];
    hipStream_t copy_stream = nullptr;
    // Deferred weight gradients.  Nothing in the backward pass consumes a weight gradient, so the MFMA-bound weight-gradient GEMMs of the
    // deep layers (conv3_1 .. conv5_3 at level >= 1, fc6 / fc7 too at level 2) are held back and launched on `side` when the data-gradient
    // chain reaches block `defer_start_block`: from there on it is HBM-bound (Winograd transforms and K = 64 / 128 position GEMMs of blocks
    // 2 and 1), and the two kinds of work share the CUs.  Each deferred layer keeps its dM = A dY A^T in a buffer of its own.
    int defer_wgrad = 0, defer_start_block = 2;                          // measured zero-sum (profiles/r03_overlap_*.txt): off by default
    // Sharing CUs between the two kinds of work is zero-sum on gfx950 (profiles/r03_overlap_shared_cus.txt: both slow down by what the other
    // gains); on DISJOINT CUs they do not disturb each other at all (tools/labs/cumask_lab.hip).  defer_tail_cus = n > 0: from the start block on
    // the data-gradient chain moves to a stream restricted to the first n CUs and the held-back GEMMs run on the other 256 - n.
    int defer_tail_cus = 0;
    hipStream_t tail = nullptr; hipEvent_t tail_done = nullptr; bool on_tail = false;
    int defer_level_now = 0;                                             // level the running backward pass uses (the bucket API caps it at 1)
    hipStream_t side = nullptr; bool side_owned = false;
    float* d_wino_u2 = nullptr; size_t ufl = 0;                          // dU scratch of the side stream (ufl floats, like d_wino_u)
    std::vector<std::pair<hipEvent_t, std::function<void(hipStream_t)>>> deferred;   // (inputs-ready event on the main stream, launches)
    std::vector<hipEvent_t> ev_pool; size_t ev_next = 0;
    hipEvent_t side_done = nullptr;
    hipStream_t launch_stream = nullptr;                                 // != nullptr while deferred work is being enqueued: ProfScope records there
    float* dm_ptr = nullptr;                                             // where dm_layer's dM lives (d_wino_m or the layer's own buffer)
    std::vector<ProfGroup> groups;
    std::string err;
};

namespace {

int fail(fcn8s_model* m, int code, const std::string& msg)
{
    if (m) m->err = msg;
    g_last_error = msg;
    return code;
}
// what a launcher recorded with defer_error during the call that is about to return (0: nothing)
int deferred_rc(fcn8s_model* m)
{
    std::string t; const int c = take_deferred_error(&t);
    return c ? fail(m, c, t) : FCN8S_OK;
}
#define HIPCHK(m, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(m, e_ == hipErrorOutOfMemory ? FCN8S_ERR_OOM : FCN8S_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

const int kNumBuckets = 4;      // (what fcn8s_num_buckets reports; callers size nothing by a compile-time constant)

void build_param_table(int C, const int widths[7], int fc6k, std::vector<ParamInfo>& out, size_t& total,
                       size_t boff[FCN8S_MAX_BUCKETS], size_t bn[FCN8S_MAX_BUCKETS])
{
    out.clear();
    size_t off = 0;
    auto add = [&](const std::string& name, std::initializer_list<int64_t> shp) {
        ParamInfo p; p.name = name; p.ndim = (int)shp.size(); p.numel = 1;
        int i = 0; for (auto s : shp) { p.shape[i++] = s; p.numel *= (size_t)s; }
        for (; i < 4; ++i) p.shape[i] = 1;
        p.offset = off; off = align_up(off + p.numel, 64);   // 256-byte aligned tensors
        out.push_back(p);
    };
    int cin = 3;
    size_t conv4_start = 0, fc6_start = 0, fc7_start = 0;
    for (int b = 0; b < 5; ++b)
        for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
            char nm[32]; snprintf(nm, sizeof nm, "conv%d_%d", b + 1, i);
            if (b == 3 && i == 1) conv4_start = off;
            add(std::string(nm) + "/filter", {3, 3, cin, widths[b]});
            add(std::string(nm) + "/biases", {widths[b]});
            cin = widths[b];
        }
    fc6_start = off;
    add("fc6/weights", {fc6k, fc6k, widths[4], widths[5]});
    add("fc6/biases", {widths[5]});
    fc7_start = off;
    add("fc7/weights", {1, 1, widths[5], widths[6]});
    add("fc7/biases", {widths[6]});
    add("pool3_1x1/kernel", {1, 1, widths[2], C});
    add("pool3_1x1/bias", {C});
    add("pool4_1x1/kernel", {1, 1, widths[3], C});
    add("pool4_1x1/bias", {C});
    add("fc7_1x1/kernel", {1, 1, widths[6], C});
    add("fc7_1x1/bias", {C});
    add("fc7_conv2d_trans/kernel", {4, 4, C, C});
    add("fc7_conv2d_trans/bias", {C});
    add("fc7_pool4_conv2d_trans/kernel", {4, 4, C, C});
    add("fc7_pool4_conv2d_trans/bias", {C});
    add("fc7_pool4_pool3_conv2d_trans/kernel", {16, 16, C, C});
    add("fc7_pool4_pool3_conv2d_trans/bias", {C});
    total = off;
    // gradient buckets = contiguous slices in backward-production order: {fc7, decoder} (68 MB, final first) | {fc6} (411 MB) |
    // {conv4, conv5} (52 MB) | {conv1..conv3} (7 MB).  fc6 has a bucket of its own so that the exchange of the other 68 MB of the head does
    // not wait for the 3 ms of fc6's weight-gradient GEMMs, and fc6's own starts the moment they end (round 3 had one 479 MB bucket).
    for (int i = 0; i < FCN8S_MAX_BUCKETS; ++i) { boff[i] = 0; bn[i] = 0; }
    boff[0] = fc7_start;   bn[0] = total - fc7_start;
    boff[1] = fc6_start;   bn[1] = fc7_start - fc6_start;
    boff[2] = conv4_start; bn[2] = fc6_start - conv4_start;
    boff[3] = 0;           bn[3] = conv4_start;
}

void resolve_cfg(const fcn8s_config* cfg, int& C, int widths[7], int& fc6k)
{
    static const int def[7] = {64, 128, 256, 512, 512, 4096, 4096};
    C = cfg->num_classes;
    fc6k = cfg->fc6_ksize > 0 ? cfg->fc6_ksize : 7;
    for (int i = 0; i < 7; ++i) widths[i] = cfg->widths[i] > 0 ? cfg->widths[i] : def[i];
}

const ParamInfo& P(const fcn8s_model* m, const std::string& n) { return m->params[m->index.at(n)]; }
float* Wp(fcn8s_model* m, const std::string& n) { return m->d_params + P(m, n).offset; }
float* Gp(fcn8s_model* m, const std::string& n) { return m->d_grads + P(m, n).offset; }
float* WTp(fcn8s_model* m, const std::string& n) { return m->d_wt + P(m, n).offset; }

// ---- profiling wrappers -------------------------------------------------------
int group_id(fcn8s_model* m, const char* name)
{
    for (size_t i = 0; i < m->groups.size(); ++i) if (m->groups[i].name == name) return (int)i;
    ProfGroup g; g.name = name; m->groups.push_back(g);
    return (int)m->groups.size() - 1;
}
struct ProfScope {
    fcn8s_model* m; int gid = -1; hipEvent_t a = nullptr, b = nullptr; double fl = 0, by = 0;
    ProfScope(fcn8s_model* m_, const char* group, double flops, double bytes, const char* layer = nullptr) : m(m_), fl(flops), by(bytes)
    {
        if (!m->profile) return;
        fcn8s::g_last_kernel = nullptr;
        gid = (m->profile_detail && layer) ? group_id(m, (std::string(group) + ":" + layer).c_str()) : group_id(m, group);
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, m->launch_stream ? m->launch_stream : m->stream);
        m->groups[gid].flops += flops; m->groups[gid].bytes += bytes; m->groups[gid].launches += 1;
    }
    ~ProfScope()
    {
        if (gid < 0) return;
        hipEventRecord(b, m->launch_stream ? m->launch_stream : m->stream);
        m->groups[gid].ev.emplace_back(a, b);
        if (fcn8s::g_last_kernel) {          // second view of the same launch, keyed by the kernel symbol that ran
            const int k = group_id(m, (std::string("kernel:") + fcn8s::g_last_kernel).c_str());
            m->groups[k].flops += fl; m->groups[k].bytes += by; m->groups[k].launches += 1;
            m->groups[k].ev_shared.emplace_back(a, b);
        }
    }
};

// ---- layer launchers ------------------------------------------------------------
struct Epi { const float* bias = nullptr; const float* addend = nullptr; const float* mask = nullptr;
             float alpha = 1.f; int relu = 0; float mask_scale = 1.f; int dropout = 0; float keep = 1.f;
             uint32_t stream_id = 0; int dgrad = 0;      // dgrad: data-gradient launch (profile tag; never keeps V)
             float* pool_out = nullptr;                  // 2x2/2 max-pool of the output, written by the Winograd output transform if that path runs
             unsigned char* pool_idx = nullptr;          // ... with the per-window argmax bytes the backward pass routes the pool gradient by
             unsigned* relu_bits_out = nullptr;          // forward, Winograd path: also record (y > 0), one bit per element
             const unsigned* relu_bits_in = nullptr;     // data gradient, Winograd path: such a record of `mask` (read instead of the tensor)
             const float* w_fwd = nullptr;
             unsigned* in_relu_bits_out = nullptr;       // forward, Winograd path: record (x > 0) of the input too (see wino_input_kernel) ...
             const char* in_layer = nullptr;              // ... under this producer's name in rbits_ok
             int lazy_wt = 0;                            // data gradient: `w` is still to be filled from w_fwd (flip + transpose) if the adjoint path is not taken
             float* dm_out = nullptr; const char* dm_out_layer = nullptr;   // adjoint data gradient: write dM of the producing layer (name) here instead of its dZ into y
             float* next_v = nullptr; const char* next_layer = nullptr;   // forward, Winograd F(6x6) path: write the NEXT conv's V here instead of this conv's output
             const char* yb_layer = nullptr; int yb_K = 3; bool yb_only = false;    // (yb_only: both of that layer's gradients take its bf16 copy -- the fp32 tensor may stay unwritten) // bf16_train data gradient: the layer whose output gradient this launch produces (its bf16 copy is written on the way), and that layer's kernel size
             int skip_y = 0; };                          // Winograd path with pool_out: do not write the full-resolution output (only its pool is consumed)            // data gradient: the layer's forward kernel [3,3,Cout_of_this_conv... = Cin here][...] (adjoint Winograd path)

// 3x3 SAME conv through Winograd F(tile x tile, 3x3): filter transform, input transform, (tile+2)^2 batched GEMMs
// on the matrix cores (2.25x / 4x fewer MFMA flops than the direct form), output transform + fused epilogue.
// u: [P][Cin][Cout], v: [P][T][Cin], mm: [P][T][Cout] scratch, P = (tile+2)^2, T = N*(H/tile)*(W/tile).
// Output tile of the Winograd path for a K x K SAME conv on an H x W map (0 = none).  K = 7 (fc6): 4 (sub-filter decomposition).
// K = 3: F(6x6) [64 positions per 36 outputs, partial edge tiles] or F(4x4) [36 per 16, needs H, W % 4 == 0], whichever multiplies
// less on this map (small maps lose more to F(6x6)'s partial tiles than they gain); F(2x2) as the fallback.
// The adjoint data gradients read the forward filter bank of the same step as a transposed B operand (gemm_glds_nt_kernel).  That kernel
// only exists in the LDS-DMA form: K % 16 == 0 and whole N tiles of the width launch_igemm picks (64 for N = 64, else 128).  Other widths
// (e.g. 192) get a second, transposed bank instead (3x3 layers) or the forward-type data gradient (fc6).
static bool bt_gemm_ok(int K, int N) { return K % 16 == 0 && (N == 64 || N % 128 == 0); }
// Arithmetic of the op-level entry points, which have no model: the context of the CALLING THREAD (fcn8s_set_option(NULL, "op_split_pieces", n)
// from that thread), copied into the bare model each such call builds -- never read by a real model, never shared between threads.
thread_local int t_op_split = 0;
thread_local int t_op_rows_bn = 0;       // op context: the form of the flat-position 3 x 3 kernel fcn8s_op_conv2d_bf16_train asks for (option "op_bf16_rows_bn")
thread_local int t_op_planes = 1;        // op context: fcn8s_op_conv2d_bf16_train keeps its padded copies as channel-chunk planes (option "op_bf16_planes")
int split_of(const fcn8s_model* m)
{
    if (!m) return t_op_split;
    if (m->precision == FCN8S_PREC_F32X3 || m->precision == FCN8S_PREC_BF16_FWD) return 3;
    if (m->precision == FCN8S_PREC_F32X2 || m->precision == FCN8S_PREC_BF16_FWD_X2) return 2;
    return 0;
}
static inline bool bf16_fwd_mode(const fcn8s_model* m) { return m->precision == FCN8S_PREC_BF16_FWD || m->precision == FCN8S_PREC_BF16_FWD_X2; }
static inline bool bf16_train_mode(const fcn8s_model* m) { return m && m->precision == FCN8S_PREC_BF16_TRAIN; }
// guard rows of a padded bf16 copy that the weight-gradient kernel reads (gemm_bf16.hip): the largest tap shift + one K-tile of rounding
// (... and, for the nine-tap kernel, the eight-row instruction that completes a 34-row group: 128 rows cover all of it)
//  ... and conv_bf16_r64_kernel's last row tile, which runs up to 255 positions past the last one and reads 272 rows from one filter row above it: 640)
static inline long long bf16_guard_rows(int K, int Wp) { return (long long)((K - 1) / 2) * Wp + (K - 1) / 2 + 640; }
// bf16_train's per-layer padded copies are stored as channel-chunk PLANES [C / 32][guard + rows + guard][32] (Bf16Conv256Args): elements between two planes
#ifndef A_PLANES
#define A_PLANES 1                   // (0: the copies as [rows][C] -- the A/B switch of tools/ab_variants.sh; every kernel takes both)
#endif
static inline long long g16_ps(int N, int H, int W, int K) { const int pad = (K - 1) / 2, Wp_ = W + 2 * pad; return A_PLANES ? ((long long)N * (H + 2 * pad) * Wp_ + 2 * bf16_guard_rows(K, Wp_)) * 32 : 0; }
static inline long long g16_off(long long G, int C) { return A_PLANES ? G * 32 : G * C; }      // elements from the start of a copy's buffer to its padded pixel 0
unsigned short* g16_for(fcn8s_model* m, std::map<std::string, unsigned short*>& bufs, std::map<std::string, size_t>& sizes, const char* layer, int N, int H, int W, int C, int K, hipStream_t s);
unsigned short* dyb_for(fcn8s_model* m, const char* layer, const float* dy, int N, int H, int W, int C, int K, hipStream_t s, float* db = nullptr, bool* db_done = nullptr);

int wino_tile_for(const fcn8s_model* m, int H, int W, int K = 3)
{
    if (!m || H % 2 || W % 2) return 0;
    if (m->wino_force_tile && K == 3) return (m->wino_force_tile == 6 || (H % m->wino_force_tile == 0 && W % m->wino_force_tile == 0)) ? m->wino_force_tile : 0;
    int tmax = m->wino_tile;
    if (K == 3 && m->wino_tile_hires && m->wino_hires_pixels > 0 && (long long)H * W >= m->wino_hires_pixels && m->wino_tile_hires < tmax) tmax = m->wino_tile_hires;
    const bool t4 = tmax >= 4 && H % 4 == 0 && W % 4 == 0;
    if (K == 7) return (m->wino_tile >= 4 && H % 4 == 0 && W % 4 == 0) ? 4 : 0;
    if (tmax == 6) {
        // multiplies per channel pair = positions x GEMM rows.  For a single image the rows are rounded up to the 64-row GEMM tile: a 32x64
        // map has 66 F(6x6) tiles -- two row tiles, the second one nearly empty -- but exactly 128 F(4x4) tiles.  Batches of two or more
        // images are NOT treated this way: the arithmetic applied to an image must not depend on how many others share its batch (the
        // gradient of a batch equals the mean over its halves, data-parallel shards equal the big batch).
        auto rows = [&](long long tiles) { return m->plan_N == 1 ? (tiles + 63) / 64 * 64 : tiles; };
        const long long c6 = 64LL * rows((long long)((H + 5) / 6) * ((W + 5) / 6)), c4 = t4 ? 36LL * rows((long long)(H / 4) * (W / 4)) : 16LL * rows((long long)(H / 2) * (W / 2));
        if (c6 < c4) return 6;
    }
    return t4 ? 4 : 2;
}
long long wino_tiles(int tile, int N, int H, int W) { return (long long)N * ((H + tile - 1) / tile) * ((W + tile - 1) / tile); }
struct WinoEpi { const float* bias = nullptr; const float* addend = nullptr; const float* mask = nullptr; float mask_scale = 1.f;
                 int relu = 0; int dropout = 0; float keep = 1.f; unsigned long long seed = 0; unsigned int stream_id = 0; float* pool = nullptr; unsigned char* pidx = nullptr;
                 unsigned* rbits_out = nullptr; const unsigned* rbits_in = nullptr; int skip_y = 0;
                 unsigned* in_rbits_out = nullptr;       // ReLU bit record of the INPUT, written by the input transform
                 float* next_v = nullptr; bool* fused_out = nullptr; };     // output transform fused with the next conv's input transform (winograd.hip: wino_out_in_kernel)
// KS = 3, or 7 (3x3 grid of 3x3 sub-filters, GEMM depth 9*Cin -- see winograd.hip)
void conv_winograd(fcn8s_model* m, int tile, int KS, const char* tag, const float* x, const float* wk, float* y, float* u, float* v, float* mm,
                   int N, int H, int W, int Cin, int Cout, const WinoEpi& e, hipStream_t s, const char* layer, bool v_ready = false)
{
    const int P = wino_alpha(tile, KS) * wino_alpha(tile, KS), nsub2 = wino_nsub(KS) * wino_nsub(KS);
    const long long T = wino_tiles(tile, N, H, W);
    const int Kg = nsub2 * Cin;
    IgemmArgs a{}; a.split = split_of(m);
    a.x = v; a.w = u; a.y = mm;
    a.N = 1; a.Ma = (int)T; a.Mb = 1; a.M = T;
    a.Hi = (int)T; a.Wi = 1; a.Cin = Kg; a.ldx = Kg;
    a.KW = 1; a.in_scale = 1; a.tap_step = 1; a.tap_off = 0; a.Ktot = Kg;
    a.Ho = (int)T; a.Wo = 1; a.Cout = Cout; a.ldy = Cout;
    a.out_scale = 1; a.phases_x = 1; a.w_phase_stride = (long long)Kg * Cout;
    a.alpha = 1.f; a.mask_scale = 1.f;
    a.batched = 1; a.x_batch_stride = wino_slab(T, Kg); a.y_batch_stride = wino_slab(T, Cout);
    const double tb = 4.0 * ((double)N * H * W * Cin * nsub2 + (double)P * T * Kg), ob = 4.0 * ((double)N * H * W * Cout * ((e.pool ? (e.skip_y ? 0.25 : 1.25) : 1.0) + (e.rbits_in ? 1.0 / 32 : (e.mask ? 1.0 : 0.0)) + (e.addend ? 1.0 : 0.0) + (e.rbits_out ? 1.0 / 32 : 0.0)) + (double)P * T * Cout);   // y (+ pool) written, ReLU mask / skip addend read
    // frozen parameters (evaluate / predict loops): the transformed filter bank of each forward layer is computed once and kept
    bool u_cached = false;
    const bool fwd_call = std::string(tag).find("dgrad") == std::string::npos;      // (v_ready in a forward call = V written by the previous conv's fused output transform)
    if (m && m->frozen && layer && fwd_call) {
        float*& cu = m->u_cache[std::string(layer) + "#" + std::to_string(tile)];       // (the tile, hence the bank's shape, depends on the image size)
        if (cu) { u = cu; u_cached = true; }
        else if (hipMalloc((void**)&cu, (size_t)P * Kg * Cout * sizeof(float)) == hipSuccess) u = cu;       // filled below, reused from the next call on
        else { cu = nullptr; (void)hipGetLastError(); }
        a.w = u;
    }
    if (m && !u_cached && m->fwd_train && layer && fwd_call && ((KS == 3 && tile == 6) || (KS == 7 && tile == 4))) {
        float*& tu = m->u_train[std::string(layer) + "#" + std::to_string(tile)];
        if (!tu && hipMalloc((void**)&tu, (size_t)P * Kg * Cout * sizeof(float)) != hipSuccess) { tu = nullptr; (void)hipGetLastError(); }
        if (tu) { u = tu; a.w = u; }
    }
    // v_ready: V was written together with the weight gradient's dM by the fused transform (launch_wino_input_dout)
    auto pre = [&]() { if (!u_cached) launch_wino_filter(tile, wk, u, Cin, Cout, KS, s); if (!v_ready) launch_wino_input(tile, x, v, N, H, W, Cin, KS, s, e.in_rbits_out); };
    auto post = [&]() {
        if (e.next_v && tile == 6 && KS == 3 && e.relu && e.bias && !e.addend && !e.mask && !e.dropout && !e.pool && !e.rbits_in &&
            launch_wino_out_in(mm, e.bias, e.next_v, e.rbits_out, N, H, W, Cout, s, m && m->fuse_out_in >= 2)) { if (e.fused_out) *e.fused_out = true; return; }
        launch_wino_output(tile, mm, e.bias, e.addend, e.mask, e.mask_scale, e.relu, (e.skip_y && e.pool) ? nullptr : y, N, H, W, Cout, e.dropout, e.keep, e.seed, e.stream_id, s, e.pool, e.pidx, KS, e.rbits_out, e.rbits_in); };
    // (bytes of the fused form: M read, the next conv's V written, the ReLU record)
    const double ob_fused = 4.0 * (2.0 * P * T * Cout + (e.rbits_out ? (double)N * H * W * Cout / 32 : 0.0));
    if (m) {
        { ProfScope ps(m, "wino_transform", 0, (v_ready ? 0.0 : tb) + (u_cached ? 0.0 : (double)(KS * KS + P * nsub2) * 4 * Cin * Cout)); pre(); }
        { ProfScope ps(m, tag, 2.0 * P * T * Kg * Cout, 4.0 * P * (T * (double)(Kg + Cout) + (double)Kg * Cout), layer); launch_igemm(a, P, s); }
        { ProfScope ps(m, "wino_transform", 0, ob); post(); if (e.fused_out && *e.fused_out && m->profile && !m->groups.empty()) { const int g = group_id(m, "wino_transform"); m->groups[g].bytes += ob_fused - ob; } }
    } else { pre(); launch_igemm(a, P, s); post(); }
}

// SAME conv (or its data gradient when `w` holds flipped+transposed weights).  Returns true if e.pool_out was written.
bool conv_same(fcn8s_model* m, const char* group, const float* x, const float* w, float* y,
               int N, int H, int W, int Cin, int Cout, int K, const Epi& e, hipStream_t s, int real_cin = 0,
               const char* layer = nullptr)
{
    if (bf16_train_mode(m) && m->train_mode && e.w_fwd && layer && !real_cin && e.alpha == 1.f && !e.bias && !e.relu && !e.dropout && Cin % 32 == 0 && Cout % 64 == 0) {
        // FCN8S_PREC_BF16_TRAIN, data gradient of `layer` (here Cin = channels of dY, Cout = channels of dX): the SAME convolution of the padded bf16
        // copy of dY with the flipped kernel, wt[ci][(flipped tap, co)] bf16, on conv_bf16_256_kernel; fp32 accumulate, fp32 epilogue (skip-path addend,
        // the ReLU / dropout mask of the layer's input).  w_fwd is the forward kernel [K][K][Cout here][Cin here].
        const size_t wneed = (size_t)K * K * Cin * Cout;
        bool ok = true;
        if (m->wbf16_elems < wneed) {
            if (m->d_wbf16) { hipStreamSynchronize(s); hipFree(m->d_wbf16); m->d_wbf16 = nullptr; m->wbf16_elems = 0; }
            if (hipMalloc((void**)&m->d_wbf16, wneed * sizeof(unsigned short)) != hipSuccess) { (void)hipGetLastError(); ok = false; } else m->wbf16_elems = wneed;
        }
        unsigned short* dyb = ok ? dyb_for(m, layer, x, N, H, W, Cin, K, s) : nullptr;
        if (dyb) {
            { ProfScope ps(m, "weight_relayout", 0, 6.0 * wneed); launch_w_to_bf16_flip_t(e.w_fwd, m->d_wbf16, K, Cout, Cin, s); }
            Bf16Conv256Args g{};
            g.xp = dyb; g.xp_ps = g16_ps(N, H, W, K); g.wt = m->d_wbf16; g.y = y; g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.K = K;
            g.addend = e.addend; g.mask = e.mask; g.mask_scale = e.mask_scale; g.any_shape = 1; g.guarded = 1; g.rows_bn = m->bf16_rows_bn;
            if (e.mask && K == 3 && e.mask_scale == 1.f) {          // the mask is the ReLU of this layer's input: its sign is in the layer's own bf16 input copy
                auto xi = m->xg16.find(layer);
                if (xi != m->xg16.end() && xi->second) { g.mask16 = xi->second + g16_off(bf16_guard_rows(3, W + 2), Cout); g.mask16_ps = g16_ps(N, H, W, 3); }
            }
            // this gradient is the output gradient of layer e.yb_layer (same map): its padded bf16 copy is written by this kernel's epilogue
            const bool only16 = e.yb_layer && e.yb_only && m->bf16_acts && K == 3 && e.yb_K == 3 && Cout % 64 == 0;
            if (e.yb_layer && (m->bf16_fuse_convert || only16) && K == 3 && e.yb_K == 3) { g.yb = g16_for(m, m->dyg16, m->dyg16_elems, e.yb_layer, N, H, W, Cout, e.yb_K, s); g.yb_pad = (e.yb_K - 1) / 2; g.yb_ps = g16_ps(N, H, W, e.yb_K); }
            // ... and if that layer's gradients read nothing else (option bf16_acts), the fp32 gradient is not written: the epilogue also takes its column sums,
            // that layer's bias gradient, from the fp32 values
            long long prow = 0;
            if (g.yb && only16) {
                prow = ((long long)N * (H + 2) * (W + 2) + conv_bf16_rows_bm(Cout, g.rows_bn) - 1) / conv_bf16_rows_bm(Cout, g.rows_bn);
                g.colpart = det_scratch(s, (size_t)prow * Cout);
                if (g.colpart) g.y = nullptr; else prow = 0;
            }
            const double M = (double)N * H * W;
            bool done;
            { ProfScope ps(m, K == 1 ? "fc7_dgrad_bf16" : (K == 3 ? "conv3x3_dgrad_bf16" : "fc6_dgrad_bf16"), 2.0 * M * K * K * Cin * Cout,
                           (g.y ? 4.0 : 0.0) * M * Cout + (g.yb ? 2.0 : 0.0) * M * Cout + (g.mask16 ? 2.0 : (e.mask ? 4.0 : 0.0)) * M * Cout + 2.0 * M * Cin + 2.0 * wneed, layer);
              done = launch_conv_bf16_256(g, s); }
            if (done) {
                if (g.yb) m->dyg16_filled.insert(e.yb_layer);
                if (g.colpart) {
                    ProfScope ps(m, "colsum", 0, 4.0 * prow * Cout);
                    launch_colsum(g.colpart, Gp(m, std::string(e.yb_layer) + "/biases"), prow, Cout, s);
                    m->db_taken.insert(e.yb_layer); m->dy_bf16_only.insert(e.yb_layer);
                }
                return false;
            }
            if (g.colpart) { defer_error(FCN8S_ERR_SHAPE, "bf16_train: %s's data gradient was refused by the flat-position kernel", layer); return false; }
        }
    }
    if (bf16_train_mode(m) && m->train_mode && layer && e.dgrad && m->dy_bf16_only.count(layer)) {
        defer_error(FCN8S_ERR_SHAPE, "bf16_train: %s's data gradient could not run on the bf16 kernel and its fp32 output gradient was not kept (option \"bf16_acts\" = 0 keeps it)", layer); return false;
    }
    if (bf16_train_mode(m) && m->train_mode && layer && e.mask && m->in_bf16_only.count(layer)) {
        defer_error(FCN8S_ERR_SHAPE, "bf16_train: %s's data gradient could not run on the bf16 kernel and its fp32 mask was not kept (option \"bf16_acts\" = 0 keeps it)", layer); return false;
    }
    const bool wino3 = m && K == 3 && m->wino_min_cin > 0 && Cin >= m->wino_min_cin && m->d_wino_v && wino_tile_for(m, H, W, 3) && !e.dropout;
    const bool wino7 = m && K == 7 && m->wino_fc6 && m->d_wino_v && wino_tile_for(m, H, W, 7) == 4;
    if (m && wino3 && e.dgrad && layer && e.w_fwd && !m->dm_layer.empty() && m->dm_layer == layer && wino_tile_for(m, H, W, 3) == 6 &&
        Cin % 64 == 0 && Cout % 64 == 0 && e.alpha == 1.f && !real_cin && !e.bias && !e.relu) {
        // Data gradient as the adjoint of the forward Winograd algorithm: dV[xi] = dM[xi] U[xi]^T with the dM = A dY A^T the weight
        // gradient just built (d_wino_m) and the FORWARD filter bank, then dx = overlap-added B dV B^T.  (Here Cin = channels of dY,
        // Cout = channels of dx; w_fwd is [3,3,Cout,Cin].)
        m->dm_layer.clear(); m->fused_v_layer.clear();
        const int P = 64;
        const long long T = wino_tiles(6, N, H, W);
        IgemmArgs a{}; a.split = split_of(m);
        a.x = m->dm_ptr ? m->dm_ptr : m->d_wino_m; a.w = m->d_wino_u; a.y = m->d_wino_v;
        a.N = 1; a.Ma = (int)T; a.Mb = 1; a.M = T;
        a.Hi = (int)T; a.Wi = 1; a.Cin = Cin; a.ldx = Cin;
        a.KW = 1; a.in_scale = 1; a.tap_step = 1; a.tap_off = 0; a.Ktot = Cin;
        a.Ho = (int)T; a.Wo = 1; a.Cout = Cout; a.ldy = Cout;
        a.out_scale = 1; a.phases_x = 1; a.w_phase_stride = (long long)Cin * Cout;
        a.alpha = 1.f; a.mask_scale = 1.f;
        a.batched = 1; a.x_batch_stride = wino_slab(T, Cin); a.y_batch_stride = wino_slab(T, Cout);
        auto kept = m->u_train.find(std::string(layer) + "#6");
        if (kept != m->u_train.end() && kept->second && bt_gemm_ok(Cin, Cout)) {
            a.w = kept->second; a.bt = 1; a.ldw = Cin;          // the forward bank U[xi][ci_fwd = Cout here][co_fwd = Cin here], read transposed
        } else {
            ProfScope ps(m, "wino_transform", 0, (double)(9 + P) * 4 * Cin * Cout); launch_wino_filter(6, e.w_fwd, m->d_wino_u, Cout, Cin, 3, s, 1);
        }
        { ProfScope ps(m, "wino_gemm_dgrad", 2.0 * P * T * Cin * Cout, 4.0 * P * (T * (double)(Cin + Cout) + (double)Cin * Cout), layer); launch_igemm(a, P, s); }
        if (e.dm_out && e.dm_out_layer && e.relu_bits_in && !e.addend && e.mask_scale == 1.f) {
            // the consumer of this gradient is the previous conv's weight gradient in the Winograd domain: hand it dM, skip dZ
            ProfScope ps(m, "wino_transform", 0, 4.0 * ((double)N * H * W * Cout / 32 + 2.0 * P * T * Cout));
            launch_wino_dgrad_output_dout(m->d_wino_v, e.relu_bits_in, e.dm_out, N, H, W, Cout, s);
            m->dm_prefilled = e.dm_out_layer;
            return false;
        }
        { ProfScope ps(m, "wino_transform", 0, 4.0 * ((double)N * H * W * Cout * (1.0 + (e.relu_bits_in ? 1.0 / 32 : (e.mask ? 1.0 : 0.0)) + (e.addend ? 1.0 : 0.0)) + (double)P * T * Cout));
          launch_wino_dgrad_output(m->d_wino_v, e.addend, e.mask, e.mask_scale, e.relu_bits_in, y, N, H, W, Cout, s); }
        return false;
    }
    if (m && wino7 && e.dgrad && layer && !m->dm_layer.empty() && m->dm_layer == layer && e.alpha == 1.f && !real_cin && !e.bias && !e.relu && !e.mask && !e.addend &&
        bt_gemm_ok(Cin, 4 * Cout)) {
        auto kept = m->u_train.find(std::string(layer) + "#4");
        if (kept != m->u_train.end() && kept->second) {
            // fc6 data gradient as the adjoint of the forward sub-filter Winograd algorithm (here Cin = channels of dz = 4096, Cout = channels
            // of dx = 512): dV[xi][t][sub * Cout + c] = dM[xi][t][:] . U[xi][sub * Cout + c][:] with the forward bank of this step read as a
            // transposed B operand, then the overlap-add gather (winograd.hip).  No transform of dz, no second filter bank, no flipped copy.
            m->dm_layer.clear(); m->fused_v_layer.clear();
            const int P = 49, Ng = 4 * Cout;
            const long long T = wino_tiles(4, N, H, W);
            IgemmArgs a{}; a.split = split_of(m);
            a.x = m->dm_ptr ? m->dm_ptr : m->d_wino_m; a.w = kept->second; a.y = m->d_wino_v;
            a.N = 1; a.Ma = (int)T; a.Mb = 1; a.M = T;
            a.Hi = (int)T; a.Wi = 1; a.Cin = Cin; a.ldx = Cin;
            a.KW = 1; a.in_scale = 1; a.tap_step = 1; a.tap_off = 0; a.Ktot = Cin;
            a.Ho = (int)T; a.Wo = 1; a.Cout = Ng; a.ldy = Ng;
            a.out_scale = 1; a.phases_x = 1; a.w_phase_stride = (long long)Ng * Cin;
            a.alpha = 1.f; a.mask_scale = 1.f;
            a.batched = 1; a.x_batch_stride = wino_slab(T, Cin); a.y_batch_stride = wino_slab(T, Ng);
            a.bt = 1; a.ldw = Cin;
            { ProfScope ps(m, "wino_gemm_fc6_dgrad", 2.0 * P * T * Cin * Ng, 4.0 * P * (T * (double)(Cin + Ng) + (double)Cin * Ng), layer); launch_igemm(a, P, s); }
            { ProfScope ps(m, "wino_transform", 0, 4.0 * ((double)N * H * W * Cout + (double)P * T * Ng)); launch_wino_dgrad_output_sub44(m->d_wino_v, y, N, H, W, Cout, s); }
            return false;
        }
    }
    if (m) m->dm_layer.clear();
    if (e.lazy_wt && e.w_fwd) {
        if (m) { ProfScope ps(m, "weight_relayout", 0, 8.0 * K * K * Cin * Cout); launch_flip_transpose(e.w_fwd, const_cast<float*>(w), K * K, Cout, Cin, s); }
        else launch_flip_transpose(e.w_fwd, const_cast<float*>(w), K * K, Cout, Cin, s);
    }
    if ((wino3 || wino7) && Cin % 16 == 0 && Cout % 64 == 0 && e.alpha == 1.f && !real_cin) {
        const bool dgrad = e.dgrad != 0;
        float* vbuf = m->d_wino_v;
        const bool v_ready = (dgrad && layer && !m->fused_v_layer.empty() && m->fused_v_layer == layer) ||
                             (!dgrad && layer && !m->fwd_v_layer.empty() && m->fwd_v_layer == layer);      // (forward: written by the previous conv's fused output transform)
        m->fused_v_layer.clear(); m->fwd_v_layer.clear();
        if (!dgrad && layer) { auto it = m->acts.find(std::string("wv:") + layer); if (it != m->acts.end()) vbuf = it->second.p; }
        WinoEpi we; we.bias = e.bias; we.addend = e.addend; we.mask = e.mask; we.mask_scale = e.mask_scale; we.relu = e.relu;
        we.dropout = e.dropout; we.keep = e.keep; we.seed = m->seed; we.stream_id = e.stream_id; we.pool = e.pool_out; we.pidx = e.pool_idx;
        we.rbits_out = e.relu_bits_out; we.rbits_in = e.relu_bits_in; we.skip_y = e.skip_y && e.pool_out;
        if (e.relu_bits_out && layer) m->rbits_ok.insert(layer);
        if (e.in_relu_bits_out && e.in_layer && K == 3 && !dgrad) { we.in_rbits_out = e.in_relu_bits_out; m->rbits_ok.insert(e.in_layer); }
        const char* tag = K == 7 ? (dgrad ? "wino_gemm_fc6_dgrad" : "wino_gemm_fc6_fwd") : (dgrad ? "wino_gemm_dgrad" : "wino_gemm_fwd");
        bool fused_out = false;
        if (!dgrad && e.next_v && e.next_layer) { we.next_v = e.next_v; we.fused_out = &fused_out; }
        conv_winograd(m, wino_tile_for(m, H, W, K), K, tag, x, w, y, m->d_wino_u, vbuf, m->d_wino_m, N, H, W, Cin, Cout, we, s, layer, v_ready);
        if (fused_out) { m->fwd_v_layer = e.next_layer; if (layer) m->y_unwritten.insert(layer); }
        return e.pool_out != nullptr;
    }
    if (m) m->fused_v_layer.clear();
    IgemmArgs a{}; a.split = split_of(m);
    a.x = x; a.w = w; a.bias = e.bias; a.addend = e.addend; a.mask = e.mask; a.y = y;
    a.N = N; a.Ma = H; a.Mb = W; a.M = (long long)N * H * W;
    a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldx = Cin;
    a.KW = K; a.in_scale = 1; a.tap_step = 1; a.tap_off = -(K - 1) / 2; a.Ktot = K * K * Cin;
    a.Ho = H; a.Wo = W; a.Cout = Cout; a.ldy = Cout;
    a.out_scale = 1; a.out_offy = 0; a.out_offx = 0; a.phases_x = 1; a.w_phase_stride = 0;
    a.alpha = e.alpha; a.relu = e.relu; a.mask_scale = e.mask_scale;
    a.dropout = e.dropout; a.keep_prob = e.keep; a.seed = m ? m->seed : 0; a.stream_id = e.stream_id;
    const double rc = real_cin ? real_cin : Cin;
    const double flops = 2.0 * a.M * K * K * rc * Cout;
    const double bytes = 4.0 * (a.M * rc + (double)a.M * Cout + (double)K * K * rc * Cout);
    // 1x1 score heads: one skinny dimension (skinny.hip); everything else goes through the general tile kernels
    auto run = [&]() {
        if (K == 1 && !real_cin && !e.addend && !e.relu && !e.dropout) {
            if (Cout <= 32 && !e.mask && launch_head_fwd(x, w, e.bias, y, a.M, Cin, Cout, e.alpha, s, m ? m->d_wino_u : nullptr, m ? m->ufl : 0)) return;      // (the filter-bank scratch is idle between convolutions)
            if (Cin <= 32 && !e.bias && launch_head_dgrad(x, w, e.mask, e.mask_scale, y, a.M, Cout, Cin, e.alpha, s)) return;
        }
        launch_igemm(a, 1, s);
    };
    if (m) { ProfScope ps(m, group, flops, bytes, layer); run(); }
    else run();
    return false;
}

// transposed conv forward (k = 2s) as s*s phase-specific 2x2 convs; wp = phase-packed weights
void tconv_fwd(fcn8s_model* m, const float* x, const float* wp, const float* bias, const float* addend,
               float* y, int N, int Hi, int Wi, int C, int K, int S, hipStream_t s)
{
    IgemmArgs a{}; a.split = split_of(m);
    a.x = x; a.w = wp; a.bias = bias; a.addend = addend; a.mask = nullptr; a.y = y;
    a.N = N; a.Ma = Hi + 1; a.Mb = Wi + 1; a.M = (long long)N * a.Ma * a.Mb;
    a.Hi = Hi; a.Wi = Wi; a.Cin = C; a.ldx = C;
    a.KW = 2; a.in_scale = 1; a.tap_step = -1; a.tap_off = 0; a.Ktot = 4 * C;
    a.Ho = Hi * S; a.Wo = Wi * S; a.Cout = C; a.ldy = C;
    a.out_scale = S; a.out_offy = -(K - S) / 2; a.out_offx = -(K - S) / 2;
    a.phases_x = S; a.w_phase_stride = 4LL * C * C;
    a.alpha = 1.f; a.relu = 0; a.mask_scale = 1.f; a.dropout = 0;
    const double opix = (double)N * Hi * S * Wi * S;
    if (m) { ProfScope ps(m, "tconv_fwd", 2.0 * opix * 4 * C * C, 4.0 * (opix * C * (addend ? 2 : 1) + (double)N * Hi * Wi * C)); launch_igemm(a, S * S, s); }
    else launch_igemm(a, S * S, s);
}

// data gradient of the transposed conv = stride-S conv of dy with the natural [kh,kw,Cout,Cin] kernel
void tconv_dgrad(fcn8s_model* m, const float* dy, const float* w, float* dx, int N, int Hi, int Wi, int C,
                 int K, int S, hipStream_t s)
{
    IgemmArgs a{}; a.split = split_of(m);
    a.x = dy; a.w = w; a.y = dx;
    a.N = N; a.Ma = Hi; a.Mb = Wi; a.M = (long long)N * Hi * Wi;
    a.Hi = Hi * S; a.Wi = Wi * S; a.Cin = C; a.ldx = C;
    a.KW = K; a.in_scale = S; a.tap_step = 1; a.tap_off = -(K - S) / 2; a.Ktot = K * K * C;
    a.Ho = Hi; a.Wo = Wi; a.Cout = C; a.ldy = C;
    a.out_scale = 1; a.phases_x = 1; a.alpha = 1.f; a.mask_scale = 1.f;
    const double flops = 2.0 * a.M * K * K * C * C;
    if (m) { ProfScope ps(m, "tconv_dgrad", flops, 4.0 * ((double)N * Hi * S * Wi * S * C + (double)a.M * C)); launch_igemm(a, 1, s); }
    else launch_igemm(a, 1, s);
}

// Deferred weight gradients (fcn8s_model::defer_wgrad): helpers
hipEvent_t defer_event(fcn8s_model* m)
{
    if (m->ev_next == m->ev_pool.size()) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); m->ev_pool.push_back(e); }
    return m->ev_pool[m->ev_next++];
}
// CU-masked streams are made once per (device, mask) and never destroyed (destroying one and creating another hung on ROCm 7.2 in
// tools/labs/cumask_lab.hip); models of one process share them, which only serialises their held-back work
hipStream_t masked_stream(int device, int first, int count)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, hipStream_t> pool;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple(device, first, count);
    auto it = pool.find(key);
    if (it != pool.end()) return it->second;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const int ncu = p.multiProcessorCount;
    if (first < 0 || count <= 0 || first + count > ncu) return nullptr;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c = first; c < first + count; ++c) mask[c / 32] |= 1u << (c % 32);
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
    pool[key] = st;
    return st;
}
int device_cus(int device)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return p.multiProcessorCount;
}

bool defer_ready(fcn8s_model* m)
{
    if (!m->side) {
        const int ncu = device_cus(m->device);
        if (m->defer_tail_cus > 0 && m->defer_tail_cus < ncu) {
            m->tail = masked_stream(m->device, 0, m->defer_tail_cus);
            m->side = m->tail ? masked_stream(m->device, m->defer_tail_cus, ncu - m->defer_tail_cus) : nullptr;
            if (!m->side) m->tail = nullptr;
            m->side_owned = false;
        }
        if (!m->side) {
            int lo = 0, hi = 0;
            hipDeviceGetStreamPriorityRange(&lo, &hi);             // lo = numerically greatest = lowest priority: the chain on the main stream goes first
            if (hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, lo) != hipSuccess) { m->side = nullptr; (void)hipGetLastError(); return false; }
            m->side_owned = true;
        }
        hipEventCreateWithFlags(&m->side_done, hipEventDisableTiming);
        hipEventCreateWithFlags(&m->tail_done, hipEventDisableTiming);
    }
    if (!m->d_wino_u2 && hipMalloc((void**)&m->d_wino_u2, m->ufl * sizeof(float)) != hipSuccess) { m->d_wino_u2 = nullptr; (void)hipGetLastError(); return false; }
    return true;
}
// launch everything held back so far on the side stream (each item waits for the event that marks its inputs ready)
void flush_deferred(fcn8s_model* m, hipEvent_t here = nullptr)
{
    if (m->deferred.empty()) return;
    // the host enqueues far ahead of the GPU: without this event the side stream would start each item as soon as its inputs exist, i.e. beside
    // the MFMA-bound data-gradient GEMMs of the deep layers, which gains nothing.  It has to wait until the MAIN stream gets here.
    if (!here) { here = defer_event(m); hipEventRecord(here, m->on_tail ? m->tail : m->stream); }
    hipStreamWaitEvent(m->side, here, 0);
    hipStream_t was = m->launch_stream;
    m->launch_stream = m->side;
    for (auto& d : m->deferred) { hipStreamWaitEvent(m->side, d.first, 0); d.second(m->side); }
    m->launch_stream = was;
    m->deferred.clear();
}
// end of the backward pass: the main stream continues only after the side stream has drained
void join_deferred(fcn8s_model* m)
{
    flush_deferred(m);
    if (m->on_tail) { hipEventRecord(m->tail_done, m->tail); hipStreamWaitEvent(m->stream, m->tail_done, 0); m->on_tail = false; m->launch_stream = nullptr; }
    if (m->side && m->ev_next) { hipEventRecord(m->side_done, m->side); hipStreamWaitEvent(m->stream, m->side_done, 0); }
    m->ev_next = 0;
}

// phase: 0 = everything on stream s; 1 = only what the data gradient needs (the transform of dz into dM) -- the weight gradient itself
// is held back; 2 = the held-back part (GEMM, filter-gradient transform, bias gradient) on stream s.
void conv_wgrad(fcn8s_model* m, const char* group, const float* x, const float* dz, float* dw, float* db,
                int N, int H, int W, int Cin, int Cout, int K, float alpha, hipStream_t s, int real_cin = 0,
                const char* layer = nullptr, bool fuse_dgrad_input = false, const unsigned char* pool_idx = nullptr, int phase = 0)
{
    // the data gradient of the layer after this one may have written this layer's dM instead of dz (backward_blocks): dz then holds nothing
    const bool promised = m && layer && phase != 2 && !m->dm_prefilled.empty() && m->dm_prefilled == layer;
    if (m && phase != 2) m->dm_prefilled.clear();
    if (m && m->keep_dy && layer && phase != 2) {
        // (dz holds the layer's fp32 dY unless it was handed over in another form: dM from the next layer's data gradient, d(pool) with routing bytes, a bf16 copy only)
        Act& k = m->kept_dy[layer];
        const size_t n = (size_t)N * H * W * Cout;
        if (promised || pool_idx || m->dy_bf16_only.count(layer) || m->dz_unwritten.count(layer)) k.n = 0;
        else {
            if (k.p && k.H != (int)(n >> 20)) { hipStreamSynchronize(s); hipFree(k.p); k.p = nullptr; }
            if (!k.p && hipMalloc((void**)&k.p, n * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); k.p = nullptr; defer_error(FCN8S_ERR_OOM, "keep_output_gradients: %s's copy cannot be allocated", layer); }
            if (k.p) { k.n = n; k.H = (int)(n >> 20); hipMemcpyAsync(k.p, dz, n * sizeof(float), hipMemcpyDeviceToDevice, s); }
        }
    }
    auto broken_promise = [&]() { defer_error(FCN8S_ERR_STATE, "%s was handed dM instead of dZ but does not take the Winograd-domain path", layer); };
    WgradArgs a{}; a.split = split_of(m);
    a.A = x; a.B = dz; a.C = dw;
    a.N = N; a.Pa = H; a.Pb = W; a.P = (long long)N * H * W;
    a.Ha = H; a.Wa = W; a.Adim = Cin; a.lda = Cin; a.Areal = real_cin ? real_cin : Cin;
    a.Bdim = Cout; a.ldb = Cout;
    a.KW = K; a.a_scale = 1; a.tap_off = -(K - 1) / 2; a.ntaps = K * K;
    a.ldc = Cout; a.alpha = alpha; a.colsum = db;
    const double rc = a.Areal;
    const double flops = 2.0 * a.P * K * K * rc * Cout;
    const double bytes = 4.0 * (a.P * rc + (double)a.P * Cout + (double)K * K * rc * Cout);
    if (bf16_train_mode(m) && m->train_mode && layer && !real_cin && alpha == 1.f && Cin % 64 == 0 && Cout % 64 == 0) {
        // FCN8S_PREC_BF16_TRAIN: dW[tap] = (padded bf16 input, moved by the tap)^T (padded bf16 dY), fp32 accumulate (gemm_bf16.hip: wgrad_bf16_kernel);
        // the bias gradient is the exact fp32 column sum of dY
        auto it = m->xg16.find(layer);
        if (it != m->xg16.end() && it->second) {
            const int pad = (K - 1) / 2, Wp_ = W + 2 * pad;
            const long long G = bf16_guard_rows(K, Wp_), R = (long long)N * (H + 2 * pad) * Wp_;
            bool db_done = false;
            unsigned short* dyb = dyb_for(m, layer, dz, N, H, W, Cout, K, s, db, &db_done);
            if (m->db_taken.count(layer)) db_done = true;          // (the kernel that wrote this layer's dY copy added the bias gradient too)
            if (dyb) {
                Bf16WgradArgs g{};
                g.A = it->second + g16_off(G, Cin); g.B = dyb; g.C = dw; g.R = R; g.Ci = Cin; g.Cj = Cout; g.K = K; g.Wp = Wp_; g.a_ps = g.b_ps = g16_ps(N, H, W, K);
                bool done;
                { ProfScope ps(m, K == 1 ? "fc7_wgrad_bf16" : (K == 3 ? "conv3x3_wgrad_bf16" : "fc6_wgrad_bf16"), flops, 2.0 * K * K * R * (Cin + Cout) + 4.0 * K * K * Cin * Cout, layer);
                  done = launch_wgrad_bf16(g, s); }
                if (done) {
                    if (db && !db_done) { ProfScope ps(m, "colsum", 0, 4.0 * a.P * Cout); launch_colsum(dz, db, a.P, Cout, s); }
                    return;
                }
                if (db_done) { defer_error(FCN8S_ERR_SHAPE, "bf16_train: %s's weight-gradient launch refused its shape after the bias gradient was taken", layer); return; }
            }
        }
    }
    if (bf16_train_mode(m) && m->train_mode && layer && (m->in_bf16_only.count(layer) || m->dy_bf16_only.count(layer))) {
        defer_error(FCN8S_ERR_SHAPE, "bf16_train: %s's weight gradient could not run on the bf16 kernel and its fp32 input was not kept (option \"bf16_acts\" = 0 keeps it)", layer); return;
    }
    if (m && (K == 3 || K == 7) && layer && alpha == 1.f && !real_cin && m->train_mode) {
        auto it = m->acts.find(std::string("wv:") + layer);
        if (it != m->acts.end() && m->d_wino_m) {             // weight gradient in the Winograd domain (V kept by the forward pass)
            const int tile = wino_tile_for(m, H, W, K), NP = wino_alpha(tile, K) * wino_alpha(tile, K);
            const long long T = wino_tiles(tile, N, H, W);
            const int Kg = wino_nsub(K) * wino_nsub(K) * Cin;           // rows of V / dU: [sub-filter][channel]
            // a deferred layer keeps its dM in a buffer of its own ("dmk:<layer>", ensure_workspace) and its dU scratch on the side stream
            float* dmbuf = m->d_wino_m; float* dubuf = m->d_wino_u;
            if (phase) { dmbuf = m->acts.at(std::string("dmk:") + layer).p; dubuf = m->d_wino_u2; }
            WgradArgs g{}; g.split = split_of(m);
            g.A = it->second.p; g.B = dmbuf; g.C = dubuf;
            g.N = 1; g.Pa = 1; g.Pb = (int)T; g.P = T;
            g.Ha = 1; g.Wa = (int)T; g.Adim = Kg; g.lda = Kg; g.Areal = Kg;
            g.Bdim = Cout; g.ldb = Cout; g.KW = 1; g.a_scale = 1; g.tap_off = 0; g.ntaps = NP; g.ldc = Cout; g.alpha = 1.f; g.colsum = nullptr;
            g.batched = 1; g.a_batch_stride = wino_slab(T, Kg); g.b_batch_stride = wino_slab(T, Cout); g.c_uninitialized = 1;
            // fuse_dgrad_input: the data gradient of this layer follows and runs through Winograd too -- its input
            // transform V = B^T dz B is written into d_wino_v by the same kernel that writes dM (one read of dz)
            bool fused = false, dm_ready = false;
            const bool adj_bytes = fuse_dgrad_input && tile == 6 && K == 3 && Cin % 64 == 0 && Cout % 64 == 0;
            const bool prefilled = promised && phase == 0 && adj_bytes && !pool_idx;      // dM already in d_wino_m (wino_dgrad_output_dout_kernel)
            if (promised && !prefilled) broken_promise();
            if (prefilled) dm_ready = true;
            else if (phase != 2) {
            { ProfScope ps(m, "wino_transform", 0, 4.0 * ((double)N * H * W * Cout * (pool_idx ? 0.3125 : 1.0) + ((fuse_dgrad_input && !adj_bytes) ? 2.0 : 1.0) * NP * T * Cout));
              // pool_idx: dz is d(pool) [N,H/2,W/2,Cout]; the max-pool backward happens inside the transform (the caller checked eligibility)
              // adjoint data gradient (tile 6): it consumes dM itself, no second transform of dz
              if (adj_bytes) { launch_wino_dout(6, dz, dmbuf, N, H, W, Cout, s, 3, pool_idx); dm_ready = true; }
              else {
                  if (fuse_dgrad_input && tile >= 4 && K == 3) fused = launch_wino_input_dout(tile, dz, m->d_wino_v, dmbuf, N, H, W, Cout, s, pool_idx);
                  if (!fused) launch_wino_dout(tile, dz, dmbuf, N, H, W, Cout, s, K);
              } }
            // fc6: the non-fused transform above left dM = A dz A^T in dmbuf; its adjoint data gradient (conv_same) consumes it
            if (K == 7 && tile == 4 && !fused && Cin % 2 == 0 && bt_gemm_ok(Cout, 4 * Cin) && m->u_train.count(std::string(layer) + "#4")) dm_ready = true;
            }
            if (phase != 2) {
            m->fused_v_layer = fused ? layer : "";
            m->dm_layer = dm_ready ? layer : "";
            m->dm_ptr = dmbuf;
            }
            if (phase == 1) return;
            { ProfScope ps(m, K == 7 ? "wino_gemm_fc6_wgrad" : "wino_gemm_wgrad", 2.0 * NP * T * Kg * Cout, 4.0 * NP * (T * (double)(Kg + Cout) + (double)Kg * Cout), layer); launch_wgrad(g, s); }
            { ProfScope ps(m, "wino_transform", 0, 4.0 * (9.0 + NP) * Cin * Cout + 4.0 * N * H * W * Cout * (tile >= 4 ? 1.0 / (tile * tile) : 1.0));
              launch_wino_dfilter(tile, dubuf, dw, Cin, Cout, K, s);
              // bias gradient = sum of dz over all pixels.  dM[(1,1)] = sum_kl A^T(k,1) dz[k][l] A^T(l,1) and column 1 of A^T is all ones:
              // the slab of position (1,1) holds the per-tile sums -- 16x fewer bytes than dz, and dz need not exist
              if (db) {
                  if (tile >= 4) launch_colsum(dmbuf + (wino_alpha(tile, K) + 1) * wino_slab(T, Cout), db, T, Cout, s);
                  else launch_colsum(dz, db, (long long)N * H * W, Cout, s);
              } }
            return;
        }
    }
    if (promised) broken_promise();
    const bool taps = (K == 3 || K == 7) && alpha == 1.f && !real_cin;
    const bool first = K == 3 && alpha == 1.f && real_cin == 3 && Cin == 4;
    auto run = [&]() {
        if (K == 1 && !real_cin && Cout <= 32 && launch_head_wgrad(x, dz, dw, a.P, Cin, Cout, alpha, s)) {
            if (db) launch_colsum(dz, db, a.P, Cout, s);
            return;
        }
        if (taps && launch_wgrad_taps(x, dz, dw, db, N, H, W, Cin, Cout, K, s)) return;
        if (first && launch_conv1_wgrad(x, dz, dw, db, N, H, W, Cout, m->conv1_wgrad_mfma, s)) return;
        launch_wgrad(a, s);
    };
    if (m) { ProfScope ps(m, group, flops, bytes, layer); run(); }
    else run();
}

void tconv_wgrad(fcn8s_model* m, const float* x, const float* dy, float* dw, int N, int Hi, int Wi, int C,
                 int K, int S, hipStream_t s)
{
    WgradArgs a{}; a.split = split_of(m);
    a.A = dy; a.B = x; a.C = dw;
    a.N = N; a.Pa = Hi; a.Pb = Wi; a.P = (long long)N * Hi * Wi;
    a.Ha = Hi * S; a.Wa = Wi * S; a.Adim = C; a.lda = C; a.Areal = C;
    a.Bdim = C; a.ldb = C;
    a.KW = K; a.a_scale = S; a.tap_off = -(K - S) / 2; a.ntaps = K * K;
    a.ldc = C; a.alpha = 1.f; a.colsum = nullptr;
    auto run = [&]() { if (!launch_tconv_wgrad(x, dy, dw, N, Hi, Wi, C, K, S, s)) launch_wgrad(a, s); };
    if (m) { ProfScope ps(m, "tconv_wgrad", 2.0 * a.P * K * K * C * C, 4.0 * ((double)N * Hi * S * Wi * S * C + (double)a.P * C)); run(); }
    else run();
}

// ---- workspace --------------------------------------------------------------------
int ensure_workspace(fcn8s_model* m, int N, int H, int W)
{
    if (N <= 0) return fail(m, FCN8S_ERR_SHAPE, "batch size must be positive");
    if (H <= 0 || W <= 0 || H % 32 || W % 32)
        return fail(m, FCN8S_ERR_SHAPE, "image height and width must be positive multiples of 32 (five 2x2 pools, then x2, x2, x8 upsampling must line up with the skip connections)");
    if (m->arena && m->N == N && m->H == H && m->W == W) return FCN8S_OK;
    if (m->arena) { hipStreamSynchronize(m->stream); hipFree(m->arena); m->arena = nullptr; }
    for (auto& kv : m->xbf16) if (kv.second) hipFree(kv.second);
    m->xbf16.clear();
    for (auto& kv : m->xg16) if (kv.second) hipFree(kv.second);
    m->xg16.clear(); m->xg16_elems.clear();
    for (auto& kv : m->dyg16) if (kv.second) hipFree(kv.second);
    m->dyg16.clear(); m->dyg16_elems.clear(); m->xg16_filled.clear(); m->dyg16_filled.clear();
    m->plan_N = N;                 // the batch size the per-layer Winograd tiles are chosen for (wino_tile_for), from here until the next re-plan
    m->acts.clear();
    m->have_forward = m->have_loss = false;
    const int C = m->C;
    struct Item { std::string name; size_t n; int h, w, c; float** extra; };
    std::vector<Item> items;
    auto add = [&](const std::string& nm, int h, int w, int c, float** extra = nullptr) {
        items.push_back({nm, (size_t)N * h * w * c, h, w, c, extra});
    };
    add("x0", H, W, 4);
    int h = H, w = W;
    for (int b = 0; b < 5; ++b) {
        for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
            char nm[32]; snprintf(nm, sizeof nm, "conv%d_%d", b + 1, i);
            add(nm, h, w, m->widths[b]);
        }
        h /= 2; w /= 2;
        char nm[32]; snprintf(nm, sizeof nm, "pool%d", b + 1);
        add(nm, h, w, m->widths[b]);
    }
    const int h5 = H / 32, w5 = W / 32, h4 = H / 16, w4 = W / 16, h3 = H / 8, w3 = W / 8;
    {   // ReLU bit masks of the convs whose output is another conv's input (read back by that conv's data gradient)
        int cin = 3;
        for (int b = 0, hh = H, ww = W; b < 5; ++b, hh /= 2, ww /= 2)
            for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
                const int tile = wino_tile_for(m, hh, ww, 3);
                // (conv1_1 is not a Winograd layer: its record is written by conv1_2's input transform, if that is one)
                const bool by_consumer = b == 0 && i == 1 && kConvsPerBlock[0] > 1 && m->wino_min_cin > 0 && m->widths[0] >= m->wino_min_cin && m->widths[0] % 64 == 0 && tile;
                if (by_consumer || (i < kConvsPerBlock[b] && m->wino_min_cin > 0 && cin >= m->wino_min_cin && cin % 16 == 0 && m->widths[b] % 64 == 0 && tile)) {
                    char nm[40]; snprintf(nm, sizeof nm, "rb:conv%d_%d", b + 1, i);
                    items.push_back({nm, wino_rbits_words(tile, N, hh, ww, m->widths[b]), 0, 0, 0, nullptr});
                }
                cin = m->widths[b];
            }
    }
    for (int b = 0; b < 5; ++b) {      // argmax bytes of pool_b (one per pooled element), see launch_wino_output
        char nm[16]; snprintf(nm, sizeof nm, "pidx%d", b + 1);
        items.push_back({nm, ((size_t)N * (H >> (b + 1)) * (W >> (b + 1)) * (size_t)m->widths[b] + 3) / 4, 0, 0, 0, nullptr});
    }
    add("fc6", h5, w5, m->widths[5]);
    add("fc7", h5, w5, m->widths[6]);
    add("s7", h5, w5, C); add("p4", h4, w4, C); add("p3", h3, w3, C);
    add("a4", h4, w4, C); add("a3", h3, w3, C); add("logits", H, W, C);
    add("dlogits", H, W, C, &m->dlogits);
    m->pm = PixMap{0, H, W, H / 8 + 1, W / 8 + 1, 8};
    m->logits_b = m->dlogits_b = m->tg_A = m->tg_dA = nullptr;
    if (m->tconv_gemm) {
        const size_t rows = (size_t)N * m->pm.QH * m->pm.QW;
        m->pm.blocked = 1;
        items.push_back({"logits_b", rows * 64 * C, 0, 0, 0, &m->logits_b});
        items.push_back({"dlogits_b", rows * 64 * C, 0, 0, 0, &m->dlogits_b});
        items.push_back({"tg_A", rows * m->tg_kp, 0, 0, 0, &m->tg_A});
        items.push_back({"tg_dA", rows * m->tg_kp, 0, 0, 0, &m->tg_dA});
    }
    add("da3", h3, w3, C, &m->da3); add("da4", h4, w4, C, &m->da4); add("ds7", h5, w5, C, &m->ds7);
    add("gskip3", h3, w3, m->widths[2], &m->gskip3); add("gskip4", h4, w4, m->widths[3], &m->gskip4);
    size_t gmax = (size_t)N * H * W * m->widths[0];
    for (int b = 0, hh = H, ww = W; b < 5; ++b, hh /= 2, ww /= 2) {
        size_t n = (size_t)N * hh * ww * m->widths[b]; if (n > gmax) gmax = n;
    }
    { size_t n = (size_t)N * h5 * w5 * (size_t)std::max(m->widths[5], m->widths[6]); if (n > gmax) gmax = n; }
    items.push_back({"gbuf0", gmax, 0, 0, 0, &m->gbuf[0]});
    items.push_back({"gbuf1", gmax, 0, 0, 0, &m->gbuf[1]});
    items.push_back({"softmax", (size_t)N * H * W * C, H, W, C, &m->d_softmax});
    {   // Winograd scratch: the largest [P][T][C] tensor any layer needs (V and M of the forward and of the data-gradient conv)
        auto slab_floats = [&](int hh, int ww, int c, int K) -> size_t {      // P slabs of wino_slab(T, nsub^2 c) floats
            const int tile = wino_tile_for(m, hh, ww, K);
            if (!tile) return 0;
            const int al = wino_alpha(tile, K), ns = wino_nsub(K);
            return (size_t)al * al * (size_t)wino_slab(wino_tiles(tile, N, hh, ww), ns * ns * c);
        };
        size_t vmax = 0;
        if (m->wino_min_cin > 0) {
            int cin = 3;
            for (int b = 0, hh = H, ww = W; b < 5; ++b, hh /= 2, ww /= 2)
                for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
                    const int cout = m->widths[b];
                    if (std::max(cin, cout) >= m->wino_min_cin) vmax = std::max(vmax, slab_floats(hh, ww, std::max(cin, cout), 3));
                    cin = cout;
                }
        }
        const int h5_ = H / 32, w5_ = W / 32;
        const bool fc6w = m->wino_fc6 && m->fc6k == 7 && wino_tile_for(m, h5_, w5_, 7) == 4 && m->widths[4] % 16 == 0 && m->widths[5] % 64 == 0;
        if (fc6w) {     // V: P * T * nsub^2 * c5;  M: P * T * c6  (the data gradient swaps the two roles; M has nsub = 1)
            const int al = wino_alpha(4, 7);
            vmax = std::max(vmax, slab_floats(h5_, w5_, std::max(m->widths[4], m->widths[5]), 7));
            vmax = std::max(vmax, (size_t)al * al * (size_t)wino_slab(wino_tiles(4, N, h5_, w5_), std::max(m->widths[4], m->widths[5])));
        }
        m->d_wino_v = m->d_wino_m = nullptr;
        if (vmax) { items.push_back({"wino_v", vmax, 0, 0, 0, &m->d_wino_v}); items.push_back({"wino_m", vmax, 0, 0, 0, &m->d_wino_m}); }
        if (m->wino_min_cin > 0) {    // the forward pass keeps each Winograd layer's transformed input for the weight gradient
            int cin = 3;
            for (int b = 0, hh = H, ww = W; b < 5; ++b, hh /= 2, ww /= 2)
                for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
                    if (cin >= m->wino_min_cin && cin % 16 == 0 && m->widths[b] % 64 == 0 && wino_tile_for(m, hh, ww, 3)) {
                        char nm[40]; snprintf(nm, sizeof nm, "wv:conv%d_%d", b + 1, i);
                        items.push_back({nm, slab_floats(hh, ww, cin, 3), 0, 0, 0, nullptr});
                    }
                    cin = m->widths[b];
                }
            if (fc6w) items.push_back({"wv:fc6", slab_floats(h5_, w5_, m->widths[4], 7), 0, 0, 0, nullptr});
            if (m->defer_wgrad > 0) {
                // deferred weight gradients: dM = A dY A^T of conv3_1 .. conv5_3 (and fc6, and fc7's dz) stays alive until the side stream has used it
                int cin2 = m->widths[1];
                for (int b = 2, hh = H / 4, ww = W / 4; b < 5; ++b, hh /= 2, ww /= 2)
                    for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
                        const int cw = m->widths[b];
                        if (cin2 >= m->wino_min_cin && cin2 % 64 == 0 && cw % 64 == 0 && wino_tile_for(m, hh, ww, 3) == 6) {
                            char nm[40]; snprintf(nm, sizeof nm, "dmk:conv%d_%d", b + 1, i);
                            items.push_back({nm, (size_t)64 * (size_t)wino_slab(wino_tiles(6, N, hh, ww), cw), 0, 0, 0, nullptr});
                        }
                        cin2 = cw;
                    }
                if (m->defer_wgrad > 1) {
                    if (fc6w && m->widths[4] % 2 == 0 && bt_gemm_ok(m->widths[5], 4 * m->widths[4])) {
                        const int al = wino_alpha(4, 7);
                        items.push_back({"dmk:fc6", (size_t)al * al * (size_t)wino_slab(wino_tiles(4, N, h5_, w5_), m->widths[5]), 0, 0, 0, nullptr});
                    }
                    items.push_back({"dz:fc7", (size_t)N * h5_ * w5_ * m->widths[6], 0, 0, 0, nullptr});
                }
            }
        }
    }

    size_t bytes = 0;
    std::vector<size_t> offs;
    for (auto& it : items) { offs.push_back(bytes); bytes = align_up(bytes + it.n * sizeof(float), 256); }
    const size_t npix = (size_t)N * H * W;
    const size_t o_img = bytes;  bytes = align_up(bytes + npix * 3 * sizeof(float), 256);
    const size_t o_lab = bytes;  bytes = align_up(bytes + npix, 256);
    const size_t o_pred = bytes; bytes = align_up(bytes + npix * sizeof(long long), 256);
    const size_t o_part = bytes; bytes = align_up(bytes + 4096 * sizeof(double), 256);
    hipError_t e = hipMalloc((void**)&m->arena, bytes);
    if (e != hipSuccess) { m->arena = nullptr; return fail(m, FCN8S_ERR_OOM, std::string("workspace hipMalloc failed: ") + hipGetErrorString(e)); }
    m->arena_bytes = bytes;
    for (size_t i = 0; i < items.size(); ++i) {
        float* p = (float*)(m->arena + offs[i]);
        if (items[i].extra) *items[i].extra = p;
        Act a; a.p = p; a.n = items[i].n; a.H = items[i].h; a.W = items[i].w; a.C = items[i].c;
        m->acts[items[i].name] = a;
    }
    m->d_images = m->arena + o_img; m->d_labels = (uint8_t*)(m->arena + o_lab);
    m->d_pred = (long long*)(m->arena + o_pred); m->d_partials = (double*)(m->arena + o_part);
    m->N = N; m->H = H; m->W = W;
    return FCN8S_OK;
}

float* A(fcn8s_model* m, const char* n) { return m->acts.at(n).p; }

// ---- the last transposed conv (16x16, stride 8) as one GEMM over output blocks -----------------------------------------------
// (layout and algebra: PixMap in fcn8s_internal.h, tconv_*_kernel in elementwise.hip).  The s*s sub-pixel phases of tconv_fwd give every
// phase a 20-column GEMM on a 32-wide MFMA tile through the generic predicated kernel (37 TFLOP/s); as ONE GEMM with rows = output
// blocks, K = 4C and 64C columns it runs on the LDS-DMA kernels, and so do its two gradients.
const float* LG(fcn8s_model* m) { return (m->tconv_gemm && m->logits_b) ? m->logits_b : m->acts.at("logits").p; }
const PixMap* LGM(fcn8s_model* m) { return (m->tconv_gemm && m->logits_b) ? &m->pm : nullptr; }
IgemmArgs tg_rows_gemm(const float* x, int ldx, int K, const float* w, float* y, int ncols, long long rows)
{
    IgemmArgs a{};
    a.x = x; a.w = w; a.y = y;
    a.N = 1; a.Ma = (int)rows; a.Mb = 1; a.M = rows;
    a.Hi = (int)rows; a.Wi = 1; a.Cin = K; a.ldx = ldx;
    a.KW = 1; a.in_scale = 1; a.tap_step = 1; a.tap_off = 0; a.Ktot = K;
    a.Ho = (int)rows; a.Wo = 1; a.Cout = ncols; a.ldy = ncols;
    a.out_scale = 1; a.phases_x = 1; a.alpha = 1.f; a.mask_scale = 1.f;
    return a;
}
void tconv_gemm_fwd(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    const int C = m->C, NC = 64 * C, KP = m->tg_kp;
    const long long rows = (long long)m->N * m->pm.QH * m->pm.QW;
    ProfScope ps(m, "tconv_fwd", 2.0 * rows * 4 * C * NC, 4.0 * rows * (4.0 * C + NC));
    launch_tconv_pack_gemm(Wp(m, "fc7_pool4_pool3_conv2d_trans/kernel"), Wp(m, "fc7_pool4_pool3_conv2d_trans/bias"), m->tg_b2, m->tg_b2t, m->tg_bias, C, 8, KP, s);
    launch_tconv_im2col(m->acts.at("a3").p, m->tg_A, m->N, m->H / 8, m->W / 8, C, KP, s);
    IgemmArgs a = tg_rows_gemm(m->tg_A, KP, 4 * C, m->tg_b2, m->logits_b, NC, rows);
    a.split = split_of(m);
    a.bias = m->tg_bias;
    launch_igemm(a, 1, s);
}
void tconv_gemm_dgrad(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    const int C = m->C, NC = 64 * C, KP = m->tg_kp;
    const long long rows = (long long)m->N * m->pm.QH * m->pm.QW;
    ProfScope ps(m, "tconv_dgrad", 2.0 * rows * NC * KP, 4.0 * rows * (NC + KP));
    IgemmArgs a = tg_rows_gemm(m->dlogits_b, NC, NC, m->tg_b2t, m->tg_dA, KP, rows);
    a.split = split_of(m);
    a.batched = 1;                                   // one slab: takes the plain (16-byte store) epilogue
    launch_igemm(a, 1, s);
    launch_tconv_col2im(m->tg_dA, m->da3, m->N, m->H / 8, m->W / 8, C, KP, s);
}
void tconv_gemm_wgrad(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    const int C = m->C, NC = 64 * C, KP = m->tg_kp;
    const long long rows = (long long)m->N * m->pm.QH * m->pm.QW;
    ProfScope ps(m, "tconv_wgrad", 2.0 * rows * KP * NC, 4.0 * rows * (NC + KP));
    WgradArgs g{}; g.split = split_of(m);
    g.A = m->tg_A; g.B = m->dlogits_b; g.C = m->tg_db2;
    g.N = 1; g.Pa = 1; g.Pb = (int)rows; g.P = rows;
    g.Ha = 1; g.Wa = (int)rows; g.Adim = KP; g.lda = KP; g.Areal = KP;
    g.Bdim = NC; g.ldb = NC; g.KW = 1; g.a_scale = 1; g.tap_off = 0; g.ntaps = 1; g.ldc = NC; g.alpha = 1.f; g.colsum = nullptr;
    g.c_uninitialized = 1;
    launch_wgrad(g, s);
    launch_tconv_unpack_dw(m->tg_db2, m->d_grads + m->params[m->index.at("fc7_pool4_pool3_conv2d_trans/kernel")].offset, C, 8, s);
}

// ---- forward ------------------------------------------------------------------------
int stage_inputs(fcn8s_model* m, const void* images, int dtype, const uint8_t* labels, int where,
                 const void** img_dev, const uint8_t** lab_dev)
{
    const size_t npix = (size_t)m->N * m->H * m->W;
    if (where == FCN8S_HOST) {
        const size_t ib = npix * 3 * (dtype == FCN8S_IMG_U8 ? 1 : 4);
        HIPCHK(m, hipMemcpyAsync(m->d_images, images, ib, hipMemcpyHostToDevice, m->stream));
        *img_dev = m->d_images;
        if (labels) {
            HIPCHK(m, hipMemcpyAsync(m->d_labels, labels, npix, hipMemcpyHostToDevice, m->stream));
            *lab_dev = m->d_labels;
        } else *lab_dev = nullptr;
    } else { *img_dev = images; *lab_dev = labels; }
    return FCN8S_OK;
}

void prepare_forward_weights(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    launch_pad_cin(Wp(m, "conv1_1/filter"), m->d_w1pad, 9, 3, 4, m->widths[0], s);
    launch_tconv_phase_pack(Wp(m, "fc7_conv2d_trans/kernel"), m->d_tph[0], 4, 2, m->C, s);
    launch_tconv_phase_pack(Wp(m, "fc7_pool4_conv2d_trans/kernel"), m->d_tph[1], 4, 2, m->C, s);
    launch_tconv_phase_pack(Wp(m, "fc7_pool4_pool3_conv2d_trans/kernel"), m->d_tph[2], 16, 8, m->C, s);
}

// Whether the backward pass routes d(pool_b) through the argmax bytes inside the Winograd transform of conv_b_last (then neither dZ
// nor the conv's full-resolution output is ever read again); b is the 1-based block number.  Same test as backward_blocks().
bool pool_backward_fused(const fcn8s_model* m, int b, bool pooled_by_transform)
{
    const int h = m->H >> (b - 1), w = m->W >> (b - 1), cw = m->widths[b - 1], nconv = kConvsPerBlock[b - 1];
    char last[32]; snprintf(last, sizeof last, "wv:conv%d_%d", b, nconv);
    return pooled_by_transform && m->wino_min_cin > 0 && cw >= m->wino_min_cin && m->d_wino_v &&
           wino_tile_for(m, h, w) >= 4 && cw % 64 == 0 && nconv > 1 && m->acts.count(last);
}

// One SAME convolution with bf16-rounded operands and fp32 accumulation on the bf16 MFMA (gemm_bf16.hip), fp32 bias / ReLU / dropout epilogue,
// fp32 output: fc6 / fc7 of FCN8S_PREC_BF16_FC, and conv3_1 .. conv5_3 as well in FCN8S_PREC_BF16_FWD.  Returns false if no bf16 kernel
// takes the shape (the caller then uses the fp32 path).
bool bf16_conv_layer(fcn8s_model* m, const char* tag, const char* wname, const char* bname, const float* in, float* out,
                     int N, int h, int w, int cin, int cout, int k, int drop, float keep_prob, uint32_t stream_id, hipStream_t s, bool allow_small = true,
                     const unsigned short* xb_ready = nullptr,          // the padded bf16 copy of `in`, already made (256 x 256 kernel only)
                     bool any_shape = false,                            // bf16_train: 64- / 128-column tiles and a partial last row tile are taken too
                     unsigned short* yb = nullptr, int yb_pad = 0,      // ... and the consumer's padded bf16 copy of the output is written by the epilogue
                     long long xb_ps = 0, long long yb_ps = 0)          // plane strides of xb_ready / yb (bf16_train's per-layer copies: channel-chunk planes), 0 = [rows][C]
{
    const int K = k * k * cin;
    const long long Mrows = (long long)N * h * w;
    const bool big = conv_bf16_256_ok(Mrows, cin, cout, any_shape ? 3 : m->bf16_gemm256);      // 256-row tiles, LDS-DMA, staggered wave groups
    if (!big && (!allow_small || cin % 32 || cout % 128)) return false;
    const size_t wneed = (size_t)K * cout;
    if (m->wbf16_elems < wneed) {
        if (m->d_wbf16) { hipStreamSynchronize(s); hipFree(m->d_wbf16); m->d_wbf16 = nullptr; m->wbf16_elems = 0; }
        if (hipMalloc((void**)&m->d_wbf16, wneed * sizeof(unsigned short)) != hipSuccess) { (void)hipGetLastError(); return false; }
        m->wbf16_elems = wneed;
    }
    // weights: bf16, K-tile-major blocks for the 128 x 128 kernel or transposed [Cout][K] for the 256 x 256 one; with frozen parameters
    // (evaluation / serving loops) each layer's copy is made once and kept
    unsigned short* wbuf = m->d_wbf16;
    bool have = false;
    if (m->frozen) {
        unsigned short*& c = m->wbf16_cache[std::string(wname) + (big ? "#t" : "#b")];
        if (c) { wbuf = c; have = true; }
        else if (hipMalloc((void**)&c, wneed * sizeof(unsigned short)) == hipSuccess) wbuf = c;
        else { c = nullptr; (void)hipGetLastError(); }
    }
    if (!have) { ProfScope ps(m, "weight_relayout", 0, 6.0 * K * cout);
                 if (big) launch_w_to_bf16_t(Wp(m, wname), wbuf, K, cout, s); else launch_w_to_bf16_tiles(Wp(m, wname), wbuf, K, cout, s); }
    const int pad = big ? (k - 1) / 2 : 0;
    const size_t nin = (size_t)N * (h + 2 * pad) * (w + 2 * pad) * cin;
    if (nin % 8 == 0 && m->abf16_elems < nin && !(big && xb_ready)) {
        if (m->d_abf16) { hipStreamSynchronize(s); hipFree(m->d_abf16); m->d_abf16 = nullptr; m->abf16_elems = 0; }
        if (hipMalloc((void**)&m->d_abf16, nin * sizeof(unsigned short)) == hipSuccess) m->abf16_elems = nin; else (void)hipGetLastError();
    }
    const double M = (double)Mrows;
    if (big && (xb_ready || (m->d_abf16 && m->abf16_elems >= nin))) {
        if (!xb_ready) { ProfScope ps(m, "weight_relayout", 0, 4.0 * Mrows * cin + 2.0 * nin); launch_f32_to_bf16_padded(in, m->d_abf16, N, h, w, cin, pad, s); }
        Bf16Conv256Args g{};
        g.xp = xb_ready ? xb_ready : m->d_abf16; g.wt = wbuf; g.bias = Wp(m, bname); g.y = out;
        g.N = N; g.H = h; g.W = w; g.Cin = cin; g.Cout = cout; g.K = k;
        g.relu = 1; g.dropout = drop; g.keep_prob = keep_prob; g.seed = m->seed; g.stream_id = stream_id; g.any_shape = any_shape ? 1 : 0; g.mask_scale = 1.f; g.yb = yb; g.yb_pad = yb_pad;
        g.xp_ps = xb_ready ? xb_ps : 0; g.yb_ps = yb_ps;
        g.rows_bn = m->bf16_rows_bn;
        g.guarded = (any_shape && xb_ready) ? 1 : 0;          // (the per-layer training copies carry guard rows; the shared inference copy does not)
        const std::string lname = std::string(wname).substr(0, std::string(wname).find('/'));
        ProfScope ps(m, tag, 2.0 * M * K * cout, (out ? 4.0 : 0.0) * M * cout + (yb ? 2.0 : 0.0) * M * cout + 2.0 * M * cin + 2.0 * K * cout, any_shape ? lname.c_str() : nullptr);
        if (launch_conv_bf16_256(g, s)) return true;
    }
    if (!out) return false;
    if (!allow_small || cin % 32 || cout % 128) return false;
    if (big) { launch_w_to_bf16_tiles(Wp(m, wname), m->d_wbf16, K, cout, s); wbuf = m->d_wbf16; }      // (could not take the 256 path after all)
    Bf16ConvArgs a{};
    a.x = in; a.wt = wbuf; a.bias = Wp(m, bname); a.y = out;
    if (nin % 8 == 0 && m->d_abf16) {         // activations to bf16 once: the GEMM re-reads each A tile Cout/128 times
        ProfScope ps(m, "weight_relayout", 0, 6.0 * nin); launch_f32_to_bf16(in, m->d_abf16, (long long)N * h * w * cin, s); a.xh = m->d_abf16;
    }
    a.N = N; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout; a.K = k;
    a.relu = 1; a.dropout = drop; a.keep_prob = keep_prob; a.seed = m->seed; a.stream_id = stream_id;
    ProfScope ps(m, tag, 2.0 * M * K * cout, 4.0 * M * (cin + cout) + 2.0 * K * cout);
    return launch_conv_bf16(a, s);
}

// bf16_train: a guarded, zero-bordered bf16 buffer for layer `layer` out of `bufs` (allocated and zeroed on first use or when the shape grows: the
// border and the guard rows are never written again, the interior is overwritten in every pass); returns the address of padded pixel 0 or nullptr
unsigned short* g16_for(fcn8s_model* m, std::map<std::string, unsigned short*>& bufs, std::map<std::string, size_t>& sizes, const char* layer,
                        int N, int H, int W, int C, int K, hipStream_t s)
{
    const int pad = (K - 1) / 2, Wp_ = W + 2 * pad;
    const long long G = bf16_guard_rows(K, Wp_), R = (long long)N * (H + 2 * pad) * Wp_;
    const size_t need = (size_t)(R + 2 * G) * C;
    unsigned short*& p = bufs[layer];
    size_t& have = sizes[layer];
    if (!p || have != need) {           // (another shape: another border)
        if (p) { hipStreamSynchronize(s); hipFree(p); p = nullptr; have = 0; }
        if (hipMalloc((void**)&p, need * sizeof(unsigned short)) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return nullptr; }
        have = need;
        hipMemsetAsync(p, 0, need * sizeof(unsigned short), s);
    }
    return p + g16_off(G, C);           // padded pixel 0 of plane 0 (planes are g16_ps() apart)
}
// the copy of layer `layer`'s INPUT (forward pass; read again by its weight gradient)
unsigned short* xg16_for(fcn8s_model* m, const char* layer, int N, int H, int W, int C, int K, hipStream_t s) { return g16_for(m, m->xg16, m->xg16_elems, layer, N, H, W, C, K, s); }
// the copy of layer `layer`'s output gradient dY [N][H][W][C]: written by the kernel that produced dY if that was a bf16 data gradient
// (dyg16_filled), else converted here, once (the layer's weight gradient asks first, its data gradient finds it)
// db: the layer's bias gradient, db[c] += sum dY[., c] -- taken by the conversion kernel on its way through dY (*db_done says whether it was)
unsigned short* dyb_for(fcn8s_model* m, const char* layer, const float* dy, int N, int H, int W, int C, int K, hipStream_t s, float* db, bool* db_done)
{
    if (db_done) *db_done = false;
    unsigned short* p = g16_for(m, m->dyg16, m->dyg16_elems, layer, N, H, W, C, K, s);
    if (!p) return nullptr;
    if (!m->dyg16_filled.count(layer)) {
        ProfScope ps(m, "bf16_convert", 0, 4.0 * N * H * W * C + 2.0 * N * H * W * C);
        if (db && launch_f32_to_bf16_padded_colsum(dy, p, db, N, H, W, C, (K - 1) / 2, s, g16_ps(N, H, W, K))) { if (db_done) *db_done = true; }
        else launch_f32_to_bf16_padded(dy, p, N, H, W, C, (K - 1) / 2, s, g16_ps(N, H, W, K));
        m->dyg16_filled.insert(layer);
    }
    return p;
}

// bf16 modes, training: the layer's Winograd input transform (run for the weight gradient anyway) can write the padded bf16 copy its direct
// bf16 convolution reads -- if that convolution takes the 256 x 256 kernel and the transform is the F(6x6,3x3) one.  Returns the layer's copy
// (allocated and its border zeroed on first use) or nullptr.
unsigned short* xb_by_transform(fcn8s_model* m, const char* layer, int N, int H, int W, int Cin, int Cout, hipStream_t s)
{
    if (!m->bf16_copy_by_transform || Cin % 8 || !m->acts.count(std::string("wv:") + layer) || wino_tile_for(m, H, W, 3) != 6) return nullptr;
    if (!conv_bf16_256_ok((long long)N * H * W, Cin, Cout, m->bf16_gemm256)) return nullptr;
    unsigned short*& p = m->xbf16[layer];
    if (!p) {
        const size_t bytes = (size_t)N * (H + 2) * (W + 2) * Cin * sizeof(unsigned short);
        if (hipMalloc((void**)&p, bytes) != hipSuccess) { p = nullptr; (void)hipGetLastError(); m->xbf16.erase(layer); return nullptr; }
        hipMemsetAsync(p, 0, bytes, s);
    }
    return p;
}

// What the backward pass of a Winograd layer expects from the forward pass when the forward convolution itself did not run through
// Winograd (the bf16 modes): the transformed input V (kept for the weight gradient in the Winograd domain) and, for the adjoint data
// gradient, the forward filter bank of THIS step's weights (a bank left over from an earlier step would be silently wrong).
// in_rbits_out / in_layer: the input is the ReLU output of conv `in_layer`, whose (x > 0) record the transform writes on the way (wino_input_kernel)
// xb: also fill the interior of this padded bf16 copy of x (F(6x6,3x3) layers; the caller has checked xb_by_transform_ok)
void wino_backward_operands(fcn8s_model* m, const char* layer, const float* x, const float* wk, int N, int H, int W, int Cin, int Cout, int KS, hipStream_t s,
                            unsigned* in_rbits_out = nullptr, const char* in_layer = nullptr, unsigned short* xb = nullptr)
{
    const int tile = wino_tile_for(m, H, W, KS);
    if (!tile) return;
    auto it = m->acts.find(std::string("wv:") + layer);
    if (it == m->acts.end()) return;
    const int P = wino_alpha(tile, KS) * wino_alpha(tile, KS), nsub2 = wino_nsub(KS) * wino_nsub(KS), Kg = nsub2 * Cin;
    const long long T = wino_tiles(tile, N, H, W);
    { ProfScope ps(m, "wino_transform", 0, 4.0 * ((double)N * H * W * Cin * nsub2 + (double)P * T * Kg) + (xb ? 2.0 * N * H * W * Cin : 0.0));
      if (xb) launch_wino_input_xb(x, it->second.p, xb, N, H, W, Cin, s, in_rbits_out);
      else launch_wino_input(tile, x, it->second.p, N, H, W, Cin, KS, s, KS == 3 ? in_rbits_out : nullptr);
      if (in_rbits_out && in_layer && KS == 3) m->rbits_ok.insert(in_layer); }
    const std::string key = std::string(layer) + "#" + std::to_string(tile);
    if ((KS == 3 && tile == 6) || (KS == 7 && tile == 4)) {
        float*& tu = m->u_train[key];
        if (!tu && hipMalloc((void**)&tu, (size_t)P * Kg * Cout * sizeof(float)) != hipSuccess) { tu = nullptr; (void)hipGetLastError(); }
        if (tu) { ProfScope ps(m, "wino_transform", 0, (double)(KS * KS + P * nsub2) * 4 * Cin * Cout); launch_wino_filter(tile, wk, tu, Cin, Cout, KS, s); }
        else m->u_train.erase(key);
    }
}

int forward(fcn8s_model* m, const void* img_dev, int dtype, float keep_prob, bool train)
{
    t_deterministic = m->deterministic;
    auto drop_banks = [&]() {
        hipStreamSynchronize(m->stream);
        for (auto& kv : m->u_cache) if (kv.second) hipFree(kv.second);
        m->u_cache.clear();
        for (auto& kv : m->wbf16_cache) if (kv.second) hipFree(kv.second);
        m->wbf16_cache.clear();
    };
    hipStream_t s = m->stream;
    const int N = m->N, H = m->H, W = m->W, C = m->C;
    bool guard_pending = false;
    if (m->frozen && !m->u_cache.empty()) {
        // the caller promised constant parameters; a cheap strided fingerprint catches the promise being broken through a side
        // door (a torch optimizer or copy_ over views of ext_params): the cached filter banks are then rebuilt instead of reused.
        // The check does not hold the pass up: the fingerprint is taken first on the stream, the pass is enqueued behind it with the kept
        // banks, and the host compares when the (long finished) copy is looked at -- at the end of this function; a mismatch repeats the pass.
        launch_fingerprint(m->d_params, (long long)m->total, m->d_fp, s);
        if (!m->h_fp && hipHostMalloc((void**)&m->h_fp, sizeof(unsigned long long)) != hipSuccess) { m->h_fp = nullptr; (void)hipGetLastError(); }
        if (m->h_fp && !m->fp_event) hipEventCreateWithFlags(&m->fp_event, hipEventDisableTiming);
        if (m->h_fp && m->fp_event) {
            hipMemcpyAsync(m->h_fp, m->d_fp, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
            hipEventRecord(m->fp_event, s);
            guard_pending = true;
        } else {
            unsigned long long fp = 0;
            hipMemcpyAsync(&fp, m->d_fp, sizeof fp, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            if (fp != m->frozen_fp) drop_banks();
        }
    }
    const bool fill_fp = m->frozen && m->u_cache.empty();
    if (!m->frozen || fill_fp) prepare_forward_weights(m);        // frozen and the kept banks still valid: so are the padded / phase-packed kernels
    m->fwd_train = train;
    // bf16_train, evaluation / prediction (round 6): the pass takes the TRAINING pass's data flow -- every layer's input as a padded bf16 copy written by its
    // producer's epilogue (no fp32 conv -> conv tensor, no conversion pass), the flat-position kernel, pools on the bf16 copies -- instead of fp32 tensors converted
    // layer by layer for the tile kernel: 13.3 -> 9.3 ms per 16 x 1024x512 batch, 1.55 -> 1.18 ms per single image (profiles/r06_bf16_infer.txt).  Same products in the same order as the training pass:
    // its logits are the training pass's bit for bit (keep_prob 1).  What it keeps for a backward pass that never comes (routing bytes) costs one byte per window.
    const bool cp = train || (bf16_train_mode(m) && m->bf16_infer_copies);
    m->rbits_ok.clear(); m->y_unwritten.clear(); m->in_bf16_only.clear(); m->fwd_v_layer.clear(); m->xg16_filled.clear();
    { ProfScope ps(m, "preprocess", 0, (double)N * H * W * (16 + (dtype ? 12 : 3))); launch_preprocess(img_dev, dtype, A(m, "x0"), (long long)N * H * W, s); }
    const float* x = A(m, "x0");
    int h = H, w = W, cin = 4;
    for (int b = 0; b < 5; ++b) {
        char pn[32]; bool pooled = false;
        for (int i = 1; i <= kConvsPerBlock[b]; ++i) {
            char nm[32]; snprintf(nm, sizeof nm, "conv%d_%d", b + 1, i);
            const bool first = (b == 0 && i == 1);
            Epi e; e.bias = Wp(m, std::string(nm) + "/biases"); e.relu = 1;
            const float* wt = first ? m->d_w1pad : Wp(m, std::string(nm) + "/filter");
            if (train && i < kConvsPerBlock[b]) {                                                            // its output is the next conv's input
                auto it = m->acts.find(std::string("rb:") + nm);
                if (it != m->acts.end()) e.relu_bits_out = (unsigned*)it->second.p;
            }
            if (train && b == 0 && i == 2) {                                                                 // input = conv1_1, made by the gather kernel
                auto it = m->acts.find("rb:conv1_1");
                if (it != m->acts.end()) { e.in_relu_bits_out = (unsigned*)it->second.p; e.in_layer = "conv1_1"; }
            }
            if (i == kConvsPerBlock[b]) {                                                                    // last conv of the block
                snprintf(pn, sizeof pn, "pool%d", b + 1); e.pool_out = A(m, pn);
                if (train) { char ix[16]; snprintf(ix, sizeof ix, "pidx%d", b + 1); e.pool_idx = (unsigned char*)A(m, ix); }
                // The block's last conv output feeds only the pool.  If the output transform writes the pool (and, for training, the backward
                // pass routes through the argmax bytes), the full-resolution tensor is never read again and is not written at all
                // (2.15 GB for conv1_2 at 16 x 1024x512); fcn8s_get_activation of such a layer then returns stale data.
                e.skip_y = !train || pool_backward_fused(m, b + 1, true);
            }
            char nxt[32] = "";
            if (m->fuse_out_in && !first && i < kConvsPerBlock[b] && !(bf16_fwd_mode(m) && b >= 2) && !bf16_train_mode(m) && (!train || e.relu_bits_out) &&
                m->wino_min_cin > 0 && cin >= m->wino_min_cin && m->widths[b] >= m->wino_min_cin && m->widths[b] % 64 == 0 && cin % 16 == 0 &&
                m->d_wino_v && wino_tile_for(m, h, w, 3) == 6) {
                // this conv and the next one both run through F(6x6,3x3) on the same tile grid: its output transform writes the next conv's
                // transformed input directly (training: into the buffer kept for that conv's weight gradient) and its own output never exists
                snprintf(nxt, sizeof nxt, "conv%d_%d", b + 1, i + 1);
                // the buffer the next conv will read its V from -- conv_same's own rule: the one kept for its weight gradient if the workspace
                // has one (whether or not this pass trains), else the shared scratch.  Training: only if that kept buffer exists -- a direct
                // weight gradient would read the activation that no longer exists.
                auto it = m->acts.find(std::string("wv:") + nxt);
                if (!train || it != m->acts.end()) { e.next_v = it != m->acts.end() ? it->second.p : m->d_wino_v; e.next_layer = nxt; }
            }
            bool done = false;
            if (first && !bf16_train_mode(m) && m->conv1_in_transform && m->widths[0] == 64 && kConvsPerBlock[0] == 2 && m->widths[0] >= m->wino_min_cin && m->wino_min_cin > 0 &&
                m->d_wino_v && wino_tile_for(m, h, w, 3) == 6) {
                // conv1_1's only reader is conv1_2's F(6x6,3x3) input transform: that transform evaluates conv1_1 on its own patches, straight from the
                // image (winograd.hip: wino_input_conv1_kernel), writes conv1_2's V -- into the buffer conv_same will look for it in -- and, in training,
                // the ReLU record the backward pass masks with.  conv1_1's 134 MB per image are never written or read.
                auto wv = m->acts.find("wv:conv1_2");
                auto rb = m->acts.find("rb:conv1_1");
                if (!train || (wv != m->acts.end() && rb != m->acts.end())) {
                    float* vdst = wv != m->acts.end() ? wv->second.p : m->d_wino_v;
                    const long long T = wino_tiles(6, N, h, w);
                    ProfScope ps(m, "wino_transform", 0, 4.0 * (4.0 * N * h * w + 64.0 * T * 64.0) + (train ? 8.0 * N * h * w : 0.0), nm);
                    launch_wino_input_conv1(x, m->d_w1pad, e.bias, vdst, N, h, w, s, train ? (unsigned*)rb->second.p : nullptr);
                    if (train) m->rbits_ok.insert(nm);
                    m->fwd_v_layer = "conv1_2"; m->y_unwritten.insert(nm);
                    done = true;
                }
            }
            if (!done && first && m->widths[0] == 64) {            // conv1_1: write-bound gather kernel (igemm.hip: conv1_glds_kernel)
                // bf16_train, training: conv1_2 reads this layer as its padded bf16 copy and nobody else reads it (the mask of conv1_2's data gradient is the
                // sign of that copy): the tile kernel writes the copy and no fp32 tensor
                unsigned short* y16 = nullptr;
                if (bf16_train_mode(m) && cp && m->bf16_acts && m->conv1_tiled && h % 8 == 0 && w % 16 == 0 && kConvsPerBlock[0] >= 2 && m->widths[0] % 64 == 0)
                    y16 = xg16_for(m, "conv1_2", N, h, w, m->widths[0], 3, s);
                ProfScope ps(m, "conv1_1_fwd", 2.0 * N * h * w * 27.0 * m->widths[0], 4.0 * N * h * w * 3.0 + (y16 ? 2.0 : 4.0) * N * h * w * m->widths[0], nm);
                done = launch_conv1_fwd(x, m->d_w1pad, e.bias, y16 ? nullptr : A(m, nm), m->d_w1pad + 12 * 4 * (size_t)m->widths[0], N, h, w, m->widths[0], m->conv1_tiled, s, y16, g16_ps(N, h, w, 3));
                if (done && y16) { m->xg16_filled.insert("conv1_2"); m->y_unwritten.insert(nm); m->in_bf16_only.insert("conv1_2"); }
            }
            if (!done && bf16_train_mode(m) && !first) {
                // FCN8S_PREC_BF16_TRAIN: every convolution but conv1_1 (3 input channels) as a direct convolution with bf16-rounded operands; the
                // training pass keeps the layer's padded bf16 input copy for its weight gradient (the convolution starts from that copy)
                unsigned short* xb = cp ? xg16_for(m, nm, N, h, w, cin, 3, s) : nullptr;
                if (xb && !m->xg16_filled.count(nm)) {
                    ProfScope ps(m, "bf16_convert", 0, 4.0 * N * h * w * cin + 2.0 * N * (h + 2) * (w + 2) * cin); launch_f32_to_bf16_padded(x, xb, N, h, w, cin, 1, s, g16_ps(N, h, w, 3));
                }
                // the next convolution of the block reads this output as ITS padded bf16 input: this kernel's epilogue writes that copy
                unsigned short* yb = nullptr; char nx[32] = "";
                if (cp && (m->bf16_fuse_convert || m->bf16_acts) && i < kConvsPerBlock[b]) { snprintf(nx, sizeof nx, "conv%d_%d", b + 1, i + 1); yb = xg16_for(m, nx, N, h, w, m->widths[b], 3, s); }
                // the block's LAST convolution is read by its pool only, and pool1 / pool2 / pool5 only by bf16 convolutions: the pool then takes this output
                // as a bf16 copy of the kernel's own geometry ("pool<b>in"), picks its maxima among the bf16 values (what bf16(max of the fp32 values) is anyway)
                // and no fp32 tensor is written (pool3 / pool4 also feed the fp32 skip heads: their blocks keep the fp32 tensor)
                const bool pool16 = cp && m->bf16_acts && m->bf16_fuse_pool && i == kConvsPerBlock[b] && b != 2 && b != 3 && cin % 64 == 0 && m->widths[b] % 64 == 0 &&
                                    m->widths[b == 4 ? 5 : b + 1] % 64 == 0;
                if (pool16) { snprintf(nx, sizeof nx, "pool%din", b + 1); yb = xg16_for(m, nx, N, h, w, m->widths[b], 3, s); }
                // ... and if that is the output's only reader (option bf16_acts; the mask of the consumer's data gradient is the sign of the copy), the fp32
                // tensor is not written at all.  (Both layers' gradients must fit the bf16 kernels: a fallback would look for the fp32 tensor.)
                const bool only16 = yb && m->bf16_acts && cin % 64 == 0 && m->widths[b] % 64 == 0;
                // (the convolution kernels address their padded copies through a 64-bit tile base; the weight-gradient kernels' 32-bit per-lane byte offsets span TWO
                //  32-channel planes of a copy -- launch_wgrad_bf16's limit, applied HERE, before anything is launched: a 32-channel plane of the padded map must stay
                //  below 4 GiB, i.e. about 67 million padded positions (127 images of 1024x512).  Round 5 refused a whole copy of 4 GiB: 64 x 1024x512.)
                if ((double)g16_ps(N, h, w, 3) * 2.0 + 65536.0 >= 4294967296.0)
                    return fail(m, FCN8S_ERR_SHAPE, std::string("bf16_train: a 32-channel plane of the padded bf16 copy of ") + nm + "'s input would reach 4 GiB at this batch size; use a smaller batch per GPU");
                done = bf16_conv_layer(m, "conv3x3_fwd_bf16", (std::string(nm) + "/filter").c_str(), (std::string(nm) + "/biases").c_str(), x, only16 ? nullptr : A(m, nm),
                                       N, h, w, cin, m->widths[b], 3, 0, 1.f, 0, s, /*allow_small=*/false, xb, /*any_shape=*/true, yb, 1, g16_ps(N, h, w, 3), g16_ps(N, h, w, 3));
                if (!done) return fail(m, FCN8S_ERR_SHAPE, std::string("bf16_train: ") + nm + " does not fit the bf16 convolution kernel");
                if (done && yb) m->xg16_filled.insert(nx);
                if (done && only16) { m->y_unwritten.insert(nm); if (!pool16) m->in_bf16_only.insert(nx); }
            }
            if (!done && bf16_fwd_mode(m) && b >= 2) {
                // FCN8S_PREC_BF16_FWD: conv3_1 .. conv5_3 as direct convolutions with bf16-rounded operands on the 256 x 256 bf16 kernel (the
                // output is materialised, the block's pool runs as its own kernel, ReLU masks come from the activations); the backward pass
                // stays in the Winograd domain, so the transformed input and this step's filter bank are made here
                // training: the transform that keeps this layer's V for the weight gradient runs first and writes the padded bf16 copy of the
                // input on the way (the convolution then starts from it; without a kept V, or in inference, the convolution converts its input itself)
                unsigned short* xb = train ? xb_by_transform(m, nm, N, h, w, cin, m->widths[b], s) : nullptr;
                auto operands = [&]() {
                    // the previous conv of the block came from the bf16 kernel too (no ReLU record): this transform of its output writes one
                    unsigned* irb = nullptr; char prev[32] = "";
                    if (i > 1) {
                        snprintf(prev, sizeof prev, "conv%d_%d", b + 1, i - 1);
                        auto it = m->acts.find(std::string("rb:") + prev);
                        if (it != m->acts.end() && !m->rbits_ok.count(prev)) irb = (unsigned*)it->second.p;
                    }
                    wino_backward_operands(m, nm, x, wt, N, h, w, cin, m->widths[b], 3, s, irb, irb ? prev : nullptr, xb);
                };
                if (xb) operands();
                done = bf16_conv_layer(m, "conv3x3_fwd_bf16", (std::string(nm) + "/filter").c_str(), (std::string(nm) + "/biases").c_str(), x, A(m, nm),
                                       N, h, w, cin, m->widths[b], 3, 0, 1.f, 0, s, /*allow_small=*/false, xb);
                if (done && train && !xb) operands();
            }
            if (!done) pooled = conv_same(m, first ? "conv1_1_fwd" : "conv3x3_fwd", x, wt, A(m, nm), N, h, w, cin, m->widths[b], 3, e, s, first ? 3 : 0, nm);
            x = A(m, nm); cin = m->widths[b];
        }
        snprintf(pn, sizeof pn, "pool%d", b + 1);
        m->pool_fused[b] = pooled && train;
        m->pool_routed[b] = false;
        if (!pooled && bf16_train_mode(m) && cp && m->bf16_fuse_pool && cin % 4 == 0) {
            // bf16_train, training: the pool keeps its routing bytes (the backward pass reads one byte per window instead of the block's last activation)
            // and writes the consumer's padded bf16 copy itself -- conv<b+2>_1 (pad 1) or fc6 (pad 3); pool1, pool2 and pool5 have no other reader, so
            // with option bf16_acts their fp32 tensors are not written (pool3 / pool4 feed the fp32 skip heads)
            char cons[32]; int ck = 3, cout_c = 0;
            if (b < 4) { snprintf(cons, sizeof cons, "conv%d_1", b + 2); cout_c = m->widths[b + 1]; } else { snprintf(cons, sizeof cons, "fc6"); ck = m->fc6k; cout_c = m->widths[5]; }
            unsigned short* yb = nullptr;
            if (m->bf16_acts && cin % 64 == 0 && cout_c % 64 == 0)
                yb = xg16_for(m, cons, N, h / 2, w / 2, cin, ck, s);
            const bool only16 = yb && b != 2 && b != 3;
            char ix[16]; snprintf(ix, sizeof ix, "pidx%d", b + 1);
            char pin[32]; snprintf(pin, sizeof pin, "pool%din", b + 1);
            auto pi = m->xg16.find(pin);
            const bool in16 = only16 && pi != m->xg16.end() && pi->second && m->xg16_filled.count(pin);      // the last conv wrote only its bf16 copy
            ProfScope ps(m, "maxpool_fwd", 0, (in16 ? 2.0 : 4.0) * N * h * w * cin + 4.0 * N * h * w * cin * (only16 ? 0.0625 : 0.3125) + (yb ? 0.5 * N * h * w * cin : 0.0));
            if (in16) launch_maxpool_fwd_route16(pi->second + g16_off(bf16_guard_rows(3, w + 2), cin), g16_ps(N, h, w, 3), (unsigned char*)A(m, ix), N, h, w, cin, s, yb, (ck - 1) / 2, g16_ps(N, h / 2, w / 2, ck));
            else launch_maxpool_fwd_route(x, only16 ? nullptr : A(m, pn), (unsigned char*)A(m, ix), N, h, w, cin, s, yb, (ck - 1) / 2, g16_ps(N, h / 2, w / 2, ck),
                                          /*round16=*/(b != 2 && b != 3) ? 1 : 0);
            m->pool_routed[b] = true; pooled = true;
            if (yb) m->xg16_filled.insert(cons);
            if (only16) { m->y_unwritten.insert(pn); m->in_bf16_only.insert(cons); }
        }
        if (!pooled) {
            // a block whose last conv did not run through the Winograd output transform (bf16 modes) but whose backward pass does run in the
            // Winograd domain: keep the same routing bytes, so that d(pool) is routed inside wino_dout_kernel and dZ is never written
            const bool route = train && cin % 4 == 0 && pool_backward_fused(m, b + 1, true);
            ProfScope ps(m, "maxpool_fwd", 0, 4.0 * N * h * w * cin * (route ? 1.3125 : 1.25));
            if (route) { char ix[16]; snprintf(ix, sizeof ix, "pidx%d", b + 1); launch_maxpool_fwd_route(x, A(m, pn), (unsigned char*)A(m, ix), N, h, w, cin, s); m->pool_fused[b] = true; }
            else launch_maxpool_fwd(x, A(m, pn), N, h, w, cin, s);
        }
        x = A(m, pn); h /= 2; w /= 2;
    }
    const int h5 = h, w5 = w;
    const bool drop = train && keep_prob < 1.f;
    m->drop_stream = (uint32_t)(2 * m->step);
    if (bf16_train_mode(m)) {
        unsigned short* xb6 = cp ? xg16_for(m, "fc6", N, h5, w5, m->widths[4], m->fc6k, s) : nullptr;
        if (xb6 && !m->xg16_filled.count("fc6")) { ProfScope ps(m, "bf16_convert", 0, 6.0 * N * h5 * w5 * m->widths[4]); launch_f32_to_bf16_padded(x, xb6, N, h5, w5, m->widths[4], (m->fc6k - 1) / 2, s, g16_ps(N, h5, w5, m->fc6k)); }
        unsigned short* xb7 = cp ? xg16_for(m, "fc7", N, h5, w5, m->widths[5], 1, s) : nullptr;
        const bool fuse7 = xb7 != nullptr && m->bf16_acts;      // fc7's input copy comes out of fc6's epilogue (16-byte stores since the tile kernel's epilogue goes through LDS)
        if (!bf16_conv_layer(m, "fc6_fwd_bf16", "fc6/weights", "fc6/biases", x, A(m, "fc6"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, drop, keep_prob, m->drop_stream, s, false, xb6, true,
                             fuse7 ? xb7 : nullptr, 0, g16_ps(N, h5, w5, m->fc6k), g16_ps(N, h5, w5, 1)))
            return fail(m, FCN8S_ERR_SHAPE, "bf16_train: fc6 does not fit the bf16 convolution kernel");
        if (xb7 && !fuse7) { ProfScope ps(m, "bf16_convert", 0, 6.0 * N * h5 * w5 * m->widths[5]); launch_f32_to_bf16_padded(A(m, "fc6"), xb7, N, h5, w5, m->widths[5], 0, s, g16_ps(N, h5, w5, 1)); }
        if (!bf16_conv_layer(m, "fc7_fwd_bf16", "fc7/weights", "fc7/biases", A(m, "fc6"), A(m, "fc7"), N, h5, w5, m->widths[5], m->widths[6], 1, drop, keep_prob, m->drop_stream + 1, s, false, xb7, true, nullptr, 0, g16_ps(N, h5, w5, 1)))
            return fail(m, FCN8S_ERR_SHAPE, "bf16_train: fc7 does not fit the bf16 convolution kernel");
    } else if (m->precision == FCN8S_PREC_BF16_FC || bf16_fwd_mode(m)) {
        // config 5: bf16-rounded operands, fp32 accumulate, fp32 epilogue and output (gemm_bf16.hip)
        bf16_conv_layer(m, "fc6_fwd_bf16", "fc6/weights", "fc6/biases", x, A(m, "fc6"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, drop, keep_prob, m->drop_stream, s);
        // the fp32 gradients of fc6 run in the Winograd domain and want the transformed input and this step's filter bank
        if (train) wino_backward_operands(m, "fc6", x, Wp(m, "fc6/weights"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, s);
        bf16_conv_layer(m, "fc7_fwd_bf16", "fc7/weights", "fc7/biases", A(m, "fc6"), A(m, "fc7"), N, h5, w5, m->widths[5], m->widths[6], 1, drop, keep_prob, m->drop_stream + 1, s);
    } else {
        {
            Epi e; e.bias = Wp(m, "fc6/biases"); e.relu = 1; e.dropout = drop; e.keep = keep_prob; e.stream_id = m->drop_stream;
            conv_same(m, "fc6_fwd", x, Wp(m, "fc6/weights"), A(m, "fc6"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, e, s, 0, "fc6");
        }
        {
            Epi e; e.bias = Wp(m, "fc7/biases"); e.relu = 1; e.dropout = drop; e.keep = keep_prob; e.stream_id = m->drop_stream + 1;
            conv_same(m, "fc7_fwd", A(m, "fc6"), Wp(m, "fc7/weights"), A(m, "fc7"), N, h5, w5, m->widths[5], m->widths[6], 1, e, s);
        }
    }
    // decoder (fcn8s_tensorflow.py:171-233)
    { Epi e; e.bias = Wp(m, "pool3_1x1/bias"); e.alpha = 0.0001f;
      conv_same(m, "score1x1_fwd", A(m, "pool3"), Wp(m, "pool3_1x1/kernel"), A(m, "p3"), N, H / 8, W / 8, m->widths[2], C, 1, e, s); }
    { Epi e; e.bias = Wp(m, "pool4_1x1/bias"); e.alpha = 0.01f;
      conv_same(m, "score1x1_fwd", A(m, "pool4"), Wp(m, "pool4_1x1/kernel"), A(m, "p4"), N, H / 16, W / 16, m->widths[3], C, 1, e, s); }
    { Epi e; e.bias = Wp(m, "fc7_1x1/bias");
      conv_same(m, "score1x1_fwd", A(m, "fc7"), Wp(m, "fc7_1x1/kernel"), A(m, "s7"), N, h5, w5, m->widths[6], C, 1, e, s); }
    tconv_fwd(m, A(m, "s7"), m->d_tph[0], Wp(m, "fc7_conv2d_trans/bias"), A(m, "p4"), A(m, "a4"), N, h5, w5, C, 4, 2, s);
    tconv_fwd(m, A(m, "a4"), m->d_tph[1], Wp(m, "fc7_pool4_conv2d_trans/bias"), A(m, "p3"), A(m, "a3"), N, H / 16, W / 16, C, 4, 2, s);
    if (m->tconv_gemm && m->logits_b) tconv_gemm_fwd(m);
    else tconv_fwd(m, A(m, "a3"), m->d_tph[2], Wp(m, "fc7_pool4_pool3_conv2d_trans/bias"), nullptr, A(m, "logits"), N, H / 8, W / 8, C, 16, 8, s);
    m->logits_nhwc_valid = !(m->tconv_gemm && m->logits_b);
    if (fill_fp) {                        // this pass (re)built the cache: remember what it was built from
        launch_fingerprint(m->d_params, (long long)m->total, m->d_fp, s);
        hipMemcpyAsync(&m->frozen_fp, m->d_fp, sizeof m->frozen_fp, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
    }
    if (guard_pending) {
        hipEventSynchronize(m->fp_event);
        if (*m->h_fp != m->frozen_fp) {            // the parameters changed behind the library's back: this pass used stale banks -- again, without them
            drop_banks();
            return forward(m, img_dev, dtype, keep_prob, train);
        }
    }
    m->have_forward = true; m->train_mode = train; m->keep_prob = keep_prob;
    return FCN8S_OK;
}

const char* kDecoderKernels[6] = {"pool3_1x1/kernel", "pool4_1x1/kernel", "fc7_1x1/kernel", "fc7_conv2d_trans/kernel",
                                  "fc7_pool4_conv2d_trans/kernel", "fc7_pool4_pool3_conv2d_trans/kernel"};

int compute_loss(fcn8s_model* m, const uint8_t* lab_dev, float l2_rate, bool with_grad)
{
    hipStream_t s = m->stream;
    const long long npix = (long long)m->N * m->H * m->W;
    const int nb = softmax_xent_blocks(npix);
    { ProfScope ps(m, "softmax_xent", 0, (double)npix * (m->C * 4 * (with_grad ? 2 : 1) + 1));
      if (with_grad) hipMemsetAsync(m->d_lastbias, 0, 64 * sizeof(float), s);
      const bool blk = m->tconv_gemm && m->logits_b;
      launch_softmax_xent(blk ? m->logits_b : A(m, "logits"), lab_dev, with_grad ? (blk ? m->dlogits_b : m->dlogits) : nullptr, m->d_partials, npix, m->C,
                          1.0f / (float)npix, s, with_grad ? m->d_lastbias : nullptr, blk ? &m->pm : nullptr, m->N); }
    const float* reg = nullptr;
    if (l2_rate != 0.f) {
        hipMemsetAsync(m->d_regsum, 0, sizeof(float), s);
        for (auto k : kDecoderKernels) launch_sumsq(Wp(m, k), m->d_regsum, (long long)P(m, k).numel, s);
        reg = m->d_regsum;
    }
    launch_finalize_loss(m->d_partials, nb, npix, reg, l2_rate, m->d_loss, s);
    // The loss is final here, a third of the way into a training step: queue its copy now, so that fcn8s_read_loss waits for this
    // point of the stream only and the host can go on queueing the next step while the backward pass runs (the reference fetches
    // the loss every step, fcn8s_tensorflow.py:554-578; waiting for the whole stream left the GPU idle for ~2 ms per step).
    m->loss_copied = m->h_loss && m->loss_ev && hipMemcpyAsync(m->h_loss, m->d_loss, sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess &&
                     hipEventRecord(m->loss_ev, s) == hipSuccess;
    m->l2_rate = l2_rate; m->have_loss = true;
    return FCN8S_OK;
}

// ---- backward --------------------------------------------------------------------------
void prepare_backward_weights(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    ProfScope ps(m, "weight_relayout", 0, 8.0 * ((double)m->widths[5] * m->widths[6] +
                                                 (double)m->C * (m->widths[2] + m->widths[3] + m->widths[6])));
    // (the 3x3 layers' and fc6's flipped + transposed kernels are made on demand in conv_same: the adjoint Winograd data gradients read
    // the forward filter banks)
    launch_flip_transpose(Wp(m, "fc7/weights"), WTp(m, "fc7/weights"), 1, m->widths[5], m->widths[6], s);
    launch_flip_transpose(Wp(m, "pool3_1x1/kernel"), WTp(m, "pool3_1x1/kernel"), 1, m->widths[2], m->C, s);
    launch_flip_transpose(Wp(m, "pool4_1x1/kernel"), WTp(m, "pool4_1x1/kernel"), 1, m->widths[3], m->C, s);
    launch_flip_transpose(Wp(m, "fc7_1x1/kernel"), WTp(m, "fc7_1x1/kernel"), 1, m->widths[6], m->C, s);
}

void l2_grad(fcn8s_model* m, const char* kernel)
{
    if (m->l2_rate != 0.f) launch_axpy(Gp(m, kernel), Wp(m, kernel), m->l2_rate, (long long)P(m, kernel).numel, m->stream);
}

// the gradients of bucket b are final behind the last kernel queued on `s`: record the event fcn8s_bucket_wait hands to other streams
void mark_bucket_final(fcn8s_model* m, int b, hipStream_t s)
{
    if (!m->bucket_ev[b]) hipEventCreateWithFlags(&m->bucket_ev[b], hipEventDisableTiming);
    hipEventRecord(m->bucket_ev[b], s);
    m->bucket_final[b] = true;
}

// backward phase 0: the decoder and fc7's weight gradient (bucket 0 = {fc7, decoder})
void backward_head(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    const int N = m->N, H = m->H, W = m->W, C = m->C;
    const int h5 = H / 32, w5 = W / 32, h4 = H / 16, w4 = W / 16, h3 = H / 8, w3 = W / 8;
    const float inv_keep = (m->train_mode && m->keep_prob < 1.f) ? 1.f / m->keep_prob : 1.f;
    hipMemsetAsync(m->d_grads, 0, m->total * sizeof(float), s);
    prepare_backward_weights(m);
    // logits = tconv16x16s8(a3)
    const bool blk = m->tconv_gemm && m->logits_b;
    if (blk) tconv_gemm_wgrad(m); else tconv_wgrad(m, A(m, "a3"), m->dlogits, Gp(m, "fc7_pool4_pool3_conv2d_trans/kernel"), N, h3, w3, C, 16, 8, s);
    launch_axpy(Gp(m, "fc7_pool4_pool3_conv2d_trans/bias"), m->d_lastbias, 1.f, C, s);      // column sums of dlogits, from the loss kernel
    if (blk) tconv_gemm_dgrad(m); else tconv_dgrad(m, m->dlogits, Wp(m, "fc7_pool4_pool3_conv2d_trans/kernel"), m->da3, N, h3, w3, C, 16, 8, s);
    l2_grad(m, "fc7_pool4_pool3_conv2d_trans/kernel");
    // a3 = tconv4x4s2(a4) + p3 ; p3 = conv1x1(pool3 * 1e-4)
    conv_wgrad(m, "score1x1_wgrad", A(m, "pool3"), m->da3, Gp(m, "pool3_1x1/kernel"), Gp(m, "pool3_1x1/bias"), N, h3, w3, m->widths[2], C, 1, 0.0001f, s);
    l2_grad(m, "pool3_1x1/kernel");
    { Epi e; e.alpha = 0.0001f; conv_same(m, "score1x1_dgrad", m->da3, WTp(m, "pool3_1x1/kernel"), m->gskip3, N, h3, w3, C, m->widths[2], 1, e, s); }
    tconv_wgrad(m, A(m, "a4"), m->da3, Gp(m, "fc7_pool4_conv2d_trans/kernel"), N, h4, w4, C, 4, 2, s);
    launch_colsum(m->da3, Gp(m, "fc7_pool4_conv2d_trans/bias"), (long long)N * h3 * w3, C, s);
    tconv_dgrad(m, m->da3, Wp(m, "fc7_pool4_conv2d_trans/kernel"), m->da4, N, h4, w4, C, 4, 2, s);
    l2_grad(m, "fc7_pool4_conv2d_trans/kernel");
    // a4 = tconv4x4s2(s7) + p4 ; p4 = conv1x1(pool4 * 1e-2)
    conv_wgrad(m, "score1x1_wgrad", A(m, "pool4"), m->da4, Gp(m, "pool4_1x1/kernel"), Gp(m, "pool4_1x1/bias"), N, h4, w4, m->widths[3], C, 1, 0.01f, s);
    l2_grad(m, "pool4_1x1/kernel");
    { Epi e; e.alpha = 0.01f; conv_same(m, "score1x1_dgrad", m->da4, WTp(m, "pool4_1x1/kernel"), m->gskip4, N, h4, w4, C, m->widths[3], 1, e, s); }
    tconv_wgrad(m, A(m, "s7"), m->da4, Gp(m, "fc7_conv2d_trans/kernel"), N, h5, w5, C, 4, 2, s);
    launch_colsum(m->da4, Gp(m, "fc7_conv2d_trans/bias"), (long long)N * h4 * w4, C, s);
    tconv_dgrad(m, m->da4, Wp(m, "fc7_conv2d_trans/kernel"), m->ds7, N, h5, w5, C, 4, 2, s);
    l2_grad(m, "fc7_conv2d_trans/kernel");
    // s7 = conv1x1(fc7)
    conv_wgrad(m, "score1x1_wgrad", A(m, "fc7"), m->ds7, Gp(m, "fc7_1x1/kernel"), Gp(m, "fc7_1x1/bias"), N, h5, w5, m->widths[6], C, 1, 1.f, s);
    l2_grad(m, "fc7_1x1/kernel");
    // (level 2 of the deferred weight gradients: fc7's dz gets a buffer of its own so that the weight gradient can run later, on the side stream)
    const bool defer_fc = m->defer_level_now >= 2 && m->acts.count("dz:fc7") && defer_ready(m);
    float* dz7 = defer_fc ? A(m, "dz:fc7") : m->gbuf[0];
    { Epi e; e.mask = A(m, "fc7"); e.mask_scale = inv_keep;
      conv_same(m, "score1x1_dgrad", m->ds7, WTp(m, "fc7_1x1/kernel"), dz7, N, h5, w5, C, m->widths[6], 1, e, s); }
    // fc7
    if (defer_fc) {
        hipEvent_t ev = defer_event(m); hipEventRecord(ev, s);
        m->deferred.emplace_back(ev, [m, dz7, N, h5, w5](hipStream_t ss) {
            conv_wgrad(m, "fc7_wgrad", A(m, "fc6"), dz7, Gp(m, "fc7/weights"), Gp(m, "fc7/biases"), N, h5, w5, m->widths[5], m->widths[6], 1, 1.f, ss, 0, "fc7"); });
    } else {
        conv_wgrad(m, "fc7_wgrad", A(m, "fc6"), dz7, Gp(m, "fc7/weights"), Gp(m, "fc7/biases"), N, h5, w5, m->widths[5], m->widths[6], 1, 1.f, s, 0, "fc7");
        mark_bucket_final(m, 0, s);
    }
    m->dz7_cur = dz7; m->defer_fc_cur = defer_fc;
}

// backward phase 1: fc7's data gradient, fc6 (bucket 1 = {fc6}: final behind its weight gradient, before its data gradient is queued)
void backward_fc6(fcn8s_model* m)
{
    hipStream_t s = m->stream;
    const int N = m->N, H = m->H, W = m->W;
    const int h5 = H / 32, w5 = W / 32;
    const float inv_keep = (m->train_mode && m->keep_prob < 1.f) ? 1.f / m->keep_prob : 1.f;
    float* dz7 = m->dz7_cur; const bool defer_fc = m->defer_fc_cur;
    { Epi e; e.mask = A(m, "fc6"); e.mask_scale = inv_keep; e.w_fwd = Wp(m, "fc7/weights"); e.yb_layer = "fc6"; e.yb_K = m->fc6k;      // (w_fwd + the layer name: the bf16_train branch of conv_same)
      conv_same(m, "fc7_dgrad", dz7, WTp(m, "fc7/weights"), m->gbuf[1], N, h5, w5, m->widths[6], m->widths[5], 1, e, s, 0, "fc7"); }
    // fc6
    if (defer_fc && m->acts.count("dmk:fc6") && m->acts.count("wv:fc6") && m->u_train.count("fc6#4")) {
        float* dz6 = m->gbuf[1];
        conv_wgrad(m, "fc6_wgrad", A(m, "pool5"), dz6, Gp(m, "fc6/weights"), Gp(m, "fc6/biases"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, 1.f, s, 0, "fc6", false, nullptr, 1);
        hipEvent_t ev = defer_event(m); hipEventRecord(ev, s);
        m->deferred.emplace_back(ev, [m, dz6, N, h5, w5](hipStream_t ss) {
            conv_wgrad(m, "fc6_wgrad", A(m, "pool5"), dz6, Gp(m, "fc6/weights"), Gp(m, "fc6/biases"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, 1.f, ss, 0, "fc6", false, nullptr, 2); });
    } else {
        conv_wgrad(m, "fc6_wgrad", A(m, "pool5"), m->gbuf[1], Gp(m, "fc6/weights"), Gp(m, "fc6/biases"), N, h5, w5, m->widths[4], m->widths[5], m->fc6k, 1.f, s, 0, "fc6");
        mark_bucket_final(m, 1, s);
    }
    { Epi e; e.dgrad = 1; e.w_fwd = Wp(m, "fc6/weights"); e.lazy_wt = 1;      // (flipped + transposed copy only if the adjoint path is not taken)
      conv_same(m, "fc6_dgrad", m->gbuf[1], WTp(m, "fc6/weights"), m->gbuf[0], N, h5, w5, m->widths[5], m->widths[4], m->fc6k, e, s, 0, "fc6"); }
    m->gcur = 0;   // gbuf[0] holds d(pool5)
}

// blocks [b_hi .. b_lo] (1-based VGG block numbers), going backwards
void backward_blocks(fcn8s_model* m, int b_hi, int b_lo)
{
    hipStream_t s = m->on_tail ? m->tail : m->stream;
    const int N = m->N;
    for (int b = b_hi; b >= b_lo; --b) {
        if (b == m->defer_start_block && !m->deferred.empty()) {
            // from here on the chain is HBM-bound: the held-back weight-gradient GEMMs run beside it -- on CUs of their own if masks are on
            // (ONE event, recorded before anything is put on the side stream: the caller's stream may be the legacy default stream, on which
            //  every operation -- an event record too -- first waits for all work already queued on blocking streams such as the masked ones)
            hipEvent_t here = defer_event(m);
            hipEventRecord(here, m->stream);
            if (m->tail) {
                hipStreamWaitEvent(m->tail, here, 0);
                s = m->tail; m->on_tail = true; m->launch_stream = m->tail;
            }
            flush_deferred(m, here);
        }
        const int h = m->H >> (b - 1), w = m->W >> (b - 1);      // resolution of this block's convs
        const int cw = m->widths[b - 1];
        const int nconv = kConvsPerBlock[b - 1];
        char last[32]; snprintf(last, sizeof last, "conv%d_%d", b, nconv);
        // d(pool_b) in gbuf[gcur] -> dZ of the last conv (ReLU mask fused).  If the forward pass kept the argmax bytes and both
        // gradients of that conv run through Winograd, the routing happens inside their shared transform and dZ is never written.
        const unsigned char* pidx = nullptr;
        {
            char ix[16]; snprintf(ix, sizeof ix, "pidx%d", b);
            const bool wino_both = pool_backward_fused(m, b, m->pool_fused[b - 1]);
            if (wino_both) pidx = (const unsigned char*)A(m, ix);
        }
        bool pool_done = false;
        if (!pidx && bf16_train_mode(m) && m->bf16_fuse_pool && m->train_mode && cw % 64 == 0) {
            // bf16_train: nobody reads the fp32 dZ of the block's last conv -- its weight and data gradients take the padded bf16 copy, its bias gradient the
            // column sums: the pool's backward kernel writes exactly those (gbuf[gcur ^ 1] stays unwritten; the "dz" handed on below is never dereferenced)
            unsigned short* dzb = g16_for(m, m->dyg16, m->dyg16_elems, last, N, h, w, cw, 3, s);
            if (dzb) {
                ProfScope ps(m, "maxpool_bwd", 0, 4.0 * N * h * w * cw * (m->pool_routed[b - 1] ? 0.3125 : 1.25) + 2.0 * N * h * w * cw);
                char ix[16]; snprintf(ix, sizeof ix, "pidx%d", b);
                pool_done = launch_maxpool_bwd_bf16(A(m, last), m->gbuf[m->gcur], dzb, Gp(m, std::string(last) + "/biases"), N, h, w, cw, s,
                                                    m->pool_routed[b - 1] ? (const unsigned char*)A(m, ix) : nullptr, g16_ps(N, h, w, 3));
            }
            if (pool_done) { m->dyg16_filled.insert(last); m->db_taken.insert(last); m->dz_unwritten.insert(last); m->gcur ^= 1; }
        }
        if (!pidx && !pool_done) {
            ProfScope ps(m, "maxpool_bwd", 0, 4.0 * N * h * w * cw * 2.25);
            launch_maxpool_bwd(A(m, last), m->gbuf[m->gcur], m->gbuf[m->gcur ^ 1], N, h, w, cw, 1, s);
            m->gcur ^= 1;
        }
        for (int i = nconv; i >= 1; --i) {
            char nm[32]; snprintf(nm, sizeof nm, "conv%d_%d", b, i);
            const float* dz = m->gbuf[m->gcur];
            const float* xin; int cin; int real_cin = 0;
            char inname[32];
            if (i > 1) { snprintf(inname, sizeof inname, "conv%d_%d", b, i - 1); xin = A(m, inname); cin = cw; }
            else if (b > 1) { snprintf(inname, sizeof inname, "pool%d", b - 1); xin = A(m, inname); cin = m->widths[b - 2]; }
            else { xin = A(m, "x0"); cin = 4; real_cin = 3; }
            const bool first = (b == 1 && i == 1);
            // the data-gradient conv (cw -> cin channels) takes the Winograd path under the same conditions as conv_same()
            const bool dgrad_wino = !first && m->wino_min_cin > 0 && cw >= m->wino_min_cin && m->d_wino_v && wino_tile_for(m, h, w) >= 4 &&
                                    cw % 16 == 0 && cin % 64 == 0;
            const unsigned char* pix = i == nconv ? pidx : nullptr;
            const bool defer = m->defer_level_now >= 1 && b > m->defer_start_block && dgrad_wino && wino_tile_for(m, h, w) == 6 && cw % 64 == 0 &&
                               m->acts.count(std::string("dmk:") + nm) && m->acts.count(std::string("wv:") + nm) && defer_ready(m);
            if (defer) {
                conv_wgrad(m, "conv3x3_wgrad", xin, dz, Gp(m, std::string(nm) + "/filter"), Gp(m, std::string(nm) + "/biases"),
                           N, h, w, cin, cw, 3, 1.f, s, 0, nm, true, pix, 1);
                hipEvent_t ev = defer_event(m); hipEventRecord(ev, s);
                const std::string lname = nm;
                m->deferred.emplace_back(ev, [m, lname, xin, dz, N, h, w, cin, cw, pix](hipStream_t ss) {
                    conv_wgrad(m, "conv3x3_wgrad", xin, dz, Gp(m, lname + "/filter"), Gp(m, lname + "/biases"), N, h, w, cin, cw, 3, 1.f, ss, 0, lname.c_str(), true, pix, 2); });
            } else
            conv_wgrad(m, first ? "conv1_1_wgrad" : "conv3x3_wgrad", xin, dz, Gp(m, std::string(nm) + "/filter"), Gp(m, std::string(nm) + "/biases"),
                       N, h, w, cin, cw, 3, 1.f, s, real_cin, nm, dgrad_wino, pix);
            if (first) break;
            Epi e; e.dgrad = 1; e.w_fwd = Wp(m, std::string(nm) + "/filter"); e.lazy_wt = 1;
            if (i > 1) {                                                   // ReLU of the previous conv
                e.mask = xin; e.mask_scale = 1.f; e.yb_layer = inname; e.yb_K = 3;
                if (m->rbits_ok.count(inname)) e.relu_bits_in = (const unsigned*)A(m, (std::string("rb:") + inname).c_str());
                // The previous conv takes this gradient only through dM = A dZ A^T (weight gradient in the Winograd domain, adjoint data
                // gradient): the gather kernel can write dM directly.  Conditions = those of conv_wgrad's adjoint branch for that layer.
                const int cin_prev = i > 2 ? cw : (b > 1 ? m->widths[b - 2] : 4);
                const bool prev_first = (b == 1 && i == 2);
                e.yb_only = !prev_first && cin_prev % 64 == 0 && cw % 64 == 0;      // (conv1_1's weight gradient is exact fp32 and reads the fp32 tensor)
                if (m->fuse_dgrad_dout && e.relu_bits_in && !prev_first && m->defer_level_now == 0 && m->train_mode && m->d_wino_m &&
                    wino_tile_for(m, h, w) == 6 && cw % 64 == 0 && cin_prev % 64 == 0 && m->acts.count(std::string("wv:") + inname)) {
                    e.dm_out = m->d_wino_m; e.dm_out_layer = inname;
                }
            }
            else if (b == 5) e.addend = m->gskip4;                        // d(pool4) also receives the pool4_1x1 path
            else if (b == 4) e.addend = m->gskip3;                        // d(pool3) also receives the pool3_1x1 path
            conv_same(m, "conv3x3_dgrad", dz, WTp(m, std::string(nm) + "/filter"), m->gbuf[m->gcur ^ 1], N, h, w, cw, cin, 3, e, s, 0, nm);
            m->gcur ^= 1;
        }
    }
}

int bucket_complete_after(const fcn8s_model* m, int bucket)
{
    if (m->defer_wgrad == 0 || (bucket <= 1 && m->defer_wgrad < 3)) return bucket;
    return kNumBuckets - 1;           // weight gradients of conv3_1 .. conv5_3 (level 3: fc6 / fc7 too) are held back until the last call
}

// level_cap: 1 for the bucket-by-bucket API (buckets 0 and 1 -- fc7 + decoder, fc6 -- are final when their calls return, so that their
// all-reduces can start at once; with deferred weight gradients the conv buckets are final after the last call, see
// fcn8s_bucket_complete_after), 2 for the fused step
int do_backward_bucket(fcn8s_model* m, int bucket, int level_cap)
{
    t_deterministic = m->deterministic;
    if (!m->have_loss || !m->train_mode) return fail(m, FCN8S_ERR_STATE, "fcn8s_backward_bucket: call fcn8s_forward_loss first");
    if (bucket != m->next_bucket) return fail(m, FCN8S_ERR_STATE, "fcn8s_backward_bucket: buckets must be run in order 0, 1, ... fcn8s_num_buckets() - 1");
    if (bucket == 0) {
        m->defer_level_now = m->defer_wgrad >= 3 ? 2 : std::min(m->defer_wgrad, level_cap);      // 3: the caller does not consume bucket 0 early
        if (m->profile && m->profile_detail) m->defer_level_now = 0;      // per-layer timing wants one kernel at a time
        // bf16_train has no Winograd-domain weight gradients to hold back, and a held-back fc6 / fc7 weight gradient would have to convert its operands on the side
        // stream: the arithmetic of the mode must not depend on "defer_wgrad" (round 5's deferred fc7 lambda ran the fp32 kernel: ADVICE round 5)
        if (bf16_train_mode(m)) m->defer_level_now = 0;
        m->deferred.clear(); m->ev_next = 0; m->dyg16_filled.clear(); m->db_taken.clear(); m->dy_bf16_only.clear(); m->dz_unwritten.clear();
        m->dm_prefilled.clear();                                           // (a promise left over from a backward pass that ended in an error)
        m->on_tail = false; m->launch_stream = nullptr;                    // (a backward pass that ended in an error may have left them set)
        for (int b = 0; b < kNumBuckets; ++b) m->bucket_final[b] = false;
        // (a caller that skipped fcn8s_apply_update after fcn8s_allreduce_bucket: the gradient buffer is about to be cleared and rewritten,
        //  so this stream first waits for whatever the library's communicator still has in flight on it)
        for (int b = 0; b < kNumBuckets; ++b)
            if (m->comm_pending[b]) { hipStreamWaitEvent(m->stream, m->comm_done[b], 0); m->comm_pending[b] = false; }
        backward_head(m);
    }
    else if (bucket == 1) backward_fc6(m);
    else if (bucket == 2) backward_blocks(m, 5, 4);
    else { backward_blocks(m, 3, 1); join_deferred(m); }
    // whatever this call completed and no kernel-exact point has marked yet (the conv buckets; everything held back to the last call)
    // (the fused step, level_cap 2, marks nothing before its last call -- and there everything that is still open: bucket 2 as well)
    for (int b = 0; b < kNumBuckets; ++b)
        if (!m->bucket_final[b] && ((level_cap == 1 && bucket_complete_after(m, b) == bucket) || bucket == kNumBuckets - 1)) mark_bucket_final(m, b, m->stream);
    m->next_bucket = bucket + 1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(m, FCN8S_ERR_HIP, std::string("backward launch: ") + hipGetErrorString(e));
    return FCN8S_OK;
}

int ensure_opt_state(fcn8s_model* m)
{
    if (m->d_m) return FCN8S_OK;
    HIPCHK(m, hipMalloc((void**)&m->d_m, m->total * sizeof(float)));
    HIPCHK(m, hipMalloc((void**)&m->d_v, m->total * sizeof(float)));
    HIPCHK(m, hipMemsetAsync(m->d_m, 0, m->total * sizeof(float), m->stream));
    HIPCHK(m, hipMemsetAsync(m->d_v, 0, m->total * sizeof(float), m->stream));
    return FCN8S_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

size_t fcn8s_param_floats(const fcn8s_config* cfg)
{
    if (!cfg || cfg->num_classes <= 0) return 0;
    int C, widths[7], k; resolve_cfg(cfg, C, widths, k);
    std::vector<ParamInfo> t; size_t total, bo[FCN8S_MAX_BUCKETS], bn[FCN8S_MAX_BUCKETS];
    build_param_table(C, widths, k, t, total, bo, bn);
    return total;
}

int fcn8s_layout_num_params(const fcn8s_config* cfg)
{
    if (!cfg || cfg->num_classes <= 0) return 0;
    int C, widths[7], k; resolve_cfg(cfg, C, widths, k);
    std::vector<ParamInfo> t; size_t total, bo[FCN8S_MAX_BUCKETS], bn[FCN8S_MAX_BUCKETS];
    build_param_table(C, widths, k, t, total, bo, bn);
    return (int)t.size();
}
int fcn8s_layout_param(const fcn8s_config* cfg, int index, char name_out[64], int32_t* ndim, int64_t shape[4], int64_t* off)
{
    if (!cfg || cfg->num_classes <= 0) return FCN8S_ERR_BAD_ARG;
    int C, widths[7], k; resolve_cfg(cfg, C, widths, k);
    std::vector<ParamInfo> t; size_t total, bo[FCN8S_MAX_BUCKETS], bn[FCN8S_MAX_BUCKETS];
    build_param_table(C, widths, k, t, total, bo, bn);
    if (index < 0 || index >= (int)t.size()) return FCN8S_ERR_BAD_ARG;
    if (name_out) { strncpy(name_out, t[index].name.c_str(), 63); name_out[63] = 0; }
    if (ndim) *ndim = t[index].ndim;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = t[index].shape[i];
    if (off) *off = (int64_t)t[index].offset;
    return FCN8S_OK;
}
int fcn8s_layout_num_buckets(const fcn8s_config* cfg) { return (cfg && cfg->num_classes > 0) ? kNumBuckets : 0; }
int fcn8s_layout_bucket(const fcn8s_config* cfg, int bucket, size_t* off, size_t* n)
{
    if (!cfg || cfg->num_classes <= 0 || bucket < 0 || bucket >= kNumBuckets) return FCN8S_ERR_BAD_ARG;
    int C, widths[7], k; resolve_cfg(cfg, C, widths, k);
    std::vector<ParamInfo> t; size_t total, bo[FCN8S_MAX_BUCKETS], bn[FCN8S_MAX_BUCKETS];
    build_param_table(C, widths, k, t, total, bo, bn);
    if (off) *off = bo[bucket];
    if (n) *n = bn[bucket];
    return FCN8S_OK;
}

int fcn8s_create(const fcn8s_config* cfg, fcn8s_model** out)
{
    if (!cfg || !out) return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_create: null argument");
    if (cfg->num_classes <= 0 || cfg->num_classes % 4 != 0 || cfg->num_classes > 64)
        return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_create: num_classes must be a positive multiple of 4, at most 64 (the Python facade pads other class counts)");
    fcn8s_model* m = new fcn8s_model();
    resolve_cfg(cfg, m->C, m->widths, m->fc6k);
    for (int i = 0; i < 7; ++i)
        if (m->widths[i] % 4) { delete m; return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_create: channel widths must be multiples of 4"); }
    if (m->fc6k % 2 == 0) { delete m; return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_create: fc6_ksize must be odd"); }
    m->device = cfg->device_id; m->seed = cfg->seed;
    hipError_t e = hipSetDevice(m->device);
    if (e != hipSuccess) { delete m; return fail(nullptr, FCN8S_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e) + " (libfcn8s_hip needs an AMD GPU; there is no CPU fallback)"); }
    {   // the kernels are written for gfx950: up to 160 KB of LDS per workgroup (conv_bf16_256_kernel), 96 KB static in wino_out_in_kernel.  Say so
        // here, once, instead of failing every launch with a generic error on a part that has 64 KB.
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.sharedMemPerBlock < FCN8S_LDS_BYTES_NEEDED) {
            const std::string msg = std::string("fcn8s_create: device ") + std::to_string(m->device) + " (" + prop.gcnArchName + ") offers " + std::to_string(prop.sharedMemPerBlock) +
                                    " bytes of LDS per workgroup; this library is written for gfx950 (MI355X) and needs " + std::to_string(FCN8S_LDS_BYTES_NEEDED);
            delete m; return fail(nullptr, FCN8S_ERR_HIP, msg);
        }
    }
    build_param_table(m->C, m->widths, m->fc6k, m->params, m->total, m->bucket_off, m->bucket_n);
    for (size_t i = 0; i < m->params.size(); ++i) m->index[m->params[i].name] = (int)i;
    const size_t bytes = m->total * sizeof(float);
    auto bail = [&](const char* what, hipError_t err) { std::string msg = std::string(what) + ": " + hipGetErrorString(err); fcn8s_destroy(m); return fail(nullptr, FCN8S_ERR_OOM, msg); };
    if (cfg->ext_params) m->d_params = (float*)cfg->ext_params;
    else { if ((e = hipMalloc((void**)&m->d_params, bytes)) != hipSuccess) return bail("hipMalloc(params)", e); m->own_params = true; hipMemset(m->d_params, 0, bytes); }
    if (cfg->ext_grads) m->d_grads = (float*)cfg->ext_grads;
    else { if ((e = hipMalloc((void**)&m->d_grads, bytes)) != hipSuccess) return bail("hipMalloc(grads)", e); m->own_grads = true; hipMemset(m->d_grads, 0, bytes); }
    if ((e = hipMalloc((void**)&m->d_wt, bytes)) != hipSuccess) return bail("hipMalloc(wt)", e);
    // 12 taps x 4 channels: rows 36..47 (three padding taps of the LDS-DMA conv1_1 kernel) and the 64 floats behind them stay zero
    if ((e = hipMalloc((void**)&m->d_w1pad, (12 * 4 * (size_t)m->widths[0] + 64) * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    hipMemset(m->d_w1pad, 0, (12 * 4 * (size_t)m->widths[0] + 64) * sizeof(float));
    const size_t cc = (size_t)m->C * m->C;
    if ((e = hipMalloc((void**)&m->d_tph[0], 16 * cc * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&m->d_tph[1], 16 * cc * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&m->d_tph[2], 256 * cc * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    {
        int cmax = 0; for (int i = 0; i < 5; ++i) cmax = std::max(cmax, m->widths[i]);
        size_t ufl = 64 * (size_t)cmax * cmax;                        // F(6x6,3x3): 64 positions
        if (m->fc6k == 7) ufl = std::max(ufl, 49 * 4 * (size_t)m->widths[4] * m->widths[5]);   // fc6: 49 positions x 4 sub-filters
        if ((e = hipMalloc((void**)&m->d_wino_u, ufl * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
        m->ufl = ufl;
    }
    if ((e = hipMalloc((void**)&m->d_loss, (2 + 64) * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    m->d_regsum = m->d_loss + 1; m->d_lastbias = m->d_loss + 2;
    if ((e = hipMalloc((void**)&m->d_conf, cc * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&m->d_fp, sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    m->tg_kp = (4 * m->C + 63) / 64 * 64;
    {
        const size_t NC = 64 * (size_t)m->C, KP = (size_t)m->tg_kp;
        if ((e = hipMalloc((void**)&m->tg_b2, 4 * (size_t)m->C * NC * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMalloc((void**)&m->tg_b2t, NC * KP * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMalloc((void**)&m->tg_db2, KP * NC * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMalloc((void**)&m->tg_bias, NC * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    }
    hipMemset(m->d_conf, 0, cc * sizeof(unsigned long long));
    hipMemset(m->d_loss, 0, 2 * sizeof(float));
    if (hipHostMalloc((void**)&m->h_loss, 64, hipHostMallocDefault) != hipSuccess) { m->h_loss = nullptr; (void)hipGetLastError(); }
    if (hipEventCreateWithFlags(&m->loss_ev, hipEventDisableTiming) != hipSuccess) { m->loss_ev = nullptr; (void)hipGetLastError(); }
    *out = m;
    return FCN8S_OK;
}

int fcn8s_destroy(fcn8s_model* m)
{
    if (!m) return FCN8S_OK;
    // the communicator first: its stream is drained by polling (a dead peer ends in an abort after comm_timeout_ms), and only then is the device
    // synchronised -- the other way round a collective that waits for a peer that is gone would hold hipDeviceSynchronize until the watchdog fires
    const int rc_comm = fcn8s_comm_destroy(m);
    const std::string comm_text = rc_comm ? m->err : std::string();
    hipDeviceSynchronize();
    // the per-stream scratch of the streams this model ran on (the caller's stream may live on: what it holds is given back, the next user grows its own)
    scratch_release(m->stream); if (m->side) scratch_release(m->side); if (m->tail) scratch_release(m->tail);
    for (auto& g : m->groups) for (auto& ev : g.ev) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    if (m->own_params && m->d_params) hipFree(m->d_params);
    if (m->own_grads && m->d_grads) hipFree(m->d_grads);
    if (m->d_m) hipFree(m->d_m);
    if (m->d_v) hipFree(m->d_v);
    if (m->d_wt) hipFree(m->d_wt);
    if (m->d_w1pad) hipFree(m->d_w1pad);
    if (m->d_wino_u2) hipFree(m->d_wino_u2);
    if (m->side && m->side_owned) hipStreamDestroy(m->side);
    if (m->tail_done) hipEventDestroy(m->tail_done);
    if (m->side_done) hipEventDestroy(m->side_done);
    for (auto e : m->ev_pool) hipEventDestroy(e);
    for (auto& e : m->bucket_ev) if (e) { hipEventDestroy(e); e = nullptr; }
    for (auto& kv : m->kept_dy) if (kv.second.p) hipFree(kv.second.p);
    for (auto& kv : m->u_train) if (kv.second) hipFree(kv.second);
    if (m->d_wino_u) hipFree(m->d_wino_u);
    for (auto& kv : m->u_cache) if (kv.second) hipFree(kv.second);
    for (auto& kv : m->wbf16_cache) if (kv.second) hipFree(kv.second);
    if (m->d_wbf16) hipFree(m->d_wbf16);
    if (m->d_abf16) hipFree(m->d_abf16);
    for (auto& kv : m->xbf16) if (kv.second) hipFree(kv.second);
    for (auto& kv : m->xg16) if (kv.second) hipFree(kv.second);
    for (auto& kv : m->dyg16) if (kv.second) hipFree(kv.second);
    for (int i = 0; i < 3; ++i) if (m->d_tph[i]) hipFree(m->d_tph[i]);
    if (m->d_loss) hipFree(m->d_loss);
    if (m->h_loss) hipHostFree(m->h_loss);
    if (m->loss_ev) hipEventDestroy(m->loss_ev);
    if (m->d_conf) hipFree(m->d_conf);
    if (m->d_fp) hipFree(m->d_fp);
    if (m->h_fp) hipHostFree(m->h_fp);
    if (m->fp_event) hipEventDestroy(m->fp_event);
    if (m->tg_b2) hipFree(m->tg_b2);
    if (m->tg_b2t) hipFree(m->tg_b2t);
    if (m->tg_db2) hipFree(m->tg_db2);
    if (m->tg_bias) hipFree(m->tg_bias);
    for (auto& sl : m->slots) {
        if (sl.h_img) hipHostFree(sl.h_img);
        if (sl.h_lab) hipHostFree(sl.h_lab);
        if (sl.d_img) hipFree(sl.d_img);
        if (sl.d_lab) hipFree(sl.d_lab);
        if (sl.ready) hipEventDestroy(sl.ready);
        if (sl.consumed) hipEventDestroy(sl.consumed);
    }
    if (m->copy_stream) hipStreamDestroy(m->copy_stream);
    if (m->arena) hipFree(m->arena);
    delete m;
    // the model is gone either way; a communicator that had failed is reported once, with its reason in fcn8s_last_error(NULL)
    if (rc_comm) { g_last_error = comm_text; return rc_comm; }
    return FCN8S_OK;
}

const char* fcn8s_last_error(const fcn8s_model* m) { return m ? m->err.c_str() : g_last_error.c_str(); }

static void drop_u_cache(fcn8s_model* m)
{
    if (m->u_cache.empty() && m->wbf16_cache.empty()) return;
    hipStreamSynchronize(m->stream);
    for (auto& kv : m->u_cache) if (kv.second) hipFree(kv.second);
    m->u_cache.clear();
    for (auto& kv : m->wbf16_cache) if (kv.second) hipFree(kv.second);
    m->wbf16_cache.clear();
}
int fcn8s_freeze_params(fcn8s_model* m, int frozen)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (!frozen || !m->frozen) drop_u_cache(m);          // entering or leaving: start from an empty cache
    m->frozen = frozen != 0;
    return FCN8S_OK;
}

int fcn8s_set_precision(fcn8s_model* m, int precision)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (precision < FCN8S_PREC_F32 || precision > FCN8S_PREC_BF16_TRAIN)
        return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_set_precision: unknown precision");
    if (precision == FCN8S_PREC_BF16_TRAIN) {
        for (int i = 0; i < 7; ++i)
            if (m->widths[i] % 64) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_set_precision: the bf16_train mode needs every channel width to be a multiple of 64");
    }
    if (precision == FCN8S_PREC_BF16_FC || precision == FCN8S_PREC_BF16_FWD || precision == FCN8S_PREC_BF16_FWD_X2) {
        if (m->widths[4] % 32 || m->widths[5] % 128 || m->widths[6] % 128)
            return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_set_precision: the bf16 modes need conv5 width % 32 == 0 and fc6 / fc7 widths % 128 == 0");
    }
    if (precision != m->precision) {
        // the forward filter banks kept for the adjoint data gradients belong to the arithmetic that made them (and a mode whose forward
        // pass does not refresh a bank must never find an old one)
        HIPCHK(m, hipStreamSynchronize(m->stream));
        for (auto& kv : m->u_train) if (kv.second) hipFree(kv.second);
        m->u_train.clear();
        drop_u_cache(m);
        for (auto& kv : m->xbf16) if (kv.second) hipFree(kv.second);      // (the padded bf16 activation copies of the bf16 forward modes)
        m->xbf16.clear();
        for (auto& kv : m->xg16) if (kv.second) hipFree(kv.second);
        m->xg16.clear(); m->xg16_elems.clear();
        for (auto& kv : m->dyg16) if (kv.second) hipFree(kv.second);
        m->dyg16.clear(); m->dyg16_elems.clear(); m->xg16_filled.clear(); m->dyg16_filled.clear();
        // bf16_train runs every convolution but conv1_1 as a DIRECT convolution on the bf16 MFMA, forward and backward: it rides on the library's
        // direct path (no Winograd transforms, the pools as kernels of their own, ReLU masks from the activations), i.e. on the settings
        // winograd_min_cin = 0 / winograd_fc6 = 0, which it takes over while it is on (and which need another workspace)
        const bool was = m->precision == FCN8S_PREC_BF16_TRAIN, now = precision == FCN8S_PREC_BF16_TRAIN;
        if (was != now) {
            if (now) { m->saved_wino_min_cin = m->wino_min_cin; m->saved_wino_fc6 = m->wino_fc6; m->wino_min_cin = 0; m->wino_fc6 = 0; }
            else { if (m->saved_wino_min_cin >= 0) m->wino_min_cin = m->saved_wino_min_cin; if (m->saved_wino_fc6 >= 0) m->wino_fc6 = m->saved_wino_fc6; }
            if (m->arena) { hipFree(m->arena); m->arena = nullptr; m->arena_bytes = 0; m->N = m->H = m->W = 0; m->acts.clear(); }
            m->have_forward = m->have_loss = false;
        }
    }
    m->precision = precision;
    return FCN8S_OK;
}
int fcn8s_get_precision(const fcn8s_model* m) { return m ? m->precision : -1; }

// Options select between maintained algorithm variants (parity reports separate the Winograd round-off from the summation order
// with them); the defaults are the measured winners.  A model option drops the workspace and every cached filter bank.
static int* model_option(fcn8s_model* m, const std::string& key)
{
    if (key == "winograd_min_cin") return &m->wino_min_cin;
    if (key == "winograd_tile") return &m->wino_tile;
    if (key == "winograd_fc6") return &m->wino_fc6;
    if (key == "tconv_gemm") return &m->tconv_gemm;
    if (key == "fuse_dgrad_dout") return &m->fuse_dgrad_dout;
    if (key == "fuse_out_in") return &m->fuse_out_in;
    if (key == "defer_wgrad") return &m->defer_wgrad;
    if (key == "defer_start_block") return &m->defer_start_block;
    if (key == "defer_tail_cus") return &m->defer_tail_cus;
    if (key == "bf16_gemm256") return &m->bf16_gemm256;
    if (key == "winograd_tile_hires") return &m->wino_tile_hires;
    if (key == "winograd_hires_pixels") return &m->wino_hires_pixels;
    if (key == "conv1_tiled") return &m->conv1_tiled;
    if (key == "conv1_wgrad_mfma") return &m->conv1_wgrad_mfma;
    if (key == "bf16_copy_by_transform") return &m->bf16_copy_by_transform;
    if (key == "conv1_in_transform") return &m->conv1_in_transform;
    if (key == "deterministic") return &m->deterministic;
    if (key == "bf16_fuse_convert") return &m->bf16_fuse_convert;
    if (key == "bf16_acts") return &m->bf16_acts;
    if (key == "bf16_rows_bn") return &m->bf16_rows_bn;
    if (key == "keep_output_gradients") return &m->keep_dy;
    if (key == "bf16_infer_copies") return &m->bf16_infer_copies;
    if (key == "bf16_fuse_pool") return &m->bf16_fuse_pool;
    return nullptr;
}
int fcn8s_set_option(fcn8s_model* m, const char* key, int64_t value)
{
    if (!key) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_set_option: null key");
    const std::string k = key;
    if (!m) {
        if (k == "op_f32x3") { t_op_split = value ? 3 : 0; return FCN8S_OK; }
        if (k == "op_deterministic") { t_deterministic = value ? 1 : 0; return FCN8S_OK; }
        if (k == "op_bf16_planes") { t_op_planes = value ? 1 : 0; return FCN8S_OK; }
        if (k == "op_bf16_rows_bn") { if (value != 0 && value != 64 && value != 128) return fail(nullptr, FCN8S_ERR_BAD_ARG, "op_bf16_rows_bn must be 0, 64 or 128"); t_op_rows_bn = (int)value; return FCN8S_OK; }
        if (k == "op_split_pieces") {
            if (value != 0 && value != 2 && value != 3) return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_set_option: op_split_pieces is 0, 2 or 3");
            t_op_split = (int)value; return FCN8S_OK;
        }
        return fail(nullptr, FCN8S_ERR_NOT_FOUND, "fcn8s_set_option: unknown op-context option '" + k + "' (model options need a model)");
    }
    if (k == "comm_timeout_ms") {
        if (value < 1) return fail(m, FCN8S_ERR_BAD_ARG, "comm_timeout_ms must be >= 1");
        std::lock_guard<std::mutex> lk(m->comm_mu); m->comm_timeout_ms = value; return FCN8S_OK;
    }
    if (k == "conv1_tiled" || k == "conv1_wgrad_mfma" || k == "bf16_copy_by_transform" || k == "conv1_in_transform" || k == "deterministic" || k == "bf16_fuse_convert" || k == "bf16_acts" || k == "bf16_rows_bn" || k == "bf16_fuse_pool" || k == "keep_output_gradients" || k == "bf16_infer_copies") {        // pick a kernel per launch: nothing cached depends on them
        *model_option(m, k) = k == "bf16_rows_bn" ? (int)value : (value ? 1 : 0);
        return FCN8S_OK;
    }
    int* slot = model_option(m, k);
    if (!slot) return fail(m, FCN8S_ERR_NOT_FOUND, "fcn8s_set_option: unknown option '" + k + "'");
    // FCN8S_PREC_BF16_TRAIN runs without Winograd transforms whatever these two say: while the mode is on a new value is kept for the day it is left
    // (round 5 overwrote the live slot: the value was used by nobody and lost on leaving)
    if (bf16_train_mode(m) && (k == "winograd_min_cin" || k == "winograd_fc6")) {
        if (value < 0) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_set_option: negative value");
        (k == "winograd_min_cin" ? m->saved_wino_min_cin : m->saved_wino_fc6) = (int)value;
        return FCN8S_OK;
    }
    if (k == "winograd_tile" && value != 2 && value != 4 && value != 6) return fail(m, FCN8S_ERR_BAD_ARG, "winograd_tile must be 2, 4 or 6");
    if (k == "winograd_tile_hires" && value != 0 && value != 2 && value != 4 && value != 6) return fail(m, FCN8S_ERR_BAD_ARG, "winograd_tile_hires must be 0, 2, 4 or 6");
    if (k == "winograd_min_cin" && value < 0) return fail(m, FCN8S_ERR_BAD_ARG, "winograd_min_cin must be >= 0 (0 = direct convolution everywhere)");
    if (k == "defer_wgrad" && (value < 0 || value > 3)) return fail(m, FCN8S_ERR_BAD_ARG, "defer_wgrad must be 0 .. 3");
    if (k == "fuse_out_in" && (value < 0 || value > 2)) return fail(m, FCN8S_ERR_BAD_ARG, "fuse_out_in must be 0, 1 or 2");
    if (k == "defer_start_block" && (value < 1 || value > 4)) return fail(m, FCN8S_ERR_BAD_ARG, "defer_start_block must be 1 .. 4");
    if (k == "defer_tail_cus" && (value < 0 || value > 248 || value % 8)) return fail(m, FCN8S_ERR_BAD_ARG, "defer_tail_cus must be a multiple of 8 in 0 .. 248");
    if (*slot == (int)value) return FCN8S_OK;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    *slot = (k == "winograd_fc6" || k == "tconv_gemm") ? (value != 0) : (int)value;
    if (k == "defer_tail_cus" && m->side) {         // the side / tail streams are re-made for the new split on next use
        hipDeviceSynchronize();
        if (m->side_owned) hipStreamDestroy(m->side);
        m->side = m->tail = nullptr;
        if (m->side_done) { hipEventDestroy(m->side_done); m->side_done = nullptr; }
        if (m->tail_done) { hipEventDestroy(m->tail_done); m->tail_done = nullptr; }
    }
    if (m->arena) { hipFree(m->arena); m->arena = nullptr; m->arena_bytes = 0; m->N = m->H = m->W = 0; m->acts.clear(); }
    m->have_forward = m->have_loss = false;
    for (auto& kv : m->u_cache) if (kv.second) hipFree(kv.second);
    for (auto& kv : m->wbf16_cache) if (kv.second) hipFree(kv.second);
    m->wbf16_cache.clear();
    m->u_cache.clear();
    for (auto& kv : m->u_train) if (kv.second) hipFree(kv.second);
    m->u_train.clear();
    return FCN8S_OK;
}
int fcn8s_get_option(const fcn8s_model* m, const char* key, int64_t* value)
{
    if (!key || !value) return FCN8S_ERR_BAD_ARG;
    const std::string k = key;
    if (!m) {
        if (k == "op_f32x3") { *value = t_op_split == 3; return FCN8S_OK; }
        if (k == "op_deterministic") { *value = t_deterministic; return FCN8S_OK; }
        if (k == "op_bf16_planes") { *value = t_op_planes; return FCN8S_OK; }
        if (k == "op_bf16_rows_bn") { *value = t_op_rows_bn; return FCN8S_OK; }
        if (k == "op_split_pieces") { *value = t_op_split; return FCN8S_OK; }
        return FCN8S_ERR_NOT_FOUND;
    }
    if (k == "comm_timeout_ms") { *value = m->comm_timeout_ms; return FCN8S_OK; }
    const int* slot = model_option(const_cast<fcn8s_model*>(m), k);
    if (!slot) return FCN8S_ERR_NOT_FOUND;
    *value = *slot;
    // (the caller's own setting, also while FCN8S_PREC_BF16_TRAIN keeps the live slots at 0)
    if (bf16_train_mode(m) && k == "winograd_min_cin" && m->saved_wino_min_cin >= 0) *value = m->saved_wino_min_cin;
    if (bf16_train_mode(m) && k == "winograd_fc6" && m->saved_wino_fc6 >= 0) *value = m->saved_wino_fc6;
    return FCN8S_OK;
}

int fcn8s_set_stream(fcn8s_model* m, void* s) { if (!m) return FCN8S_ERR_BAD_ARG; m->stream = (hipStream_t)s; return FCN8S_OK; }
int fcn8s_synchronize(fcn8s_model* m) { if (!m) return FCN8S_ERR_BAD_ARG; HIPCHK(m, hipStreamSynchronize(m->stream)); return FCN8S_OK; }

int fcn8s_num_params(const fcn8s_model* m) { return m ? (int)m->params.size() : 0; }
int fcn8s_param_info(const fcn8s_model* m, int i, const char** name, int32_t* ndim, int64_t shape[4], int64_t* off)
{
    if (!m || i < 0 || i >= (int)m->params.size()) return FCN8S_ERR_BAD_ARG;
    const ParamInfo& p = m->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    if (off) *off = (int64_t)p.offset;
    return FCN8S_OK;
}
int fcn8s_param_index(const fcn8s_model* m, const char* name)
{
    if (!m || !name) return -1;
    auto it = m->index.find(name);
    return it == m->index.end() ? -1 : it->second;
}
static int xfer_named(fcn8s_model* m, float* base, const char* name, void* host, size_t n, bool to_dev)
{
    if (!m || !name || !host) return fail(m, FCN8S_ERR_BAD_ARG, "null argument");
    const int i = fcn8s_param_index(m, name);
    if (i < 0) return fail(m, FCN8S_ERR_NOT_FOUND, std::string("unknown variable '") + name + "'");
    const ParamInfo& p = m->params[i];
    if (n != p.numel) return fail(m, FCN8S_ERR_SHAPE, std::string("variable '") + name + "' has " + std::to_string(p.numel) + " elements, got " + std::to_string(n));
    if (base == m->d_grads) { int rcw = fcn8s_comm_wait(m); if (rcw) return rcw; }      // an all-reduce still rewriting the bucket in place: read behind it
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (to_dev) HIPCHK(m, hipMemcpy(base + p.offset, host, n * sizeof(float), hipMemcpyHostToDevice));
    else HIPCHK(m, hipMemcpy(host, base + p.offset, n * sizeof(float), hipMemcpyDeviceToHost));
    return FCN8S_OK;
}
int fcn8s_set_param(fcn8s_model* m, const char* name, const float* host, size_t n)
{
    if (m && m->frozen) fcn8s_freeze_params(m, 0);      // parameters are about to change: leave the frozen state
    return xfer_named(m, m ? m->d_params : nullptr, name, (void*)host, n, true);
}
int fcn8s_get_param(fcn8s_model* m, const char* name, float* host, size_t n) { return xfer_named(m, m ? m->d_params : nullptr, name, host, n, false); }
int fcn8s_get_grad(fcn8s_model* m, const char* name, float* host, size_t n) { return xfer_named(m, m ? m->d_grads : nullptr, name, host, n, false); }
void* fcn8s_param_buffer(fcn8s_model* m, size_t* n) { if (!m) return nullptr; if (n) *n = m->total; return m->d_params; }
void* fcn8s_grad_buffer(fcn8s_model* m, size_t* n) { if (!m) return nullptr; if (n) *n = m->total; return m->d_grads; }
int fcn8s_num_buckets(const fcn8s_model* m) { return m ? kNumBuckets : 0; }
int fcn8s_bucket_range(const fcn8s_model* m, int b, size_t* off, size_t* n)
{
    if (!m || b < 0 || b >= kNumBuckets) return FCN8S_ERR_BAD_ARG;
    if (off) *off = m->bucket_off[b];
    if (n) *n = m->bucket_n[b];
    return FCN8S_OK;
}

int fcn8s_init_params(fcn8s_model* m, uint64_t seed)
{
    if (m && m->frozen) fcn8s_freeze_params(m, 0);      // parameters are about to change (or a training pass starts): leave the frozen state
    if (!m) return FCN8S_ERR_BAD_ARG;
    HIPCHK(m, hipMemsetAsync(m->d_params, 0, m->total * sizeof(float), m->stream));
    uint32_t sid = 1000;
    for (auto& p : m->params) {
        ++sid;
        if (p.ndim == 1) continue;                                   // biases zero
        float std; int trunc = 0;
        const bool vgg = p.name.find("/filter") != std::string::npos || p.name.find("/weights") != std::string::npos;
        if (vgg) std = sqrtf(2.0f / (float)(p.shape[0] * p.shape[1] * p.shape[2]));
        else if (p.name.find("trans") != std::string::npos) { std = 0.01f; trunc = 1; }   // :160
        else { std = 0.001f; trunc = 1; }                                                     // :159
        launch_init_normal(m->d_params + p.offset, (long long)p.numel, std, trunc, seed, sid, m->stream);
    }
    HIPCHK(m, hipGetLastError());
    return FCN8S_OK;
}

int fcn8s_forward_loss(fcn8s_model* m, const void* images, int dtype, const uint8_t* labels, int N, int H, int W,
                       float keep_prob, float l2_rate, int where)
{
    if (m && m->frozen) fcn8s_freeze_params(m, 0);      // parameters are about to change (or a training pass starts): leave the frozen state
    if (!m || !images || !labels) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_forward_loss: null argument");
    if (!(keep_prob > 0.f && keep_prob <= 1.f)) return fail(m, FCN8S_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    int rc = ensure_workspace(m, N, H, W); if (rc) return rc;
    const void* img; const uint8_t* lab;
    rc = stage_inputs(m, images, dtype, labels, where, &img, &lab); if (rc) return rc;
    take_deferred_error(nullptr);                      // (a slot left set by a call that returned early on this thread)
    rc = forward(m, img, dtype, keep_prob, true); if (rc) { take_deferred_error(nullptr); return rc; }
    rc = deferred_rc(m); if (rc) return rc;
    rc = compute_loss(m, lab, l2_rate, true); if (rc) return rc;
    m->next_bucket = 0;
    HIPCHK(m, hipGetLastError());
    return FCN8S_OK;
}

int fcn8s_backward_bucket(fcn8s_model* m, int bucket)
{
    if (!m || bucket < 0 || bucket >= kNumBuckets) return fail(m, FCN8S_ERR_BAD_ARG, "bad bucket");
    take_deferred_error(nullptr);
    int rc = do_backward_bucket(m, bucket, 1); if (rc) { take_deferred_error(nullptr); return rc; }
    return deferred_rc(m);
}

int fcn8s_bucket_wait(fcn8s_model* m, int bucket, void* hip_stream)
{
    if (!m || bucket < 0 || bucket >= kNumBuckets) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_bucket_wait: bad bucket");
    if (!m->bucket_final[bucket] || !m->bucket_ev[bucket])
        return fail(m, FCN8S_ERR_STATE, "fcn8s_bucket_wait: the bucket's gradients are not queued yet (call fcn8s_backward_bucket up to fcn8s_bucket_complete_after(bucket) first)");
    HIPCHK(m, hipStreamWaitEvent((hipStream_t)hip_stream, m->bucket_ev[bucket], 0));
    return FCN8S_OK;
}

int fcn8s_bucket_complete_after(const fcn8s_model* m, int bucket)
{
    if (!m || bucket < 0 || bucket >= kNumBuckets) return -1;
    return bucket_complete_after(m, bucket);
}

// ---- the library's own RCCL communicator ------------------------------------------------------------------------------------------
// librccl is opened at run time (dlopen by soname): a process that already holds PyTorch-ROCm's copy gets that one, a plain C caller gets
// /opt/rocm's, and a single-GPU user of the library never loads the 570 MB of it.
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                                   // (optional: absent in very old builds)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;            // (optional)
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi* rccl()
{
    static RcclApi api; static std::once_flag once;
    std::call_once(once, [] {
        // FCN8S_RCCL_LIBRARY: the RCCL build to use, by path (a site's own build; the shared-memory stand-in of tests/fake_rccl) -- no fallback if it is set
        const char* forced = getenv("FCN8S_RCCL_LIBRARY");
        if (forced && *forced) api.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        else for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (api.h) break; }
        if (!api.h) { const char* e = dlerror(); api.err = std::string("dlopen(") + (forced && *forced ? forced : "librccl.so.1") + "): " + (e ? e : "not found"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.h, n); if (!p && api.err.empty()) api.err = std::string("librccl has no symbol ") + n; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
        api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.CommAbort = (decltype(api.CommAbort))dlsym(api.h, "ncclCommAbort");
        api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.h, "ncclCommGetAsyncError");
    });
    return &api;
}
int rccl_fail(fcn8s_model* m, const char* what, ncclResult_t r)
{
    RcclApi* a = rccl();
    return fail(m, FCN8S_ERR_RCCL, std::string(what) + ": " + ((a->GetErrorString && r != ncclSuccess) ? a->GetErrorString(r) : a->err.c_str()));
}
#define RCCLCHK(m, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return rccl_fail(m, #call, r_); } while (0)

int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The communicator has failed (asynchronous RCCL error, or a collective older than comm_timeout_ms: a dead or hung peer).  ncclCommAbort makes
// RCCL's kernels leave the streams they sit on, so that neither the model's stream nor a host synchronisation waits for that peer forever.
// Caller holds comm_mu.
void comm_abort_locked(fcn8s_model* m, const std::string& why)
{
    if (m->comm_failed.load()) return;
    m->comm_error = why;
    m->comm_failed.store(true);
    RcclApi* a = rccl();
    if (m->comm) {
        if (a->CommAbort) a->CommAbort(m->comm);          // (frees the communicator like ncclCommDestroy, without waiting for its peers)
        m->comm = nullptr;
    }
    for (int b = 0; b < fcn8s_model::kCommSlots; ++b) m->comm_inflight[b].store(false);
    for (int b = 0; b < FCN8S_MAX_BUCKETS; ++b) m->comm_pending[b] = false;      // (the model's stream must not wait for events of a communicator that is gone)
}

// a collective has just been enqueued on `s` (caller holds comm_mu): the watchdog measures its age from now and learns of its completion from the event
int comm_track_locked(fcn8s_model* m, int slot, hipStream_t s)
{
    if (!m->comm_done[slot]) HIPCHK(m, hipEventCreateWithFlags(&m->comm_done[slot], hipEventDisableTiming));
    HIPCHK(m, hipEventRecord(m->comm_done[slot], s));
    m->comm_enq_ns[slot] = now_ns(); m->comm_inflight[slot].store(true);
    return FCN8S_OK;
}

void comm_watchdog(fcn8s_model* m)
{
    hipSetDevice(m->device);
    RcclApi* a = rccl();
    while (!m->comm_watch_stop.load()) {
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        std::lock_guard<std::mutex> lk(m->comm_mu);
        if (!m->comm || m->comm_failed.load()) continue;
        ncclResult_t async = ncclSuccess;
        if (a->CommGetAsyncError && a->CommGetAsyncError(m->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            comm_abort_locked(m, std::string("asynchronous RCCL error: ") + (a->GetErrorString ? a->GetErrorString(async) : "?") + " (communicator aborted)");
            continue;
        }
        const int64_t t = now_ns();
        for (int b = 0; b < fcn8s_model::kCommSlots; ++b) {
            if (!m->comm_inflight[b].load() || !m->comm_done[b]) continue;
            if (hipEventQuery(m->comm_done[b]) == hipSuccess) { m->comm_inflight[b].store(false); continue; }
            (void)hipGetLastError();                      // (hipErrorNotReady is not an error)
            if ((t - m->comm_enq_ns[b]) / 1000000 > m->comm_timeout_ms) {
                comm_abort_locked(m, (b == fcn8s_model::kCommMisc ? std::string("the parameter broadcast / metrics all-reduce") : "all-reduce of gradient bucket " + std::to_string(b)) +
                                     " did not complete within " + std::to_string(m->comm_timeout_ms) +
                                     " ms (option comm_timeout_ms): a peer rank is dead or hung; communicator aborted");
                break;
            }
        }
    }
}

// FCN8S_ERR_RCCL with the recorded reason once the communicator has failed
int comm_failed_rc(fcn8s_model* m, const char* where)
{
    std::lock_guard<std::mutex> lk(m->comm_mu);
    return fail(m, FCN8S_ERR_RCCL, std::string(where) + ": " + m->comm_error);
}
}  // namespace

int fcn8s_device_pci_bus_id(int device_id, char* out, size_t len)
{
    if (!out || len < 16) return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_device_pci_bus_id: need a buffer of at least 16 bytes");
    hipError_t e = hipDeviceGetPCIBusId(out, (int)len, device_id);
    if (e != hipSuccess) return fail(nullptr, FCN8S_ERR_HIP, std::string("hipDeviceGetPCIBusId: ") + hipGetErrorString(e));
    return FCN8S_OK;
}

int fcn8s_comm_unique_id(void* id_out, size_t nbytes)
{
    if (!id_out || nbytes < sizeof(ncclUniqueId)) return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_comm_unique_id: need a buffer of FCN8S_COMM_ID_BYTES bytes");
    RcclApi* a = rccl();
    if (!a->err.empty()) return rccl_fail(nullptr, "fcn8s_comm_unique_id", ncclSuccess);
    ncclUniqueId id;
    RCCLCHK(nullptr, a->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return FCN8S_OK;
}

int fcn8s_comm_init(fcn8s_model* m, const void* unique_id, size_t nbytes, int rank, int world)
{
    if (!m || !unique_id || nbytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_comm_init: bad argument");
    if (m->comm) return fail(m, FCN8S_ERR_STATE, "fcn8s_comm_init: this model already has a communicator (fcn8s_comm_destroy first)");
    RcclApi* a = rccl();
    if (!a->err.empty()) return rccl_fail(m, "fcn8s_comm_init", ncclSuccess);
    HIPCHK(m, hipSetDevice(m->device));
    ncclUniqueId id; memcpy(&id, unique_id, sizeof id);
    if (m->comm_watch.joinable()) { m->comm_watch_stop.store(true); m->comm_watch.join(); }       // (left over from a communicator that failed)
    // the enum values this file declares for itself (ncclFloat = 7, ncclDouble = 8, ncclSum = 0, the result codes) are those of NCCL / RCCL 2.x
    { int v = 0; RCCLCHK(m, a->GetVersion(&v));
      const int major = v >= 10000 ? v / 10000 : v / 1000;       // (NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch since 2.9, major * 1000 + ... before)
      if (major != 2) return fail(m, FCN8S_ERR_RCCL, "fcn8s_comm_init: librccl reports version code " + std::to_string(v) + "; this library speaks the RCCL 2.x ABI only"); }
    RCCLCHK(m, a->CommInitRank(&m->comm, world, id, rank));
    m->comm_rank = rank; m->comm_world = world;
    m->comm_failed.store(false); m->comm_error.clear();
    for (int b = 0; b < fcn8s_model::kCommSlots; ++b) m->comm_inflight[b].store(false);
    for (int b = 0; b < FCN8S_MAX_BUCKETS; ++b) m->comm_pending[b] = false;
    if (!m->comm_stream) HIPCHK(m, hipStreamCreateWithFlags(&m->comm_stream, hipStreamNonBlocking));
    if (world > 1) { m->comm_watch_stop.store(false); m->comm_watch = std::thread(comm_watchdog, m); }      // (a one-rank communicator has no peer to lose)
    return FCN8S_OK;
}

int fcn8s_comm_destroy(fcn8s_model* m)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    int rc = FCN8S_OK;
    if (m->comm) {
        // drain the communicator's stream -- but never wait for a peer that is gone: poll, with the watchdog's rules (asynchronous error or
        // comm_timeout_ms without progress -> ncclCommAbort).  A one-rank communicator, or RCCL without the two entry points, just synchronises.
        RcclApi* a = rccl();
        if (m->comm_stream) {
            const int64_t t0 = now_ns();
            while (hipStreamQuery(m->comm_stream) == hipErrorNotReady) {
                if (m->comm_failed.load()) break;
                if (m->comm_world > 1 && (now_ns() - t0) / 1000000 > m->comm_timeout_ms) {
                    std::lock_guard<std::mutex> lk(m->comm_mu);
                    comm_abort_locked(m, "fcn8s_comm_destroy: collectives still in flight after " + std::to_string(m->comm_timeout_ms) + " ms; communicator aborted");
                    break;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            (void)hipGetLastError();
        }
        if (m->comm_watch.joinable()) { m->comm_watch_stop.store(true); m->comm_watch.join(); }
        std::lock_guard<std::mutex> lk(m->comm_mu);
        if (m->comm) { a->CommDestroy(m->comm); m->comm = nullptr; }
    }
    if (m->comm_watch.joinable()) { m->comm_watch_stop.store(true); m->comm_watch.join(); }
    if (m->comm_failed.load()) { rc = fail(m, FCN8S_ERR_RCCL, "fcn8s_comm_destroy: " + m->comm_error); m->comm_failed.store(false); }
    for (int b = 0; b < fcn8s_model::kCommSlots; ++b) { if (m->comm_done[b]) { hipEventDestroy(m->comm_done[b]); m->comm_done[b] = nullptr; } m->comm_inflight[b].store(false); }
    for (int b = 0; b < FCN8S_MAX_BUCKETS; ++b) m->comm_pending[b] = false;
    if (m->comm_stream) { hipStreamDestroy(m->comm_stream); m->comm_stream = nullptr; }
    m->comm_rank = 0; m->comm_world = 1;
    return rc;
}

int fcn8s_comm_info(const fcn8s_model* m, int* rank, int* world, int* rccl_version)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (rank) *rank = m->comm_rank;
    if (world) *world = (m->comm || m->comm_failed.load()) ? m->comm_world : 0;
    if (rccl_version) { *rccl_version = 0; RcclApi* a = rccl(); if (a->GetVersion) a->GetVersion(rccl_version); }
    return FCN8S_OK;
}

int fcn8s_allreduce_bucket(fcn8s_model* m, int bucket)
{
    if (!m || bucket < 0 || bucket >= kNumBuckets) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_allreduce_bucket: bad bucket");
    if (m->comm_failed.load()) return comm_failed_rc(m, "fcn8s_allreduce_bucket");
    if (!m->comm) return fail(m, FCN8S_ERR_STATE, "fcn8s_allreduce_bucket: no communicator (fcn8s_comm_init first)");
    if (!m->bucket_final[bucket]) return fail(m, FCN8S_ERR_STATE, "fcn8s_allreduce_bucket: the bucket's gradients are not queued yet (fcn8s_bucket_complete_after)");
    if (m->comm_pending[bucket]) return fail(m, FCN8S_ERR_STATE, "fcn8s_allreduce_bucket: this bucket is already being reduced");
    HIPCHK(m, hipStreamWaitEvent(m->comm_stream, m->bucket_ev[bucket], 0));
    float* g = m->d_grads + m->bucket_off[bucket];
    std::lock_guard<std::mutex> lk(m->comm_mu);          // (the watchdog may abort the communicator: not in the middle of an enqueue)
    if (!m->comm) return fail(m, FCN8S_ERR_RCCL, "fcn8s_allreduce_bucket: " + m->comm_error);
    RCCLCHK(m, rccl()->AllReduce(g, g, m->bucket_n[bucket], ncclFloat, ncclSum, m->comm, m->comm_stream));
    { int rc = comm_track_locked(m, bucket, m->comm_stream); if (rc) return rc; }
    m->comm_pending[bucket] = true;
    return FCN8S_OK;
}

int fcn8s_comm_wait(fcn8s_model* m)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (m->comm_failed.load()) return comm_failed_rc(m, "fcn8s_comm_wait");
    for (int b = 0; b < kNumBuckets; ++b)
        if (m->comm_pending[b]) { HIPCHK(m, hipStreamWaitEvent(m->stream, m->comm_done[b], 0)); m->comm_pending[b] = false; }
    return FCN8S_OK;
}

namespace {
// The HOST waits until every pending all-reduce has completed -- polling, so that the watchdog's abort ends the wait -- and reports a failed communicator.
// fcn8s_apply_update calls it when the communicator has more than one rank: an update must never be applied to gradients whose exchange was aborted (the
// stream-ordered fcn8s_comm_wait alone would let the update kernel run behind an aborted all-reduce and report the failure one call too late).  What it
// costs is the host's run-ahead over the update kernel: a few microseconds of launch latency per step.
int comm_wait_host(fcn8s_model* m, const char* where)
{
    for (int b = 0; b < kNumBuckets; ++b) {
        if (!m->comm_pending[b]) continue;
        while (!m->comm_failed.load() && m->comm_done[b] && hipEventQuery(m->comm_done[b]) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(50));
        (void)hipGetLastError();
    }
    if (m->comm_failed.load()) return comm_failed_rc(m, where);
    return FCN8S_OK;
}
}  // namespace

int fcn8s_comm_broadcast_params(fcn8s_model* m, int root)
{
    if (!m || root < 0) return FCN8S_ERR_BAD_ARG;
    if (m->comm_failed.load()) return comm_failed_rc(m, "fcn8s_comm_broadcast_params");
    if (!m->comm) return fail(m, FCN8S_ERR_STATE, "fcn8s_comm_broadcast_params: no communicator (fcn8s_comm_init first)");
    if (root >= m->comm_world) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_comm_broadcast_params: root outside the communicator");
    if (m->frozen) fcn8s_freeze_params(m, 0);
    std::lock_guard<std::mutex> lk(m->comm_mu);          // (the watchdog may abort the communicator: not in the middle of an enqueue)
    if (!m->comm) return fail(m, FCN8S_ERR_RCCL, "fcn8s_comm_broadcast_params: " + m->comm_error);
    RCCLCHK(m, rccl()->Broadcast(m->d_params, m->d_params, m->total, ncclFloat, root, m->comm, m->stream));
    return comm_track_locked(m, fcn8s_model::kCommMisc, m->stream);      // stream-ordered like every parameter write; a root that never sends trips the watchdog
}

int fcn8s_comm_allreduce_metrics(fcn8s_model* m)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (m->comm_failed.load()) return comm_failed_rc(m, "fcn8s_comm_allreduce_metrics");
    if (!m->comm) return fail(m, FCN8S_ERR_STATE, "fcn8s_comm_allreduce_metrics: no communicator (fcn8s_comm_init first)");
    // confusion counts (exact as doubles below 2^53), the sum of the per-batch losses and their count: one SUM all-reduce
    const size_t cc = (size_t)m->C * m->C, n = cc + 2;
    std::vector<int64_t> conf(cc); double ls; int64_t lc;
    int rc = fcn8s_metrics_raw(m, conf.data(), &ls, &lc); if (rc) return rc;
    std::vector<double> h(n);
    for (size_t i = 0; i < cc; ++i) h[i] = (double)conf[i];
    h[cc] = ls; h[cc + 1] = (double)lc;
    double* d = nullptr;
    HIPCHK(m, hipMalloc((void**)&d, n * sizeof(double)));
    hipMemcpyAsync(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice, m->stream);
    ncclResult_t r = ncclSuccess; int rct = FCN8S_OK; bool gone = false;
    {
        std::lock_guard<std::mutex> lk(m->comm_mu);      // (the watchdog may abort the communicator: not in the middle of an enqueue)
        if (!m->comm) gone = true;
        else {
            r = rccl()->AllReduce(d, d, n, ncclDouble, ncclSum, m->comm, m->stream);
            if (r == ncclSuccess) rct = comm_track_locked(m, fcn8s_model::kCommMisc, m->stream);
        }
    }
    // the host needs the sums: wait for them by polling, so that a peer that never arrives ends in the watchdog's abort (which releases the stream), not in a hang
    if (!gone && r == ncclSuccess && rct == FCN8S_OK) {
        while (hipEventQuery(m->comm_done[fcn8s_model::kCommMisc]) == hipErrorNotReady && !m->comm_failed.load()) std::this_thread::sleep_for(std::chrono::microseconds(100));
        (void)hipGetLastError();
    }
    if (!gone && r == ncclSuccess && rct == FCN8S_OK && !m->comm_failed.load()) hipMemcpyAsync(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, m->stream);
    hipStreamSynchronize(m->stream);                     // (behind a completed or aborted collective: returns)
    hipFree(d);
    if (gone || m->comm_failed.load()) return comm_failed_rc(m, "fcn8s_comm_allreduce_metrics");
    if (r != ncclSuccess) return rccl_fail(m, "ncclAllReduce(metrics)", r);
    if (rct) return rct;
    for (size_t i = 0; i < cc; ++i) conf[i] = (int64_t)llround(h[i]);
    return fcn8s_metrics_set_raw(m, conf.data(), h[cc], (int64_t)llround(h[cc + 1]));
}

int fcn8s_apply_update(fcn8s_model* m, int optimizer, float lr, float grad_scale)
{
    if (m && m->frozen) fcn8s_freeze_params(m, 0);      // parameters are about to change (or a training pass starts): leave the frozen state
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (m->comm_world > 1 && (m->comm || m->comm_failed.load())) { int rch = comm_wait_host(m, "fcn8s_apply_update"); if (rch) return rch; }
    { int rcw = fcn8s_comm_wait(m); if (rcw) return rcw; }      // gradient buckets still being all-reduced by the library's own communicator
    const int64_t t = m->step + 1;
    if (optimizer == FCN8S_OPT_TF_ADAM) {
        int rc = ensure_opt_state(m); if (rc) return rc;
        const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
        const float lr_t = lr * (float)std::sqrt(1.0 - std::pow((double)b2, (double)t)) / (float)(1.0 - std::pow((double)b1, (double)t));
        ProfScope ps(m, "adam", 0, 28.0 * m->total);
        launch_tf_adam(m->d_params, m->d_grads, m->d_m, m->d_v, (long long)m->total, lr_t, b1, b2, eps, grad_scale, m->stream);
    } else if (optimizer == FCN8S_OPT_SGD_MOMENTUM) {
        int rc = ensure_opt_state(m); if (rc) return rc;
        ProfScope ps(m, "sgd_momentum", 0, 20.0 * m->total);
        launch_sgd_momentum(m->d_params, m->d_grads, m->d_m, (long long)m->total, lr, 0.9f, grad_scale, m->stream);
    } else if (optimizer != FCN8S_OPT_NONE) return fail(m, FCN8S_ERR_BAD_ARG, "unknown optimizer");
    m->step = t;
    HIPCHK(m, hipGetLastError());
    return FCN8S_OK;
}

int fcn8s_read_loss(fcn8s_model* m, float* loss_out)
{
    if (!m || !loss_out) return FCN8S_ERR_BAD_ARG;
    if (m->loss_copied) {
        HIPCHK(m, hipEventSynchronize(m->loss_ev));
        *loss_out = *m->h_loss;
        return FCN8S_OK;
    }
    HIPCHK(m, hipMemcpyAsync(loss_out, m->d_loss, sizeof(float), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    return FCN8S_OK;
}

int fcn8s_train_step(fcn8s_model* m, const void* images, int dtype, const uint8_t* labels, int N, int H, int W,
                     float lr, float keep_prob, float l2_rate, int where, float* loss_out, int64_t* step_out)
{
    if (m && m->frozen) fcn8s_freeze_params(m, 0);      // parameters are about to change (or a training pass starts): leave the frozen state
    int rc = fcn8s_forward_loss(m, images, dtype, labels, N, H, W, keep_prob, l2_rate, where); if (rc) return rc;
    // a model with a communicator of more than one rank trains data-parallel through this entry point too: the bucket-by-bucket
    // backward pass, every bucket all-reduced as soon as it is final, 1/world in the update (a C caller must never get silently diverging replicas)
    const bool dp = (m->comm || m->comm_failed.load()) && m->comm_world > 1;
    for (int b = 0; b < kNumBuckets; ++b) {
        take_deferred_error(nullptr);
        rc = do_backward_bucket(m, b, dp ? 1 : 2); if (rc) { take_deferred_error(nullptr); return rc; }
        rc = deferred_rc(m); if (rc) return rc;
        if (dp) for (int r = 0; r < kNumBuckets; ++r)
            if (bucket_complete_after(m, r) == b) { rc = fcn8s_allreduce_bucket(m, r); if (rc) return rc; }
    }
    rc = fcn8s_apply_update(m, FCN8S_OPT_TF_ADAM, lr, dp ? 1.f / (float)m->comm_world : 1.f); if (rc) return rc;
    if (loss_out) { rc = fcn8s_read_loss(m, loss_out); if (rc) return rc; }
    if (step_out) *step_out = m->step;
    return FCN8S_OK;
}

int fcn8s_eval_step(fcn8s_model* m, const void* images, int dtype, const uint8_t* labels, int N, int H, int W,
                    float l2_rate, int where)
{
    if (!m || !images || !labels) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_eval_step: null argument");
    int rc = ensure_workspace(m, N, H, W); if (rc) return rc;
    const void* img; const uint8_t* lab;
    rc = stage_inputs(m, images, dtype, labels, where, &img, &lab); if (rc) return rc;
    take_deferred_error(nullptr);
    rc = forward(m, img, dtype, 1.0f, false); if (rc) { take_deferred_error(nullptr); return rc; }
    rc = deferred_rc(m); if (rc) return rc;
    rc = compute_loss(m, lab, l2_rate, false); if (rc) return rc;
    const long long npix = (long long)N * H * W;
    { ProfScope ps(m, "softmax_argmax", 0, (double)npix * (m->C * 4 + 8)); launch_softmax_argmax(LG(m), nullptr, m->d_pred, npix, m->C, m->stream, LGM(m), m->N); }
    launch_confusion(lab, m->d_pred, npix, m->d_conf, m->C, m->stream);
    float loss = 0.f;
    rc = fcn8s_read_loss(m, &loss); if (rc) return rc;          // tf.metrics.mean(total_loss): one sample per batch
    m->loss_sum += (double)loss; m->loss_cnt += 1;
    return FCN8S_OK;
}

int fcn8s_metrics_reset(fcn8s_model* m)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    HIPCHK(m, hipMemsetAsync(m->d_conf, 0, (size_t)m->C * m->C * sizeof(unsigned long long), m->stream));
    m->loss_sum = 0; m->loss_cnt = 0;
    return FCN8S_OK;
}

int fcn8s_metrics_raw(fcn8s_model* m, int64_t* conf, double* loss_sum, int64_t* loss_cnt)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    if (conf) {
        HIPCHK(m, hipMemcpyAsync(conf, m->d_conf, (size_t)m->C * m->C * sizeof(int64_t), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
    }
    if (loss_sum) *loss_sum = m->loss_sum;
    if (loss_cnt) *loss_cnt = m->loss_cnt;
    return FCN8S_OK;
}

int fcn8s_metrics_set_raw(fcn8s_model* m, const int64_t* conf, double loss_sum, int64_t loss_cnt)
{
    if (!m || !conf) return FCN8S_ERR_BAD_ARG;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(m->d_conf, conf, (size_t)m->C * m->C * sizeof(int64_t), hipMemcpyHostToDevice));
    m->loss_sum = loss_sum; m->loss_cnt = loss_cnt;
    return FCN8S_OK;
}

int fcn8s_metrics_get(fcn8s_model* m, double* mean_loss, double* mean_iou, double* accuracy)
{
    return fcn8s_metrics_get_ex(m, mean_loss, mean_iou, accuracy, 0);
}

int fcn8s_metrics_get_ex(fcn8s_model* m, double* mean_loss, double* mean_iou, double* accuracy, int all_classes)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    const int C = m->C;
    std::vector<int64_t> cm((size_t)C * C);
    int rc = fcn8s_metrics_raw(m, cm.data(), nullptr, nullptr); if (rc) return rc;
    // tf.metrics.mean_iou: iou_c = diag/(row+col-diag) with zero denominators replaced by 1; mean over the classes with a non-zero
    // denominator (later TF 1.x), or -- all_classes, the TF 1.3.0 of fcn8s_tutorial.ipynb:311 -- over all C classes (an absent class counts as IoU 0)
    double iou_sum = 0, tot = 0, diag = 0; int valid = 0;
    for (int c = 0; c < C; ++c) {
        double row = 0, col = 0;
        for (int k = 0; k < C; ++k) { row += (double)cm[(size_t)c * C + k]; col += (double)cm[(size_t)k * C + c]; }
        const double d = (double)cm[(size_t)c * C + c];
        const double den = row + col - d;
        if (den > 0) { iou_sum += d / den; ++valid; }
        tot += row; diag += d;
    }
    if (mean_loss) *mean_loss = m->loss_cnt > 0 ? m->loss_sum / (double)m->loss_cnt : 0.0;
    if (mean_iou) *mean_iou = all_classes ? iou_sum / C : (valid > 0 ? iou_sum / valid : 0.0);
    if (accuracy) *accuracy = tot > 0 ? diag / tot : 0.0;
    return FCN8S_OK;
}

int fcn8s_predict(fcn8s_model* m, const void* images, int dtype, int N, int H, int W, int argmax, void* out, int where)
{
    if (!m || !images || !out) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_predict: null argument");
    int rc = ensure_workspace(m, N, H, W); if (rc) return rc;
    const void* img; const uint8_t* lab;
    rc = stage_inputs(m, images, dtype, nullptr, where, &img, &lab); if (rc) return rc;
    take_deferred_error(nullptr);
    rc = forward(m, img, dtype, 1.0f, false); if (rc) { take_deferred_error(nullptr); return rc; }
    rc = deferred_rc(m); if (rc) return rc;
    const long long npix = (long long)N * H * W;
    if (where == FCN8S_DEVICE) {
        ProfScope ps(m, "softmax_argmax", 0, (double)npix * (m->C * 4 + 8));
        if (argmax) launch_softmax_argmax(LG(m), nullptr, (long long*)out, npix, m->C, m->stream, LGM(m), m->N);
        else launch_softmax_argmax(LG(m), (float*)out, nullptr, npix, m->C, m->stream, LGM(m), m->N);
        HIPCHK(m, hipGetLastError());
        return FCN8S_OK;
    }
    if (argmax) {
        launch_softmax_argmax(LG(m), nullptr, m->d_pred, npix, m->C, m->stream, LGM(m), m->N);
        HIPCHK(m, hipMemcpyAsync(out, m->d_pred, npix * sizeof(long long), hipMemcpyDeviceToHost, m->stream));
    } else {
        launch_softmax_argmax(LG(m), m->d_softmax, nullptr, npix, m->C, m->stream, LGM(m), m->N);
        HIPCHK(m, hipMemcpyAsync(out, m->d_softmax, npix * m->C * sizeof(float), hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(m, hipStreamSynchronize(m->stream));
    return FCN8S_OK;
}

// ---- asynchronous host boundary ------------------------------------------------------------------
int fcn8s_stage_inputs(fcn8s_model* m, int slot, const void* images, int dtype, const uint8_t* label_ids, int N, int H, int W,
                       void** images_dev, uint8_t** labels_dev)
{
    if (!m || !images || slot < 0 || slot >= FCN8S_NUM_STAGE_SLOTS || N <= 0 || H <= 0 || W <= 0) return fail(m, FCN8S_ERR_BAD_ARG, "fcn8s_stage_inputs: bad argument");
    HIPCHK(m, hipSetDevice(m->device));                  // may be called from a feeder thread
    if (!m->copy_stream) HIPCHK(m, hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    fcn8s_model::StageSlot& sl = m->slots[slot];
    const size_t npix = (size_t)N * H * W, ib = npix * 3 * (dtype == FCN8S_IMG_U8 ? 1 : 4), lb = label_ids ? npix : 0;
    if (!sl.ready) { HIPCHK(m, hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming)); HIPCHK(m, hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming)); }
    if (sl.used) HIPCHK(m, hipEventSynchronize(sl.ready));        // the previous copy out of this slot's pinned buffers has finished
    if (sl.cap_img < ib) {
        if (sl.used) HIPCHK(m, hipEventSynchronize(sl.consumed));
        if (sl.h_img) hipHostFree(sl.h_img);
        if (sl.d_img) hipFree(sl.d_img);
        sl.h_img = nullptr; sl.d_img = nullptr; sl.cap_img = 0;
        HIPCHK(m, hipHostMalloc(&sl.h_img, ib, hipHostMallocDefault));
        HIPCHK(m, hipMalloc(&sl.d_img, ib));
        sl.cap_img = ib;
    }
    if (lb && sl.cap_lab < lb) {
        if (sl.used) HIPCHK(m, hipEventSynchronize(sl.consumed));
        if (sl.h_lab) hipHostFree(sl.h_lab);
        if (sl.d_lab) hipFree(sl.d_lab);
        sl.h_lab = nullptr; sl.d_lab = nullptr; sl.cap_lab = 0;
        HIPCHK(m, hipHostMalloc((void**)&sl.h_lab, lb, hipHostMallocDefault));
        HIPCHK(m, hipMalloc((void**)&sl.d_lab, lb));
        sl.cap_lab = lb;
    }
    memcpy(sl.h_img, images, ib);                         // pageable -> pinned on this (feeder) thread
    if (lb) memcpy(sl.h_lab, label_ids, lb);
    if (sl.used) HIPCHK(m, hipStreamWaitEvent(m->copy_stream, sl.consumed, 0));   // the step that last read this slot's device buffers
    HIPCHK(m, hipMemcpyAsync(sl.d_img, sl.h_img, ib, hipMemcpyHostToDevice, m->copy_stream));
    if (lb) HIPCHK(m, hipMemcpyAsync(sl.d_lab, sl.h_lab, lb, hipMemcpyHostToDevice, m->copy_stream));
    HIPCHK(m, hipEventRecord(sl.ready, m->copy_stream));
    sl.used = true;
    if (images_dev) *images_dev = sl.d_img;
    if (labels_dev) *labels_dev = lb ? sl.d_lab : nullptr;
    return FCN8S_OK;
}
int fcn8s_stage_wait(fcn8s_model* m, int slot)
{
    if (!m || slot < 0 || slot >= FCN8S_NUM_STAGE_SLOTS || !m->slots[slot].used) return fail(m, FCN8S_ERR_STATE, "fcn8s_stage_wait: nothing staged in this slot");
    HIPCHK(m, hipStreamWaitEvent(m->stream, m->slots[slot].ready, 0));
    return FCN8S_OK;
}
int fcn8s_stage_release(fcn8s_model* m, int slot)
{
    if (!m || slot < 0 || slot >= FCN8S_NUM_STAGE_SLOTS || !m->slots[slot].used) return fail(m, FCN8S_ERR_STATE, "fcn8s_stage_release: nothing staged in this slot");
    HIPCHK(m, hipEventRecord(m->slots[slot].consumed, m->stream));
    return FCN8S_OK;
}

int64_t fcn8s_global_step(const fcn8s_model* m) { return m ? m->step : -1; }
int fcn8s_set_global_step(fcn8s_model* m, int64_t s) { if (!m || s < 0) return FCN8S_ERR_BAD_ARG; m->step = s; return FCN8S_OK; }

int fcn8s_get_opt_state(fcn8s_model* m, float* hm, float* hv, size_t n)
{
    if (!m || n != m->total) return fail(m, FCN8S_ERR_BAD_ARG, "optimizer state size mismatch");
    int rc = ensure_opt_state(m); if (rc) return rc;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (hm) HIPCHK(m, hipMemcpy(hm, m->d_m, n * sizeof(float), hipMemcpyDeviceToHost));
    if (hv) HIPCHK(m, hipMemcpy(hv, m->d_v, n * sizeof(float), hipMemcpyDeviceToHost));
    return FCN8S_OK;
}
int fcn8s_set_opt_state(fcn8s_model* m, const float* hm, const float* hv, size_t n)
{
    if (!m || n != m->total) return fail(m, FCN8S_ERR_BAD_ARG, "optimizer state size mismatch");
    int rc = ensure_opt_state(m); if (rc) return rc;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (hm) HIPCHK(m, hipMemcpy(m->d_m, hm, n * sizeof(float), hipMemcpyHostToDevice));
    if (hv) HIPCHK(m, hipMemcpy(m->d_v, hv, n * sizeof(float), hipMemcpyHostToDevice));
    return FCN8S_OK;
}

int fcn8s_get_activation(fcn8s_model* m, const char* name, float* host, size_t n)
{
    if (!m || !name || !host) return FCN8S_ERR_BAD_ARG;
    if (!m->have_forward) return fail(m, FCN8S_ERR_STATE, "no forward pass has been run");
    if (strncmp(name, "dy:", 3) == 0) {      // a layer's output gradient as the last backward pass handed it to the weight gradient (option "keep_output_gradients")
        auto k = m->kept_dy.find(name + 3);
        if (k == m->kept_dy.end() || !k->second.p || !k->second.n)
            return fail(m, FCN8S_ERR_STATE, std::string("no fp32 output gradient of '") + (name + 3) + "' was kept: set option \"keep_output_gradients\" before the backward pass (bf16_train: with \"bf16_acts\" = 0; "
                                            "a gradient handed on as routing bytes, as a Winograd-domain image or as a bf16 copy only has no fp32 tensor)");
        if (n != k->second.n) return fail(m, FCN8S_ERR_SHAPE, std::string("output gradient of '") + (name + 3) + "' has " + std::to_string(k->second.n) + " elements");
        HIPCHK(m, hipDeviceSynchronize());
        HIPCHK(m, hipMemcpy(host, k->second.p, n * sizeof(float), hipMemcpyDeviceToHost));
        return FCN8S_OK;
    }
    auto it = m->acts.find(name);
    if (it == m->acts.end()) return fail(m, FCN8S_ERR_NOT_FOUND, std::string("unknown activation '") + name + "'");
    if (std::string(name) == "logits" && !m->logits_nhwc_valid && m->logits_b) {      // kept in the blocked GEMM layout: convert for the caller
        launch_unblock_logits(m->logits_b, it->second.p, m->pm, m->N, m->C, m->stream);
        m->logits_nhwc_valid = true;
    }
    if (n != it->second.n) return fail(m, FCN8S_ERR_SHAPE, std::string("activation '") + name + "' has " + std::to_string(it->second.n) + " elements");
    if (m->y_unwritten.count(name))
        return fail(m, FCN8S_ERR_STATE, std::string("activation '") + name + "' was not materialised by the last forward pass: " +
                                        (bf16_train_mode(m) ? "bf16_train keeps a conv -> conv activation only as the consumer's padded bf16 copy (option \"bf16_acts\" = 0 keeps the fp32 tensor too)" :
                                         std::string(name) == "conv1_1" ? "conv1_2's input transform evaluated it on its own patches (option \"conv1_in_transform\" = 0 keeps it)"
                                                                        : "its output transform wrote the next conv's transformed input directly (option \"fuse_out_in\" = 0 keeps it)"));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(host, it->second.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return FCN8S_OK;
}

int fcn8s_get_relu_record(fcn8s_model* m, const char* layer, unsigned char* host, size_t n)
{
    if (!m || !layer || !host) return FCN8S_ERR_BAD_ARG;
    if (!m->have_forward || !m->train_mode) return fail(m, FCN8S_ERR_STATE, "fcn8s_get_relu_record: no training forward pass has been run");
    int b = 0, i = 0;
    if (sscanf(layer, "conv%d_%d", &b, &i) != 2 || b < 1 || b > 5 || i < 1 || i >= kConvsPerBlock[b - 1])
        return fail(m, FCN8S_ERR_NOT_FOUND, std::string("fcn8s_get_relu_record: '") + layer + "' is not a conv that feeds another conv");
    auto it = m->acts.find(std::string("rb:") + layer);
    if (it == m->acts.end() || !m->rbits_ok.count(layer))
        return fail(m, FCN8S_ERR_STATE, std::string("fcn8s_get_relu_record: the last forward pass kept no ReLU record of '") + layer + "' (the backward pass reads the activation itself)");
    const int H = m->H >> (b - 1), W = m->W >> (b - 1), Cc = m->widths[b - 1], N = m->N;
    if (n != (size_t)N * H * W * Cc) return fail(m, FCN8S_ERR_SHAPE, std::string("ReLU record of '") + layer + "' has " + std::to_string((size_t)N * H * W * Cc) + " elements");
    // layout of wino_output_kernel / wino_input_kernel / wino_out_in_kernel: RW words per (tile, channel vector), bit (oy * M + ox) * VEC + lane-channel
    const int tile = wino_tile_for(m, H, W, 3), vec = tile == 2 ? 4 : 2, RW = (tile * tile * vec + 31) / 32, C4 = Cc / vec;
    const int th = (H + tile - 1) / tile, tw = (W + tile - 1) / tile;
    const size_t words = wino_rbits_words(tile, N, H, W, Cc);
    std::vector<unsigned> w(words);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(w.data(), it->second.p, words * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (int nn = 0; nn < N; ++nn)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t t = ((size_t)nn * th + y / tile) * tw + x / tile;
                const int oy = y % tile, ox = x % tile;
                unsigned char* dst = host + (((size_t)nn * H + y) * W + x) * Cc;
                for (int c = 0; c < Cc; ++c) {
                    const int bit = (oy * tile + ox) * vec + c % vec;
                    dst[c] = (w[(t * C4 + c / vec) * RW + (bit >> 5)] >> (bit & 31)) & 1u;
                }
            }
    return FCN8S_OK;
}

int fcn8s_get_pool_routing(fcn8s_model* m, int block, unsigned char* host, size_t n)
{
    if (!m || !host || block < 1 || block > 5) return FCN8S_ERR_BAD_ARG;
    if (!m->have_forward || !m->train_mode) return fail(m, FCN8S_ERR_STATE, "fcn8s_get_pool_routing: no training forward pass has been run");
    const int h = m->H >> (block - 1), w = m->W >> (block - 1), cw = m->widths[block - 1];
    const size_t want = (size_t)m->N * (h / 2) * (w / 2) * cw;
    if (n != want) return fail(m, FCN8S_ERR_SHAPE, "pool routing of block " + std::to_string(block) + " has " + std::to_string(want) + " bytes");
    char ix[16]; snprintf(ix, sizeof ix, "pidx%d", block);
    unsigned char* d = (unsigned char*)A(m, ix);
    if (!pool_backward_fused(m, block, m->pool_fused[block - 1]) && !m->pool_routed[block - 1]) {      // (pool_routed: the forward pool of bf16_train kept the bytes)
        // the backward pass routes through maxpool_bwd_kernel on the block's last conv output (materialised in this case): same rule
        char last[32]; snprintf(last, sizeof last, "conv%d_%d", block, kConvsPerBlock[block - 1]);
        launch_maxpool_route(A(m, last), d, m->N, h, w, cw, m->stream);
    }
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(host, d, n, hipMemcpyDeviceToHost));
    return FCN8S_OK;
}

int fcn8s_get_dropout_masks(fcn8s_model* m, float* h6, size_t n6, float* h7, size_t n7)
{
    if (!m || !m->have_forward) return fail(m, FCN8S_ERR_STATE, "no forward pass has been run");
    const Act& a6 = m->acts.at("fc6"); const Act& a7 = m->acts.at("fc7");
    if (n6 != a6.n || n7 != a7.n) return fail(m, FCN8S_ERR_SHAPE, "mask size mismatch");
    const float keep = (m->train_mode ? m->keep_prob : 1.f);
    float* d = nullptr;
    const size_t nmax = n6 > n7 ? n6 : n7;
    HIPCHK(m, hipMalloc((void**)&d, nmax * sizeof(float)));
    launch_dropout_mask(d, (long long)n6, keep, m->seed, m->drop_stream, m->stream);
    hipMemcpyAsync(h6, d, n6 * sizeof(float), hipMemcpyDeviceToHost, m->stream);
    hipStreamSynchronize(m->stream);
    launch_dropout_mask(d, (long long)n7, keep, m->seed, m->drop_stream + 1, m->stream);
    hipMemcpyAsync(h7, d, n7 * sizeof(float), hipMemcpyDeviceToHost, m->stream);
    hipStreamSynchronize(m->stream);
    hipFree(d);
    return FCN8S_OK;
}

int fcn8s_profile_enable(fcn8s_model* m, int on) { if (!m) return FCN8S_ERR_BAD_ARG; m->profile = on != 0; m->profile_detail = on == 2; return FCN8S_OK; }
int fcn8s_profile_reset(fcn8s_model* m)
{
    if (!m) return FCN8S_ERR_BAD_ARG;
    hipStreamSynchronize(m->stream);
    for (auto& g : m->groups) { for (auto& ev : g.ev) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); } g.ev.clear(); g.ev_shared.clear(); g.flops = g.bytes = 0; g.launches = 0; }
    return FCN8S_OK;
}
int fcn8s_profile_num_groups(const fcn8s_model* m) { return m ? (int)m->groups.size() : 0; }
int fcn8s_profile_get(fcn8s_model* m, int gi, const char** name, double* total_ms, int64_t* launches, double* flops, double* bytes)
{
    if (!m || gi < 0 || gi >= (int)m->groups.size()) return FCN8S_ERR_BAD_ARG;
    ProfGroup& g = m->groups[gi];
    HIPCHK(m, hipStreamSynchronize(m->stream));
    double ms = 0;
    for (auto& ev : g.ev) { float t = 0; if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) ms += t; }
    for (auto& ev : g.ev_shared) { float t = 0; if (hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) ms += t; }
    if (name) *name = g.name.c_str();
    if (total_ms) *total_ms = ms;
    if (launches) *launches = g.launches;
    if (flops) *flops = g.flops;
    if (bytes) *bytes = g.bytes;
    return FCN8S_OK;
}

// ---- single ops ------------------------------------------------------------------------------
#define OPCHK() do { { std::string t_; const int c_ = take_deferred_error(&t_); if (c_) return fail(nullptr, c_, t_); } \
                     hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(nullptr, FCN8S_ERR_HIP, hipGetErrorString(e_)); } while (0)

uint32_t fcn8s_crc32c(const void* data, size_t n, uint32_t crc)
{
    static uint32_t table[8][256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
        init = true;
    }
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n >= 8) {                                   // slice-by-8
        uint32_t lo, hi; memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
            table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int fcn8s_onehot_to_ids(void* stream, const void* onehot, int elem_bytes, int64_t npix, int C, uint8_t* ids, int32_t* bad)
{
    if (!onehot || !ids || (elem_bytes != 1 && elem_bytes != 4) || C <= 0 || C > 255) return fail(nullptr, FCN8S_ERR_BAD_ARG, "fcn8s_onehot_to_ids: bad argument");
    launch_onehot_to_ids(onehot, elem_bytes, npix, C, ids, bad, (hipStream_t)stream); OPCHK(); return FCN8S_OK;
}

int fcn8s_op_preprocess(void* stream, const void* images, int dtype, float* out4, int64_t npix)
{ launch_preprocess(images, dtype, out4, npix, (hipStream_t)stream); OPCHK(); return FCN8S_OK; }

int fcn8s_op_augment_u8(void* stream, const uint8_t* images, const uint8_t* labels, uint8_t* out_images, uint8_t* out_labels,
                        const int32_t* params, const uint8_t* vlut, int N, int H, int W, int Ho, int Wo, int void_id)
{
    if (!images || !out_images || !params || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || (labels && !out_labels))
        return fail(nullptr, FCN8S_ERR_BAD_ARG, "augment_u8: bad argument");
    launch_augment_u8(images, labels, out_images, out_labels, params, vlut, N, H, W, Ho, Wo, void_id, (hipStream_t)stream);
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_resample_u8(void* stream, const uint8_t* images, const uint8_t* labels, uint8_t* out_images, uint8_t* out_labels,
                         const int32_t* params, int N, int H, int W, int Ho, int Wo, int void_id)
{
    if ((!images && !labels) || (images && !out_images) || (labels && !out_labels) || !params || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
        return fail(nullptr, FCN8S_ERR_BAD_ARG, "resample_u8: bad argument");
    launch_resample_u8(images, labels, images ? out_images : nullptr, labels ? out_labels : nullptr, params, N, H, W, Ho, Wo, void_id, (hipStream_t)stream);
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv2d(void* stream, const float* x, const float* w, const float* bias, float* y,
                    int N, int H, int W, int Cin, int Cout, int K, int relu)
{
    if (Cin % 4 || Cout % 4 || K % 2 == 0) return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d: Cin, Cout must be multiples of 4 and K odd");
    Epi e; e.bias = bias; e.relu = relu;
    conv_same(nullptr, "", x, w, y, N, H, W, Cin, Cout, K, e, (hipStream_t)stream);
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv2d_winograd(void* stream, const float* x, const float* w, const float* bias, float* y,
                             int N, int H, int W, int Cin, int Cout, int K, int relu, int tile)
{
    if ((tile != 2 && tile != 4 && tile != 6) || (K != 3 && K != 7) || (K == 7 && tile != 4) || Cin % 16 || Cout % 32 || (tile != 6 && (H % tile || W % tile)) || H % 2 || W % 2)
        return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d_winograd: needs tile in {2,4,6} (4 for K = 7), K in {3,7}, Cin % 16, Cout % 32, even H and W (multiples of tile for tile 2 and 4)");
    hipStream_t s = (hipStream_t)stream;
    const size_t T = (size_t)wino_tiles(tile, N, H, W), P = (size_t)wino_alpha(tile, K) * wino_alpha(tile, K), Kg = (size_t)wino_nsub(K) * wino_nsub(K) * Cin;
    float *u = nullptr, *v = nullptr, *mm = nullptr;
    if (hipMalloc((void**)&u, P * Kg * Cout * 4) != hipSuccess || hipMalloc((void**)&v, P * (size_t)wino_slab(T, (int)Kg) * 4) != hipSuccess ||
        hipMalloc((void**)&mm, P * (size_t)wino_slab(T, Cout) * 4) != hipSuccess) return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc");
    WinoEpi we; we.bias = bias; we.relu = relu;
    conv_winograd(nullptr, tile, K, "", x, w, y, u, v, mm, N, H, W, Cin, Cout, we, s, nullptr);
    hipStreamSynchronize(s); hipFree(u); hipFree(v); hipFree(mm);
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv3x3_winograd_fwd_bwd(void* stream, const float* x, const float* w, const float* bias, const float* dy, const float* dx_addend,
                                      float* y, float* pool, float* dx, float* dw, float* db,
                                      int N, int H, int W, int Cin, int Cout, int tile, int pooled, int mask_mode)
{
    if ((tile != 2 && tile != 4 && tile != 6) || Cin % 64 || Cout % 64 || H % 2 || W % 2 || (tile != 6 && (H % tile || W % tile)) || !x || !w || !dy || !dx || !dw ||
        (pooled && (!pool || tile < 4)) || (!pooled && !y) || mask_mode < 0 || mask_mode > 2)
        return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv3x3_winograd_fwd_bwd: needs tile in {2,4,6}, Cin % 64 == 0, Cout % 64 == 0, even H and W (multiples of tile for tiles 2, 4), "
                                                "pooled only with tiles 4 and 6, mask_mode in {0,1,2}");
    hipStream_t s = (hipStream_t)stream;
    // a bare model context: the launch sequences below are the model's own (conv_same / conv_wgrad), with one layer called "op"
    fcn8s_model mm; fcn8s_model* m = &mm;
    m->stream = s; m->wino_min_cin = 16; m->wino_tile = 6; m->wino_force_tile = tile; m->N = N; m->H = H; m->W = W;
    m->precision = t_op_split == 3 ? FCN8S_PREC_F32X3 : t_op_split == 2 ? FCN8S_PREC_F32X2 : FCN8S_PREC_F32;
    const int al = tile + 2, P = al * al, cmax = std::max(Cin, Cout);
    const size_t T = (size_t)wino_tiles(tile, N, H, W);
    const size_t slab_max = (size_t)P * (size_t)wino_slab((long long)T, cmax), slab_in = (size_t)P * (size_t)wino_slab((long long)T, Cin);
    std::vector<void*> owned;
    auto dalloc = [&](size_t nfloats) -> float* { void* p = nullptr; if (hipMalloc(&p, nfloats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; } owned.push_back(p); return (float*)p; };
    auto cleanup = [&]() { hipStreamSynchronize(s); for (void* p : owned) hipFree(p); for (auto& kv : m->u_train) if (kv.second) hipFree(kv.second); m->u_train.clear(); };
    m->d_wino_u = dalloc((size_t)P * cmax * cmax); m->d_wino_v = dalloc(slab_max); m->d_wino_m = dalloc(slab_max);
    float* wv = dalloc(slab_in); float* wt = dalloc((size_t)9 * Cin * Cout);
    float* pidx = pooled ? dalloc(((size_t)N * (H / 2) * (W / 2) * Cout + 3) / 4) : nullptr;
    float* rbits = mask_mode == 2 ? dalloc(wino_rbits_words(tile, N, H, W, Cin)) : nullptr;
    if (!m->d_wino_u || !m->d_wino_v || !m->d_wino_m || !wv || !wt || (pooled && !pidx) || (mask_mode == 2 && !rbits)) { cleanup(); return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc"); }
    Act a; a.p = wv; a.n = slab_in; m->acts["wv:op"] = a;
    // forward (training mode: keeps V and the filter bank, writes the pool argmax bytes / the input's ReLU bit record)
    m->fwd_train = true; m->train_mode = true;
    { Epi e; e.bias = bias; e.relu = 1;
      if (pooled) { e.pool_out = pool; e.pool_idx = (unsigned char*)pidx; e.skip_y = y == nullptr; }
      if (mask_mode == 2) { e.in_relu_bits_out = (unsigned*)rbits; e.in_layer = "prev"; }
      conv_same(m, "op", x, w, y, N, H, W, Cin, Cout, 3, e, s, 0, "op"); }
    // backward: weight + bias gradient in the Winograd domain, then the data gradient (adjoint form for tile 6)
    hipMemsetAsync(dw, 0, (size_t)9 * Cin * Cout * sizeof(float), s);
    if (db) hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s);
    conv_wgrad(m, "op", x, dy, dw, db, N, H, W, Cin, Cout, 3, 1.f, s, 0, "op", tile >= 4, pooled ? (const unsigned char*)pidx : nullptr);
    { Epi e; e.dgrad = 1; e.w_fwd = w; e.lazy_wt = 1; e.addend = dx_addend;
      if (mask_mode) { e.mask = x; e.mask_scale = 1.f; }
      if (mask_mode == 2 && m->rbits_ok.count("prev")) e.relu_bits_in = (const unsigned*)rbits;
      conv_same(m, "op", dy, wt, dx, N, H, W, Cout, Cin, 3, e, s, 0, "op"); }
    cleanup();
    m->acts.clear();
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv2d_bf16(void* stream, const float* x, const float* w, const float* bias, float* y,
                         int N, int H, int W, int Cin, int Cout, int K, int relu)
{
    if (Cin % 32 || Cout % 128 || K % 2 == 0) return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d_bf16: needs Cin % 32 == 0, Cout % 128 == 0, K odd");
    hipStream_t s = (hipStream_t)stream;
    unsigned short* wt = nullptr;
    if (hipMalloc((void**)&wt, (size_t)K * K * Cin * Cout * 2) != hipSuccess) return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc");
    launch_w_to_bf16_tiles(w, wt, K * K * Cin, Cout, s);
    Bf16ConvArgs a{};
    a.x = x; a.wt = wt; a.bias = bias; a.y = y; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K = K; a.relu = relu;
    launch_conv_bf16(a, s);
    hipStreamSynchronize(s); hipFree(wt);
    OPCHK(); return FCN8S_OK;
}

// One K x K SAME convolution in the arithmetic of FCN8S_PREC_BF16_TRAIN on the kernels that mode runs (gemm_bf16.hip): any of y (forward, + bias,
// optional ReLU), dx (data gradient of dy, optional mask: dx = mask > 0 ? dx : 0), dw / db (weight / bias gradient) may be NULL.
int fcn8s_op_conv2d_bf16_train(void* stream, const float* x, const float* w, const float* bias, float* y, int relu,
                               const float* dy, const float* mask, float* dx, float* dw, float* db,
                               int N, int H, int W, int Cin, int Cout, int K)
{
    if (Cin % 64 || Cout % 64 || K % 2 == 0) return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d_bf16_train: needs Cin % 64 == 0, Cout % 64 == 0, K odd");
    hipStream_t s = (hipStream_t)stream;
    const int pad = (K - 1) / 2, Wp_ = W + 2 * pad;
    const long long G = bf16_guard_rows(K, Wp_), R = (long long)N * (H + 2 * pad) * Wp_;
    // the padded copies as channel-chunk planes [C / 32][G + R + G][32], as the bf16_train mode keeps them (op option "op_bf16_planes" = 0: [rows][C], the
    // layout every kernel still takes -- plane stride 0 -- and the other bf16 modes use)
    const long long PS = t_op_planes ? (R + 2 * G) * 32 : 0;
    const long long OX = t_op_planes ? G * 32 : G * Cin, OY = t_op_planes ? G * 32 : G * Cout;
    unsigned short *xb = nullptr, *dyb = nullptr, *wt = nullptr;
    auto cleanup = [&]() { hipStreamSynchronize(s); if (xb) hipFree(xb); if (dyb) hipFree(dyb); if (wt) hipFree(wt); };
    const size_t nx = (size_t)(R + 2 * G) * Cin, ny = (size_t)(R + 2 * G) * Cout, nw = (size_t)K * K * Cin * Cout;
    if (hipMalloc((void**)&wt, nw * 2) != hipSuccess) { cleanup(); return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc"); }
    if (x) {
        if (hipMalloc((void**)&xb, nx * 2) != hipSuccess) { cleanup(); return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc"); }
        hipMemsetAsync(xb, 0, nx * 2, s);
        launch_f32_to_bf16_padded(x, xb + OX, N, H, W, Cin, pad, s, PS);
    }
    if (dy) {
        if (hipMalloc((void**)&dyb, ny * 2) != hipSuccess) { cleanup(); return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc"); }
        hipMemsetAsync(dyb, 0, ny * 2, s);
        launch_f32_to_bf16_padded(dy, dyb + OY, N, H, W, Cout, pad, s, PS);
    }
    bool ok = true;
    if (y && x && w) {
        launch_w_to_bf16_t(w, wt, K * K * Cin, Cout, s);
        Bf16Conv256Args g{};
        g.xp = xb + OX; g.xp_ps = PS; g.wt = wt; g.bias = bias; g.y = y; g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.K = K; g.relu = relu; g.any_shape = 1; g.mask_scale = 1.f; g.guarded = 1; g.rows_bn = t_op_rows_bn;
        ok = ok && launch_conv_bf16_256(g, s);
    }
    if (dx && dy && w) {
        launch_w_to_bf16_flip_t(w, wt, K, Cin, Cout, s);
        Bf16Conv256Args g{};
        g.xp = dyb + OY; g.xp_ps = PS; g.wt = wt; g.y = dx; g.N = N; g.H = H; g.W = W; g.Cin = Cout; g.Cout = Cin; g.K = K; g.mask = mask; g.mask_scale = 1.f; g.any_shape = 1; g.guarded = 1; g.rows_bn = t_op_rows_bn;
        ok = ok && launch_conv_bf16_256(g, s);
    }
    if (dw && x && dy) {
        Bf16WgradArgs g{};
        g.A = xb + OX; g.B = dyb + OY; g.a_ps = g.b_ps = PS; g.C = dw; g.R = R; g.Ci = Cin; g.Cj = Cout; g.K = K; g.Wp = Wp_;
        ok = ok && launch_wgrad_bf16(g, s);
    }
    if (db && dy) { hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s); launch_colsum(dy, db, (long long)N * H * W, Cout, s); }
    cleanup();
    if (!ok) return fail(nullptr, FCN8S_ERR_SHAPE, "conv2d_bf16_train: a launch refused the shape");
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv2d_bwd(void* stream, const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                        int N, int H, int W, int Cin, int Cout, int K)
{
    if (Cin % 4 || Cout % 4 || K % 2 == 0) return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d_bwd: Cin, Cout must be multiples of 4 and K odd");
    hipStream_t s = (hipStream_t)stream;
    if (dw) {
        hipMemsetAsync(dw, 0, (size_t)K * K * Cin * Cout * sizeof(float), s);
        if (db) hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s);
        conv_wgrad(nullptr, "", x, dy, dw, db, N, H, W, Cin, Cout, K, 1.f, s);
    } else if (db) { hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), s); launch_colsum(dy, db, (long long)N * H * W, Cout, s); }
    if (dx) {
        float* wt = nullptr;
        if (hipMalloc((void**)&wt, (size_t)K * K * Cin * Cout * sizeof(float)) != hipSuccess) return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc");
        launch_flip_transpose(w, wt, K * K, Cin, Cout, s);
        Epi e; conv_same(nullptr, "", dy, wt, dx, N, H, W, Cout, Cin, K, e, s);
        hipStreamSynchronize(s); hipFree(wt);
    }
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_maxpool2x2(void* stream, const float* x, float* y, int N, int H, int W, int C)
{
    if (C % 4 || H % 2 || W % 2) return fail(nullptr, FCN8S_ERR_BAD_ARG, "maxpool: C%4, H%2, W%2 must be 0");
    launch_maxpool_fwd(x, y, N, H, W, C, (hipStream_t)stream); OPCHK(); return FCN8S_OK;
}
int fcn8s_op_maxpool2x2_bwd(void* stream, const float* x, const float* dy, float* dx, int N, int H, int W, int C, int relu_mask)
{
    if (C % 4 || H % 2 || W % 2) return fail(nullptr, FCN8S_ERR_BAD_ARG, "maxpool: C%4, H%2, W%2 must be 0");
    launch_maxpool_bwd(x, dy, dx, N, H, W, C, relu_mask, (hipStream_t)stream); OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv2d_transpose(void* stream, const float* x, const float* w, const float* bias, const float* addend, float* y,
                              int N, int Hi, int Wi, int C, int K, int S)
{
    if (C % 4 || K != 2 * S) return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d_transpose: needs C%4==0 and K == 2*S");
    hipStream_t s = (hipStream_t)stream;
    float* wp = nullptr;
    if (hipMalloc((void**)&wp, (size_t)S * S * 4 * C * C * sizeof(float)) != hipSuccess) return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc");
    launch_tconv_phase_pack(w, wp, K, S, C, s);
    tconv_fwd(nullptr, x, wp, bias, addend, y, N, Hi, Wi, C, K, S, s);
    hipStreamSynchronize(s); hipFree(wp);
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_conv2d_transpose_bwd(void* stream, const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                  int N, int Hi, int Wi, int C, int K, int S)
{
    if (C % 4 || K != 2 * S) return fail(nullptr, FCN8S_ERR_BAD_ARG, "conv2d_transpose_bwd: needs C%4==0 and K == 2*S");
    hipStream_t s = (hipStream_t)stream;
    if (dw) { hipMemsetAsync(dw, 0, (size_t)K * K * C * C * sizeof(float), s); tconv_wgrad(nullptr, x, dy, dw, N, Hi, Wi, C, K, S, s); }
    if (db) { hipMemsetAsync(db, 0, (size_t)C * sizeof(float), s); launch_colsum(dy, db, (long long)N * Hi * S * Wi * S, C, s); }
    if (dx) tconv_dgrad(nullptr, dy, w, dx, N, Hi, Wi, C, K, S, s);
    OPCHK(); return FCN8S_OK;
}

int fcn8s_op_softmax_xent(void* stream, const float* logits, const uint8_t* labels, float* dlogits, float* loss_dev, int64_t npix, int C)
{
    hipStream_t s = (hipStream_t)stream;
    double* part = nullptr;
    if (hipMalloc((void**)&part, 4096 * sizeof(double)) != hipSuccess) return fail(nullptr, FCN8S_ERR_OOM, "hipMalloc");
    launch_softmax_xent(logits, labels, dlogits, part, npix, C, 1.0f / (float)npix, s);
    launch_finalize_loss(part, softmax_xent_blocks(npix), npix, nullptr, 0.f, loss_dev, s);
    hipStreamSynchronize(s); hipFree(part);
    OPCHK(); return FCN8S_OK;
}
int fcn8s_op_softmax_argmax(void* stream, const float* logits, float* sm, int64_t* am, int64_t npix, int C)
{ launch_softmax_argmax(logits, sm, (long long*)am, npix, C, (hipStream_t)stream); OPCHK(); return FCN8S_OK; }
int fcn8s_op_confusion(void* stream, const uint8_t* labels, const int64_t* pred, int64_t npix, int64_t* conf, int C)
{ launch_confusion(labels, (const long long*)pred, npix, (unsigned long long*)conf, C, (hipStream_t)stream); OPCHK(); return FCN8S_OK; }
int fcn8s_op_tf_adam(void* stream, float* theta, const float* g, float* mm, float* v, int64_t n, int t, float lr, float b1, float b2, float eps, float gs)
{
    const float lr_t = lr * (float)std::sqrt(1.0 - std::pow((double)b2, (double)t)) / (float)(1.0 - std::pow((double)b1, (double)t));
    launch_tf_adam(theta, g, mm, v, n, lr_t, b1, b2, eps, gs, (hipStream_t)stream); OPCHK(); return FCN8S_OK;
}
int fcn8s_op_sgd_momentum(void* stream, float* theta, const float* g, float* buf, int64_t n, float lr, float mom, float gs)
{ launch_sgd_momentum(theta, g, buf, n, lr, mom, gs, (hipStream_t)stream); OPCHK(); return FCN8S_OK; }

}  // extern "C"
