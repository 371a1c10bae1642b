/* fcn8s_hip.h -- C ABI of libfcn8s_hip.so, the MI355X (gfx950) replacement for
 * the TensorFlow session behind pierluigiferrari/fcn8s_tensorflow's `FCN8s`.
 *
 * The reference has no FFI: its hot path sits behind `tf.Session.run`
 * feed/fetch calls made by class FCN8s (fcn8s_tensorflow.py).  Each entry point
 * below cites the `sess.run` site (or graph-building method) it replaces.
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * Conventions
 *   - all tensors NHWC, C-contiguous; weights in the TensorFlow layouts
 *     (conv HWIO [kh,kw,Cin,Cout]; conv2d_transpose [kh,kw,Cout,Cin]);
 *   - variables are addressed by the reference's variable names
 *     ("conv1_1/filter" ... "fc7_pool4_pool3_conv2d_trans/bias",
 *     fcn8s_tensorflow.py:331-350);
 *   - every function returns FCN8S_OK (0) or an error code; the text is
 *     available from fcn8s_last_error();
 *   - `where` = FCN8S_HOST: pointers are host memory (copied in/out on the
 *     model's stream); FCN8S_DEVICE: pointers are device memory on the model's
 *     GPU and the call is stream-ordered;
 *   - a model is not thread-safe; one model per process per GPU
 *     (the reference is single-threaded, one tf.Session, :65).
 */
#ifndef FCN8S_HIP_H
#define FCN8S_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCN8S_OK            0
#define FCN8S_ERR_BAD_ARG   1   /* facade raises ValueError */
#define FCN8S_ERR_SHAPE     2   /* H or W not a multiple of 32, N<=0 ... (TF: InvalidArgumentError at tf.add :213/:224) */
#define FCN8S_ERR_OOM       3
#define FCN8S_ERR_HIP       4
#define FCN8S_ERR_STATE     5   /* call order violated (e.g. backward before forward) */
#define FCN8S_ERR_NOT_FOUND 6   /* unknown variable / activation name */
#define FCN8S_HOST   0
#define FCN8S_DEVICE 1

#define FCN8S_IMG_U8  0   /* uint8 RGB  [N,H,W,3] */
#define FCN8S_IMG_F32 1   /* float32 RGB [N,H,W,3] */

#define FCN8S_OPT_TF_ADAM      0  /* tf.train.AdamOptimizer, fcn8s_tensorflow.py:256 */
#define FCN8S_OPT_SGD_MOMENTUM 1  /* BASELINE.json config 3 */
#define FCN8S_OPT_NONE         2  /* caller updates the parameter buffer itself (torch optimizer over views) */

#define FCN8S_ERR_RCCL      7   /* a collective of the library's own RCCL communicator failed (fcn8s_comm_*) */
#define FCN8S_MAX_BUCKETS 8       /* upper bound of fcn8s_num_buckets(): gradient buckets in backward-production order */
#define FCN8S_NUM_STAGE_SLOTS 3   /* host-input staging slots (fcn8s_stage_inputs) */

#define FCN8S_PREC_F32     0      /* everything exact fp32 (the reference's arithmetic; default) */
#define FCN8S_PREC_BF16_FC 1      /* BASELINE.json config 5: the forward fc6 / fc7 contractions take bf16-rounded operands on the
                                     bf16 MFMA with fp32 accumulation; everything else, and the whole backward pass, stays fp32 */
#define FCN8S_PREC_F32X3   2      /* fp32-accurate products on the bf16 MFMA: in every LDS-DMA GEMM (Winograd positions, fc6, fc7, the last
                                     transposed conv; forward, data and weight gradients) each fp32 operand is split exactly into three
                                     bf16 pieces and six of the nine piece products are accumulated in fp32 -- error below one fp32
                                     rounding per product, not bit-identical to an fmaf chain; 1.25x faster GEMMs.  Not the default. */

#define FCN8S_PREC_BF16_FWD 3      /* BASELINE.json config 5 taken further ("bf16 fwd / fp32 accum"): besides fc6 / fc7, the forward convolutions of
                                     conv3_1 .. conv5_3 (Cout a multiple of 256) run as direct convolutions with bf16-rounded operands on the bf16
                                     MFMA, fp32 accumulation, fp32 outputs; every other GEMM of the step -- conv1_2 / conv2_x forward, all data and
                                     weight gradients -- uses the F32X3 arithmetic, i.e. all matrix work is on the bf16 MFMA.  Gradients are those of
                                     the fp32 graph evaluated at the bf16-forward activations (straight-through rounding). */
#define FCN8S_PREC_F32X2   4      /* like F32X3 with two pieces per operand, x ~ hi + lo: 16 significand bits of each operand enter the product
                                     (what is cut off is below 2^-17 |x|), three of the four piece products are accumulated in fp32.  Between TF32
                                     (11 bits) and fp32 (24); a reduced-precision mode, half the matrix-pipe and conversion work of F32X3. */
#define FCN8S_PREC_BF16_FWD_X2 5   /* BF16_FWD with the F32X2 arithmetic in place of F32X3 for every GEMM that is not a bf16 forward convolution
                                     (conv1_2 / conv2_x forward, all data and weight gradients): config 5's "bf16 fwd / fp32 accum" with 16-bit
                                     operands in the backward pass */
#define FCN8S_PREC_BF16_TRAIN 6    /* BASELINE.json config 5 with the backward pass on the bf16 pipe as well (round 5): every convolution
                                    * but conv1_1 -- conv1_2 .. conv5_3, fc6, fc7 -- is a DIRECT convolution whose operands are rounded to bf16 (RNE), products and
                                    * sums fp32: forward y = conv(bf16 x, bf16 w); data gradient dx = conv^T(bf16 dy, bf16 w); weight gradient
                                    * dw = corr(bf16 x, bf16 dy); bias gradient, ReLU / dropout masks, pools, the loss, the decoder and the optimizer exact fp32
                                    * on fp32 master weights.  No Winograd transform runs in this mode.  Needs channel widths % 64 == 0. */

typedef struct fcn8s_model fcn8s_model;

typedef struct fcn8s_config {
    int32_t  num_classes;   /* FCN8s(num_classes=...), :19 */
    int32_t  fc6_ksize;     /* 7 (convolutionalized VGG-16 fc6); 0 -> 7 */
    int32_t  widths[7];     /* conv1..conv5, fc6, fc7 channel counts; zeros -> 64,128,256,512,512,4096,4096 */
    int32_t  device_id;     /* HIP device ordinal */
    uint64_t seed;          /* dropout Philox key (per-rank seed for DP) */
    void*    ext_params;    /* optional caller-owned DEVICE buffer of fcn8s_param_floats() floats, or NULL */
    void*    ext_grads;     /* optional caller-owned DEVICE buffer of the same size, or NULL */
} fcn8s_config;

/* ---- layout of the flat variable buffer; pure host logic, needs no GPU ------- */
size_t      fcn8s_param_floats(const fcn8s_config* cfg);             /* size of the flat parameter buffer */
int         fcn8s_layout_num_params(const fcn8s_config* cfg);
int         fcn8s_layout_param(const fcn8s_config* cfg, int index, char name_out[64], int32_t* ndim,
                               int64_t shape[4], int64_t* offset_floats);
int         fcn8s_layout_num_buckets(const fcn8s_config* cfg);       /* = fcn8s_num_buckets of a model made from cfg */
int         fcn8s_layout_bucket(const fcn8s_config* cfg, int bucket, size_t* offset_floats, size_t* nfloats);

/* ---- lifetime: FCN8s.__init__ :19-125 / close :946-952 ------------------- */
int         fcn8s_create(const fcn8s_config* cfg, fcn8s_model** out);
int         fcn8s_destroy(fcn8s_model* m);                           /* frees everything; FCN8S_ERR_RCCL (text: fcn8s_last_error(NULL)) if the model's communicator had failed */
const char* fcn8s_last_error(const fcn8s_model* m);                  /* m may be NULL: last create() error */
int         fcn8s_set_stream(fcn8s_model* m, void* hip_stream);      /* run on the caller's stream (e.g. torch's current stream) */
int         fcn8s_synchronize(fcn8s_model* m);

/* ---- variables (graph.get_tensor_by_name, :331-350; Saver :124,:927,:943) --- */
int    fcn8s_num_params(const fcn8s_model* m);
int    fcn8s_param_info(const fcn8s_model* m, int index, const char** name, int32_t* ndim,
                        int64_t shape[4], int64_t* offset_floats);
int    fcn8s_param_index(const fcn8s_model* m, const char* name);    /* -1 if unknown */
int    fcn8s_set_param(fcn8s_model* m, const char* name, const float* host, size_t nfloats);
int    fcn8s_get_param(fcn8s_model* m, const char* name, float* host, size_t nfloats);
int    fcn8s_get_grad(fcn8s_model* m, const char* name, float* host, size_t nfloats);
void*  fcn8s_param_buffer(fcn8s_model* m, size_t* nfloats);          /* DEVICE pointers, flat, name->offset via param_info */
void*  fcn8s_grad_buffer(fcn8s_model* m, size_t* nfloats);
int    fcn8s_num_buckets(const fcn8s_model* m);                       /* a run-time value (4 today): size nothing by a constant */
int    fcn8s_bucket_range(const fcn8s_model* m, int bucket, size_t* offset_floats, size_t* nfloats);
int    fcn8s_init_params(fcn8s_model* m, uint64_t seed);              /* synthetic init: He-normal VGG, truncated-normal decoder (:159-160) */

/* ---- labels: the reference feeds one-hot rows [N,H,W,C] (placeholder int32 :110, bool from the generator,
 * helpers/ground_truth_conversion_utils.py:84-88).  The kernels consume uint8 class ids; this converts a
 * DEVICE one-hot tensor (elem_bytes 1 = bool/uint8, 4 = int32/float32 bit patterns, non-zero = set) into
 * ids on the device: ids[p] = first c with onehot[p,c] != 0 (= np.argmax of a one-hot row); an all-zero row becomes
 * id 255.  Ids outside [0, C) contribute neither loss nor gradient (softmax_cross_entropy_with_logits of an all-zero
 * row is 0, :253).  `bad_count` (device, may be NULL) is incremented by the number of rows that are not one-hot.       */
int fcn8s_onehot_to_ids(void* stream, const void* onehot_dev, int elem_bytes, int64_t npix, int C,
                        uint8_t* ids_dev, int32_t* bad_count_dev);

/* ---- training: sess.run([train_op,total_loss,global_step]) :554-572 -------- *
 * images: [N,H,W,3]; label_ids: uint8 class ids [N,H,W] (the argmax of the
 * one-hot rows the reference feeds, :110).  loss_out / step_out may be NULL
 * (then the call does not synchronise with the host).                          */
int fcn8s_train_step(fcn8s_model* m, const void* images, int image_dtype, const uint8_t* label_ids,
                     int N, int H, int W, float learning_rate, float keep_prob, float l2_rate,
                     int where, float* loss_out, int64_t* step_out);

/* split-phase form for data-parallel training (no counterpart in the single-device reference): forward + loss, then
 * the backward pass in fcn8s_num_buckets() calls, then the update.  The gradient buffer is cut into that many contiguous buckets in
 * backward-production order -- {fc7, decoder} (68 MB) | {fc6} (411 MB) | {conv4, conv5} (52 MB) | {conv1 .. conv3} (7 MB) at full
 * width -- and bucket b's gradients are final (stream-ordered) once the call
 * fcn8s_backward_bucket(m, fcn8s_bucket_complete_after(m, b)) has returned.  By default ("defer_wgrad" 0) that is call b itself, so
 * every bucket's exchange overlaps the rest of the backward pass; with defer_wgrad >= 1 the conv buckets (2, 3) are final only at the
 * last call, with defer_wgrad = 3 all four are.
 * fcn8s_bucket_wait makes another stream (RCCL's, a torch side stream) wait for exactly the last kernel that writes into bucket b
 * -- for fc6 that is its weight gradient, 3 ms of data gradient earlier than the end of call 1 -- without blocking the host; it is
 * valid once the completing call has returned and until the next fcn8s_forward_loss.
 * `grad_scale` multiplies the gradients inside the update (1/world_size).      */
int fcn8s_forward_loss(fcn8s_model* m, const void* images, int image_dtype, const uint8_t* label_ids,
                       int N, int H, int W, float keep_prob, float l2_rate, int where);
int fcn8s_backward_bucket(fcn8s_model* m, int bucket);
int fcn8s_bucket_complete_after(const fcn8s_model* m, int bucket);
int fcn8s_bucket_wait(fcn8s_model* m, int bucket, void* hip_stream);
int fcn8s_apply_update(fcn8s_model* m, int optimizer, float learning_rate, float grad_scale);   /* waits (stream-ordered) for pending fcn8s_allreduce_bucket calls */
int fcn8s_read_loss(fcn8s_model* m, float* loss_out);                /* synchronises */

/* ---- data parallelism inside the library: one RCCL rank per model (SURVEY section 7 step 7, 8b "RCCL error"; the reference is one
 * tf.Session on one device, fcn8s_tensorflow.py:65, so there is nothing to cite for the collective itself).  A caller that keeps the
 * reference's Python and binds this ABI gets multi-GPU training without torch.distributed:
 *     rank 0: fcn8s_comm_unique_id(id)  -> hand the FCN8S_COMM_ID_BYTES bytes to every rank by any means (file, socket, MPI)
 *     all   : fcn8s_comm_init(m, id, FCN8S_COMM_ID_BYTES, rank, world);  fcn8s_comm_broadcast_params(m, 0)
 *     step  : fcn8s_forward_loss; for b in 0 .. fcn8s_num_buckets-1 { fcn8s_backward_bucket(m, b); for every bucket r with
 *             fcn8s_bucket_complete_after(m, r) == b: fcn8s_allreduce_bucket(m, r) }; fcn8s_apply_update(m, opt, lr, 1.0f / world)
 *     eval  : fcn8s_eval_step ...; fcn8s_comm_allreduce_metrics(m); fcn8s_metrics_get
 * fcn8s_allreduce_bucket queues an in-place SUM all-reduce of the bucket on a stream the library owns, behind the event of the last
 * kernel that writes into the bucket (never behind later backward kernels) and returns at once; fcn8s_comm_wait makes the model's
 * stream wait for all of them (fcn8s_apply_update does it implicitly, and so do fcn8s_get_grad and the next fcn8s_backward_bucket(m, 0);
 * a caller that reads fcn8s_grad_buffer() itself calls fcn8s_comm_wait first).  fcn8s_train_step on a model whose communicator has more
 * than one rank runs exactly this sequence (it never trains diverging replicas silently).  librccl is opened with dlopen at the first
 * fcn8s_comm_* call (no RCCL headers are needed to build the library; environment variable FCN8S_RCCL_LIBRARY names the file to open
 * instead of the soname -- a site's own RCCL build, or the shared-memory stand-in of tests/fake_rccl -- and fcn8s_comm_init refuses a
 * library whose ncclGetVersion is not 2.x, the ABI whose enum values this library declares for itself): a failure there, or at the
 * enqueue of any collective, is FCN8S_ERR_RCCL with RCCL's text in fcn8s_last_error.  Failures AFTER the enqueue -- a peer that dies or
 * hangs -- are caught by a watchdog thread every communicator of more than one rank owns: it polls ncclCommGetAsyncError and the age of
 * EVERY collective in flight (the bucket all-reduces, the parameter broadcast, the metrics all-reduce; each is enqueued under the mutex
 * the watchdog aborts under), and on an asynchronous error, or when a collective has not completed within option "comm_timeout_ms"
 * (default 600 000), calls ncclCommAbort -- RCCL's kernels then leave the streams, so neither the model's stream nor a host
 * synchronisation waits for that peer forever -- and fcn8s_allreduce_bucket / fcn8s_comm_wait / fcn8s_apply_update / fcn8s_comm_* return
 * FCN8S_ERR_RCCL with the reason.  With more than one rank fcn8s_apply_update waits ON THE HOST (polling) for the pending all-reduces
 * before it queues the update: gradients whose exchange was aborted are never applied, and the failure is reported by the very call
 * that would have used them (cost: the host's run-ahead over one kernel launch per step).  fcn8s_comm_allreduce_metrics waits for its
 * sums the same way.  fcn8s_comm_destroy (also run by fcn8s_destroy, before anything else) drains the communicator's stream by polling
 * under the same rules, never by an unconditional synchronisation, and returns FCN8S_ERR_RCCL once if the communicator had failed.      */
/* "dddd:bb:dd.f" of a HIP device (hipDeviceGetPCIBusId): lets a launcher bind each rank's host threads and decode workers to the
 * NUMA node of its GPU through /sys/bus/pci/devices/<id>/local_cpulist (fcn8s_tensorflow_amd/dp.py: bind_to_gpu_numa). */
int fcn8s_device_pci_bus_id(int device_id, char* out, size_t len);
#define FCN8S_COMM_ID_BYTES 128
int fcn8s_comm_unique_id(void* id_out, size_t nbytes);
int fcn8s_comm_init(fcn8s_model* m, const void* unique_id, size_t nbytes, int rank, int world);
int fcn8s_comm_destroy(fcn8s_model* m);
int fcn8s_comm_info(const fcn8s_model* m, int* rank, int* world, int* rccl_version);   /* world = 0 without a communicator */
int fcn8s_allreduce_bucket(fcn8s_model* m, int bucket);
int fcn8s_comm_wait(fcn8s_model* m);
int fcn8s_comm_broadcast_params(fcn8s_model* m, int root);
int fcn8s_comm_allreduce_metrics(fcn8s_model* m);

/* ---- asynchronous host boundary (the reference's feed_dict copies, :558-560, are synchronous inside sess.run) ----------------
 * fcn8s_stage_inputs copies one host batch (images [N,H,W,3] uint8/float32, optional uint8 class ids [N,H,W]) into pinned
 * staging memory and starts the host-to-device copy on the model's own copy stream; it returns as soon as the copy is queued and
 * may be called from a feeder thread while the model's stream is busy with the previous step.  The device pointers it returns
 * are then fed to fcn8s_train_step / fcn8s_eval_step / fcn8s_predict with where = FCN8S_DEVICE, bracketed by fcn8s_stage_wait
 * (the model's stream waits for the copy) and fcn8s_stage_release (marks the point after which the slot may be refilled).
 * FCN8S_NUM_STAGE_SLOTS slots; refilling a slot waits for its previous release.                                              */
int fcn8s_stage_inputs(fcn8s_model* m, int slot, const void* images, int image_dtype, const uint8_t* label_ids,
                       int N, int H, int W, void** images_dev, uint8_t** labels_dev);
int fcn8s_stage_wait(fcn8s_model* m, int slot);
int fcn8s_stage_release(fcn8s_model* m, int slot);

/* ---- evaluation: sess.run(metric_update_ops) :685-689; reset :674; values :692 */
int fcn8s_eval_step(fcn8s_model* m, const void* images, int image_dtype, const uint8_t* label_ids,
                    int N, int H, int W, float l2_rate, int where);
int fcn8s_metrics_reset(fcn8s_model* m);
int fcn8s_metrics_get(fcn8s_model* m, double* mean_loss, double* mean_iou, double* accuracy);
/* all_classes != 0: mean IoU over ALL classes, an absent class counting as IoU 0 -- what tf.metrics.mean_iou of the tutorial's
 * TensorFlow 1.3.0 computes (fcn8s_tensorflow.py:291-293, fcn8s_tutorial.ipynb:311); 0: mean over the classes that occur in the
 * labels or the predictions (later TF 1.x; = fcn8s_metrics_get).  Identical whenever every class occurs.                        */
int fcn8s_metrics_get_ex(fcn8s_model* m, double* mean_loss, double* mean_iou, double* accuracy, int all_classes);
/* raw accumulators (for cross-rank reduction): confusion[C*C] row = label, col = prediction */
int fcn8s_metrics_raw(fcn8s_model* m, int64_t* confusion, double* loss_sum, int64_t* loss_count);
int fcn8s_metrics_set_raw(fcn8s_model* m, const int64_t* confusion, double loss_sum, int64_t loss_count);

/* ---- prediction: sess.run(predictions_argmax | softmax_output) :764-770 ---- *
 * argmax != 0: out = int64 [N,H,W]; else out = float32 softmax [N,H,W,C].        */
int fcn8s_predict(fcn8s_model* m, const void* images, int image_dtype, int N, int H, int W,
                  int argmax, void* out, int where);

/* ---- state that must round-trip: global_step :246,:526; Adam slots ---------- */
int64_t fcn8s_global_step(const fcn8s_model* m);
int     fcn8s_set_global_step(fcn8s_model* m, int64_t step);
int     fcn8s_get_opt_state(fcn8s_model* m, float* host_m, float* host_v, size_t nfloats);
int     fcn8s_set_opt_state(fcn8s_model* m, const float* host_m, const float* host_v, size_t nfloats);

/* ---- introspection for parity tests ----------------------------------------- *
 * names: "pool3","pool4","fc7","logits", "conv1_1"... (post-ReLU activations)     */
/* frozen = 1: the caller promises that the parameters stay constant (an evaluate() / predict loop, the reference's sessions never
 * train inside one, fcn8s_tensorflow.py:660-697, :743-770); the library then keeps derived tensors (Winograd-transformed filter
 * banks) across calls instead of rebuilding them per forward pass.  Any library call that changes parameters or starts a training
 * pass unfreezes; whoever writes into an external parameter buffer (fcn8s_config.ext_params) must call this with 0 first. */
int fcn8s_freeze_params(fcn8s_model* m, int frozen);

/* Arithmetic mode (FCN8S_PREC_*); not in the reference, which is fp32 throughout.
 * BF16_FC needs fc6/fc7 widths that are multiples of 128 and a conv5 width that is a multiple of 32 (BAD_ARG otherwise). */
int fcn8s_set_precision(fcn8s_model* m, int precision);
int fcn8s_get_precision(const fcn8s_model* m);

/* Algorithm options (not in the reference).  They select between maintained variants of the same arithmetic so that a parity report can
 * separate Winograd round-off from summation order; the defaults are the measured winners and production code never sets them.
 *   model options (m != NULL; setting one drops the workspace and all cached filter banks):
 *     "winograd_min_cin"  64   3x3 layers with at least this many input channels run through Winograd; 0 = direct convolution everywhere
 *     "winograd_tile"     6    largest 3x3 output tile (6, 4 or 2); per layer the allowed tile with the fewest multiplies is used
 *     "winograd_tile_hires", "winograd_hires_pixels"  0, 0   3x3 layers whose map has at least `hires_pixels` pixels per image use at most
 *                              tile `tile_hires` (F(4x4) carries half the round-off of F(6x6); the layers next to the input see the longest sums)
 *     "winograd_fc6"      1    fc6 as a 2x2 grid of 4x4 sub-filters through F(4x4,4x4); 0 = direct 7x7
 *     "tconv_gemm"        1    the 16x16/8 transposed conv as one GEMM over output blocks (blocked logits); 0 = 64 sub-pixel phases
 *     "defer_wgrad"       0    deferred weight gradients (an experiment kept for its evidence, profiles/r03_overlap_*.txt: co-running gains nothing on
 *                              gfx950, on shared or on disjoint CUs): 1 = the weight-gradient GEMMs of conv3_1 .. conv5_3 are held back and run on a
 *                              second stream beside the end of the data-gradient chain (blocks 2 and 1); 2 = fc6 / fc7 as well (fused
 *                              fcn8s_train_step only: the bucket API keeps buckets 0 and 1 final at their own calls, for an early all-reduce);
 *                              3 = fc6 / fc7 through the bucket API too (fcn8s_bucket_complete_after then names the last call for every bucket)
 *     "defer_start_block" 2    the VGG block at whose backward pass the held-back GEMMs are launched
 *     "defer_tail_cus"    0    > 0: from that block on the data-gradient chain runs on a stream restricted to the first n CUs and the held-back
 *                              GEMMs on the remaining 256 - n (hipExtStreamCreateWithCUMask); 0: both share all CUs
 *     "fuse_dgrad_dout"   1    inside a VGG block the gather kernel of a conv's data gradient writes dM = A dZ A^T of the previous conv directly
 *                              (that conv's weight gradient and adjoint data gradient consume only dM): its dZ is never written; 0 = two kernels
 *     "fuse_out_in"       1    forward, inside a VGG block: the output transform of a conv is fused with the input transform of the next one (both
 *                              F(6x6,3x3) on the same tile grid): Y = relu(A^T M A + b) stays in registers / LDS and the kernel writes the next conv's
 *                              V = B^T d B (bit-identical to the two-kernel form); the conv's own activation tensor is then never written and
 *                              fcn8s_get_activation of it returns FCN8S_ERR_STATE (fcn8s_get_relu_record still answers).  0 = two kernels; 1 = fused unless the
 *                              launch would need row ranges of fewer than four tile rows per block to fill the chip (a single image, conv5_x: the
 *                              recomputed halo rows then eat the gain); 2 = fused whenever the shapes allow
 *     "bf16_gemm256"      1    FCN8S_PREC_BF16_FC: fc6 / fc7 forward on the 256 x 256 LDS-DMA kernel -- 0 never, 1 when the launch has at least
 *                              128 tiles (training batches), 2 whenever the shapes allow (rows and Cout multiples of 256)
 *     "conv1_tiled"       1    conv1_1 forward on the spatial-tile kernel (halo tile in LDS); 0 = the LDS-DMA gather kernel (bit-identical results)
 *     "conv1_wgrad_mfma"  1    conv1_1 weight gradient as a (27 -> 32) x 64 MFMA product over pixels; 0 = the VALU kernel
 *     "conv1_in_transform" 1   conv1_1 (3x3, 3 -> 64, bias, ReLU) is evaluated inside conv1_2's F(6x6,3x3) input transform, on that transform's own 8 x 8
 *                              patches and straight from the preprocessed image: conv1_1's activation tensor (134 MB per 1024x512 image) is never written
 *                              or read, fcn8s_get_activation("conv1_1") returns FCN8S_ERR_STATE (fcn8s_get_relu_record still answers in training).  Same
 *                              products in another summation order (VALU fma chain instead of the MFMA kernel's).  0 = conv1_1 as a kernel of its own
 *     "bf16_copy_by_transform" 1  FCN8S_PREC_BF16_FWD / _X2, training: the padded bf16 copy of a layer's input that its direct bf16 convolution reads is
 *                              written by the layer's Winograd input transform (which runs anyway, for the weight gradient) instead of a
 *                              conversion pass of its own over the activations (identical bits); 0 = the separate pass
 *     "deterministic"     0    1 = every reduction that the default path splits over blocks and joins with fp32 atomics (weight gradients in and out of
 *                              the Winograd domain, conv1_1's and the score heads' weight gradients, the bf16 weight gradients, the last bias gradient, the L2
 *                              sum; split-K GEMMs are simply not split) stores one partial result per split into a scratch slab, and a second kernel adds
 *                              the slabs in split order: two runs of the same steps on the same inputs give the same bits.  +2.4 % at 16 x 1024x512
 *     "bf16_acts"         1    FCN8S_PREC_BF16_TRAIN, training passes: a tensor between two bf16 convolutions exists only as the consumer's padded bf16 copy -- written by the
 *                              producer's epilogue (conv -> conv activations, conv1_1 included; the last convolutions of blocks 1, 2, 5, which their pools read as
 *                              bf16 copies -- maxima and gradient routing among the bf16 values; the pooled maps pool1 / pool2 / pool5; the output gradient of a conv that
 *                              follows a bf16 conv, together with that layer's bias gradient).  fcn8s_get_activation of such a layer then returns FCN8S_ERR_STATE naming
 *                              this option; 0 keeps every fp32 tensor too (same values into every product: bit-identical losses and weight gradients)
 *     "bf16_fuse_convert" 0    FCN8S_PREC_BF16_TRAIN: the producing convolution's epilogue also writes its consumer's padded bf16 copy (measured slower)
 *     "bf16_fuse_pool"    1    FCN8S_PREC_BF16_TRAIN: the forward pools keep routing bytes (and write their consumer's bf16 copy), the max-pool backward kernel reads those bytes and
 *                              writes the last conv's padded bf16 dZ copy and bias gradient itself; 0 = plain pools + conversion passes
 *     "bf16_rows_bn"      0    FCN8S_PREC_BF16_TRAIN: 128 = the flat-position 3 x 3 convolution kernel takes its 128-column tile where it can (A/B; slower)
 *     "bf16_infer_copies" 1    FCN8S_PREC_BF16_TRAIN: evaluation / prediction passes take the training pass's data flow (padded bf16 copies written by the producers' epilogues, the
 *                              flat-position kernel, pools on the bf16 copies; logits = the training pass's, bit for bit); 0 = fp32 tensors converted layer by layer (round 5)
 *     "keep_output_gradients" 0 (tests) every weighted layer's fp32 output gradient dY is copied as the backward pass hands it to the layer's weight gradient and can be read with
 *                              fcn8s_get_activation(m, "dy:<layer>", ...) (conv1_1 .. conv5_3, fc6, fc7); FCN8S_ERR_STATE for a layer whose gradient travelled in another form
 *                              (these ten pick a kernel per launch and drop nothing)
 *     "comm_timeout_ms" 600000 the communicator's watchdog (see fcn8s_comm_init): a collective older than this is given up, the communicator aborted
 *   op-context options (m == NULL): the arithmetic of the op-level entry points below, which have no model.  The value belongs to the
 *   CALLING THREAD (thread-local) and is read by that thread's later fcn8s_op_* calls only; no model ever reads it, so two models -- or a
 *   feeder thread beside a compute thread -- cannot change each other's kernels:
 *     "op_f32x3"          0    their LDS-DMA GEMMs use the split-bf16 arithmetic of FCN8S_PREC_F32X3
 *     "op_split_pieces"   0    the same switch by piece count: 0 = f32 MFMA, 3 = FCN8S_PREC_F32X3, 2 = FCN8S_PREC_F32X2
 *     "op_bf16_planes"    1    fcn8s_op_conv2d_bf16_train keeps its padded bf16 copies as channel-chunk planes [C / 32][rows][32] (what FCN8S_PREC_BF16_TRAIN does);
 *                              0 = [rows][C], the layout the kernels also take (same results)
 *     "op_bf16_rows_bn"   0    the same switch ("bf16_rows_bn": 0 / 64 = 256 x 64 blocks, 128 = 128 x 128) for fcn8s_op_conv2d_bf16_train
 *     "op_deterministic"  0    the slab reductions of "deterministic" for the op-level entry points (a model call on the same thread sets the
 *                              thread's switch from that model's option: set it again before the next op-level call)
 * Unknown keys return FCN8S_ERR_NOT_FOUND. */
int fcn8s_set_option(fcn8s_model* m, const char* key, int64_t value);
int fcn8s_get_option(const fcn8s_model* m, const char* key, int64_t* value);

int fcn8s_get_activation(fcn8s_model* m, const char* name, float* host, size_t nfloats);
int fcn8s_get_dropout_masks(fcn8s_model* m, float* host_mask6, size_t n6, float* host_mask7, size_t n7);
/* Parity instrumentation (no counterpart in the reference): which element of each 2x2 window the backward pass of pool `block` (1..5)
 * routes the gradient to -- one byte per pooled element [N, h/2, w/2, c]: 0..3 = window element (2*row + col) holding the FIRST maximum,
 * 4 = the maximum is not > 0 (ReLU off: no gradient).  Max-pool's argmax is discontinuous at ties; the checker differentiates along
 * the recorded routes and verifies separately that every route that differs from its own is a tie to round-off.  Valid after a
 * training forward pass (fcn8s_forward_loss / fcn8s_train_step). */
int fcn8s_get_pool_routing(fcn8s_model* m, int block, unsigned char* host, size_t nbytes);
/* Parity instrumentation: the ReLU record the backward pass masks with, one byte (0 / 1) per element [N,h,w,c] of a conv that feeds another
 * conv ("conv1_1" .. "conv5_2") -- what `activation > 0` says when the activation exists; with "fuse_out_in" the activation of such a layer
 * is never written and this record is all there is.  Valid after a training forward pass whose layer kept a record (Winograd layers). */
int fcn8s_get_relu_record(fcn8s_model* m, const char* layer, unsigned char* host, size_t nbytes);

/* ---- host helper for the TF tensor-bundle writer (tf_bundle.py): CRC-32C (Castagnoli) of a host buffer ----- */
uint32_t fcn8s_crc32c(const void* data, size_t nbytes, uint32_t crc);

/* ---- in-library HIP-event timing of kernel groups (bench.py roofline) -------- *
 * groups: "conv3x3_fwd","conv3x3_dgrad","conv3x3_wgrad","fc_fwd","fc_dgrad","fc_wgrad",...
 * Fills total milliseconds, number of launches, algorithmic flops and bytes.
 * fcn8s_profile_enable(m, 2) splits the conv groups per layer ("conv3x3_fwd:conv1_2").  */
int fcn8s_profile_enable(fcn8s_model* m, int on);
int fcn8s_profile_reset(fcn8s_model* m);
int fcn8s_profile_num_groups(const fcn8s_model* m);
int fcn8s_profile_get(fcn8s_model* m, int group, const char** name, double* total_ms,
                      int64_t* launches, double* flops, double* bytes);

/* ---- single ops on DEVICE pointers (unit parity tests; same kernels the model
 * uses).  `stream` may be NULL (default stream).                                 */
int fcn8s_op_preprocess(void* stream, const void* images, int image_dtype, float* out4, int64_t npix);
/* GPU-side augmentation of a uint8 batch on DEVICE pointers (the crop / canvas placement, horizontal flip and
 * brightness steps of data_generator/batch_generator.py:293-341, :469-486, which do not resample).  params: int32[4] per image =
 * {y offset, x offset, flip (0|1), brightness (0|1)}; out[n,y,x] = in[n, y+oy, (flip ? Wo-1-x : x) + ox], zero / void_id outside the
 * source.  Brightness is the reference's `_brightness` in OpenCV's 8-bit arithmetic (cv2.COLOR_RGB2HSV integer path, V replaced through
 * vlut[n][V] -- uint8[N][256], the host's evaluation of `V * random_br` saturated at 255 and truncated --, cv2.COLOR_HSV2RGB float32 path);
 * bit-exact with fcn8s_tensorflow_amd/cv2_compat.py.  labels / out_labels / vlut may be NULL. */
int fcn8s_op_augment_u8(void* stream, const uint8_t* images, const uint8_t* labels, uint8_t* out_images, uint8_t* out_labels,
                        const int32_t* params, const uint8_t* vlut, int N, int H, int W, int Ho, int Wo, int void_id);
/* the resampling augmentations of data_generator/batch_generator.py:328-384 (resize :328-331, translate :344-356, scale :358-384) on
 * DEVICE uint8 batches: image n is resized to params[4n+0] x params[4n+1] -- images as cv2.resize(INTER_LINEAR) does for 8-bit data (11-bit
 * fixed-point taps; 2x2 box mean for an exact 2x shrink), labels as cv2.resize(INTER_NEAREST) does (floor(dst * src/dst)) -- and placed at
 * offset (params[4n+2], params[4n+3]) of the [Ho,Wo] output, uncovered pixels 0 / void_id.  Bit-exact with cv2_compat.py.
 * images / labels may each be NULL. */
int fcn8s_op_resample_u8(void* stream, const uint8_t* images, const uint8_t* labels, uint8_t* out_images, uint8_t* out_labels,
                         const int32_t* params, int N, int H, int W, int Ho, int Wo, int void_id);
int fcn8s_op_conv2d(void* stream, const float* x, const float* w_hwio, const float* bias, float* y,
                    int N, int H, int W, int Cin, int Cout, int K, int relu);
/* the same SAME conv through Winograd F(tile x tile, 3x3), tile = 2, 4 or 6 (the path the model takes for its 3x3
 * layers; tile 6 handles partial edge tiles, tiles 2 and 4 need H, W multiples of tile; K = 7 with tile 4 runs fc6's
 * decomposition into 4x4 sub-filters, F(4x4,4x4)) */
int fcn8s_op_conv2d_winograd(void* stream, const float* x, const float* w_hwio, const float* bias, float* y,
                             int N, int H, int W, int Cin, int Cout, int K, int relu, int tile);
/* One 3x3 SAME conv + bias + ReLU the way the model's 3x3 layers run in a TRAINING step, forward and backward, through the model's own
 * launch sequences (Winograd F(tile x tile, 3x3), tile = 2, 4 or 6; Cin, Cout multiples of 64):
 *   forward   y = relu(conv(x, w) + bias), keeping the transformed input V and the filter bank U; pooled != 0: the output transform
 *             also writes pool = maxpool2x2(y) [N,H/2,W/2,Cout] and one argmax byte per window (y may then be NULL: a block's last
 *             conv output is never materialised);
 *   backward  dy = gradient w.r.t. the PRE-activation [N,H,W,Cout] (pooled == 0), or w.r.t. the pool output [N,H/2,W/2,Cout]
 *             (pooled != 0: max-pool and ReLU backward happen inside the transform, routed by the argmax bytes);
 *             dM = A dY A^T once; dw = Winograd-domain weight gradient V^T dM; db from dM's (1,1) slab; dx = data gradient, for tile 6
 *             as the adjoint of the forward algorithm (dV = dM U^T on the transposed-B GEMM, overlap-add gather), for tiles 2 / 4 as a
 *             forward-type Winograd conv on flipped filters; + dx_addend (optional skip gradient), then masked by (x > 0) if
 *             mask_mode = 1 (x read as the mask) or 2 (one-bit record written by this conv's input transform, the conv1_1 case).
 * Not in the reference; exists so that the Winograd backward has op-level parity tests. */
int fcn8s_op_conv3x3_winograd_fwd_bwd(void* stream, const float* x, const float* w_hwio, const float* bias, const float* dy, const float* dx_addend,
                                      float* y, float* pool, float* dx, float* dw, float* db,
                                      int N, int H, int W, int Cin, int Cout, int tile, int pooled, int mask_mode);
/* the same SAME conv with bf16-rounded operands and fp32 accumulation on the bf16 MFMA (FCN8S_PREC_BF16_FC's kernel);
 * Cin % 32 == 0, Cout % 128 == 0, K odd */
int fcn8s_op_conv2d_bf16(void* stream, const float* x, const float* w_hwio, const float* bias, float* y,
                         int N, int H, int W, int Cin, int Cout, int K, int relu);
/* One K x K SAME convolution in the arithmetic of FCN8S_PREC_BF16_TRAIN, on the kernels that mode runs: forward y = conv(bf16 x, bf16 w) + bias (ReLU if
 * relu), data gradient dx = conv^T(bf16 dy, bf16 w) (masked by `mask` > 0 if given), weight gradient dw = corr(bf16 x, bf16 dy), bias gradient
 * db = sum dy.  Any of y, dx, dw, db may be NULL (then x / dy / w are only needed for what is asked).  Cin, Cout % 64 == 0, K odd. */
int fcn8s_op_conv2d_bf16_train(void* stream, const float* x, const float* w_hwio, const float* bias, float* y, int relu,
                               const float* dy, const float* mask, float* dx, float* dw, float* db,
                               int N, int H, int W, int Cin, int Cout, int K);
int fcn8s_op_conv2d_bwd(void* stream, const float* x, const float* w_hwio, const float* dy,
                        float* dx, float* dw, float* db,
                        int N, int H, int W, int Cin, int Cout, int K);
int fcn8s_op_maxpool2x2(void* stream, const float* x, float* y, int N, int H, int W, int C);
int fcn8s_op_maxpool2x2_bwd(void* stream, const float* x, const float* dy, float* dx,
                            int N, int H, int W, int C, int relu_mask);
int fcn8s_op_conv2d_transpose(void* stream, const float* x, const float* w_kkoi, const float* bias,
                              const float* addend, float* y,
                              int N, int Hi, int Wi, int C, int K, int S);
int fcn8s_op_conv2d_transpose_bwd(void* stream, const float* x, const float* w_kkoi, const float* dy,
                                  float* dx, float* dw, float* db,
                                  int N, int Hi, int Wi, int C, int K, int S);
int fcn8s_op_softmax_xent(void* stream, const float* logits, const uint8_t* label_ids, float* dlogits,
                          float* loss_out_dev, int64_t npix, int C);
int fcn8s_op_softmax_argmax(void* stream, const float* logits, float* softmax_out, int64_t* argmax_out,
                            int64_t npix, int C);
int fcn8s_op_confusion(void* stream, const uint8_t* label_ids, const int64_t* pred_ids, int64_t npix,
                       int64_t* conf, int C);
int fcn8s_op_tf_adam(void* stream, float* theta, const float* g, float* m, float* v, int64_t n, int t,
                     float lr, float beta1, float beta2, float eps, float grad_scale);
int fcn8s_op_sgd_momentum(void* stream, float* theta, const float* g, float* buf, int64_t n,
                          float lr, float momentum, float grad_scale);

#ifdef __cplusplus
}
#endif
#endif /* FCN8S_HIP_H */
