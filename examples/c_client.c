/* A plain C99 host of libfcn8s_hip.so: no Python, no PyTorch, no HIP headers -- only include/fcn8s_hip.h.
 *
 * It walks the three hot sess.run sites of the reference through the C ABI with HOST buffers, the way a C / C++ / cgo / JNI caller
 * would: the training step (fcn8s_tensorflow.py:554-572), the evaluation step + metric read-out (:685-692) and prediction (:764-770),
 * then the split-phase form a data-parallel caller uses (forward + loss, backward bucket by bucket, update), then a state round trip
 * (parameters, optimizer slots, global step) into a second model that must continue like the first.  Everything it prints is a
 * key=value line; tests/test_c_client_gpu.py runs it on the MI355X and compares the numbers with the same calls made through ctypes.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_client.c -o examples/c_client -Lfcn8s_tensorflow_amd -lfcn8s_hip \
 *       -Wl,-rpath,'$ORIGIN/../fcn8s_tensorflow_amd' -Wl,-rpath-link,/opt/rocm/lib
 *   examples/c_client [N H W steps]            (defaults 2 64 96 3)
 *   examples/c_client --layout                 (no GPU needed: prints the flat variable layout and the gradient buckets)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "fcn8s_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != FCN8S_OK) { \
    fprintf(stderr, "%s failed: status %d: %s\n", #call, rc_, fcn8s_last_error(m)); return 10 + rc_; } } while (0)

/* the generator every language can restate: x -> x * 6364136223846793005 + 1442695040888963407 (Knuth), top bits used */
static uint64_t lcg_state;
static uint32_t lcg(void) { lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(lcg_state >> 33); }

static int print_layout(void)
{
    fcn8s_config cfg; memset(&cfg, 0, sizeof cfg); cfg.num_classes = 20;
    const int n = fcn8s_layout_num_params(&cfg);
    printf("param_floats=%zu\nnum_params=%d\n", fcn8s_param_floats(&cfg), n);
    for (int i = 0; i < n; ++i) {
        char name[64]; int32_t nd; int64_t shape[4], off;
        if (fcn8s_layout_param(&cfg, i, name, &nd, shape, &off) != FCN8S_OK) return 1;
        printf("param %d %s offset=%lld shape=", i, name, (long long)off);
        for (int d = 0; d < nd; ++d) printf("%s%lld", d ? "x" : "", (long long)shape[d]);
        printf("\n");
    }
    const int nb = fcn8s_layout_num_buckets(&cfg);
    for (int b = 0; b < nb; ++b) {
        size_t off, cnt;
        if (fcn8s_layout_bucket(&cfg, b, &off, &cnt) != FCN8S_OK) return 1;
        printf("bucket %d offset=%zu floats=%zu\n", b, off, cnt);
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && strcmp(argv[1], "--layout") == 0) return print_layout();
    const int N = argc > 1 ? atoi(argv[1]) : 2, H = argc > 2 ? atoi(argv[2]) : 64, W = argc > 3 ? atoi(argv[3]) : 96;
    const int steps = argc > 4 ? atoi(argv[4]) : 3, C = 20;
    const size_t npix = (size_t)N * H * W;

    fcn8s_model* m = NULL;
    fcn8s_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.num_classes = C; cfg.device_id = 0; cfg.seed = 1234;
    {
        int rc = fcn8s_create(&cfg, &m);
        if (rc != FCN8S_OK) { fprintf(stderr, "fcn8s_create failed: status %d: %s\n", rc, fcn8s_last_error(NULL)); return 2; }
    }
    CHECK(fcn8s_init_params(m, 7));

    /* synthetic batch: uint8 RGB [N,H,W,3], uint8 class ids [N,H,W] -- labels follow the image, so that there is something to learn */
    uint8_t* img = (uint8_t*)malloc(npix * 3); uint8_t* lab = (uint8_t*)malloc(npix);
    int64_t* pred = (int64_t*)malloc(npix * sizeof(int64_t));
    if (!img || !lab || !pred) return 3;
    lcg_state = 42;
    for (size_t p = 0; p < npix; ++p) {
        const size_t x = p % (size_t)W, y = (p / (size_t)W) % (size_t)H;
        const uint8_t cls = (uint8_t)(((x / 16) + 3 * (y / 16)) % (size_t)C);
        lab[p] = cls;
        for (int c = 0; c < 3; ++c) img[3 * p + c] = (uint8_t)((cls * 12 + 37 * c + (lcg() & 31)) & 255);
    }

    /* (1) training steps, loss and step counter fetched every step like the reference's sess.run */
    for (int s = 0; s < steps; ++s) {
        float loss = 0.f; int64_t gs = 0;
        CHECK(fcn8s_train_step(m, img, FCN8S_IMG_U8, lab, N, H, W, 1e-4f, 0.5f, 1e-3f, FCN8S_HOST, &loss, &gs));
        printf("train_step=%lld loss=%.9g\n", (long long)gs, (double)loss);
        if (!isfinite(loss)) { fprintf(stderr, "non-finite loss\n"); return 4; }
    }

    /* (2) evaluation: reset, one batch, read the streaming metrics */
    double mloss, miou, acc;
    CHECK(fcn8s_metrics_reset(m));
    CHECK(fcn8s_eval_step(m, img, FCN8S_IMG_U8, lab, N, H, W, 1e-3f, FCN8S_HOST));
    CHECK(fcn8s_metrics_get(m, &mloss, &miou, &acc));
    printf("eval_loss=%.9g eval_mean_iou=%.9g eval_accuracy=%.9g\n", mloss, miou, acc);

    /* (3) prediction: int64 argmax [N,H,W] into a host buffer the caller owns */
    CHECK(fcn8s_predict(m, img, FCN8S_IMG_U8, N, H, W, 1, pred, FCN8S_HOST));
    {
        uint64_t sum = 0; size_t agree = 0; int bad = 0;
        for (size_t p = 0; p < npix; ++p) { sum = sum * 31 + (uint64_t)pred[p]; agree += (pred[p] == (int64_t)lab[p]); bad |= (pred[p] < 0 || pred[p] >= C); }
        printf("predict_checksum=%llu predict_agree=%zu predict_pixels=%zu\n", (unsigned long long)sum, agree, npix);
        if (bad) { fprintf(stderr, "prediction outside [0, C)\n"); return 5; }
        /* the metric kernel and the prediction kernel must tell the same story about the same batch */
        if (fabs((double)agree / (double)npix - acc) > 1e-12) { fprintf(stderr, "accuracy %.12f != argmax agreement %.12f\n", acc, (double)agree / (double)npix); return 6; }
    }

    /* (4) the split-phase step a data-parallel host drives: forward + loss, the backward pass bucket by bucket (a collective over
     *     bucket b may start once call fcn8s_bucket_complete_after(m, b) has returned), then the update */
    {
        const int nb = fcn8s_num_buckets(m);
        float loss = 0.f;
        CHECK(fcn8s_forward_loss(m, img, FCN8S_IMG_U8, lab, N, H, W, 0.5f, 1e-3f, FCN8S_HOST));
        for (int b = 0; b < nb; ++b) CHECK(fcn8s_backward_bucket(m, b));
        CHECK(fcn8s_read_loss(m, &loss));
        CHECK(fcn8s_apply_update(m, FCN8S_OPT_TF_ADAM, 1e-4f, 1.0f));
        printf("split_step=%lld buckets=%d loss=%.9g\n", (long long)fcn8s_global_step(m), nb, (double)loss);
        /* call order is checked by the library, not assumed */
        if (fcn8s_backward_bucket(m, 0) != FCN8S_ERR_STATE) { fprintf(stderr, "backward without a forward pass was accepted\n"); return 7; }
    }

    /* (5) state round trip into a second model: parameters by reference variable name, Adam slots, global step; the next training
     *     step of both models (same dropout key) must then produce the same loss */
    {
        fcn8s_model* m2 = NULL;
        if (fcn8s_create(&cfg, &m2) != FCN8S_OK) { fprintf(stderr, "second fcn8s_create failed: %s\n", fcn8s_last_error(NULL)); return 8; }
        const size_t total = fcn8s_param_floats(&cfg);
        float* buf = (float*)malloc(total * sizeof(float)); float* mo = (float*)malloc(total * sizeof(float)); float* ve = (float*)malloc(total * sizeof(float));
        if (!buf || !mo || !ve) return 3;
        const int np_ = fcn8s_num_params(m);
        for (int i = 0; i < np_; ++i) {
            const char* name; int32_t nd; int64_t shape[4], off;
            CHECK(fcn8s_param_info(m, i, &name, &nd, shape, &off));
            size_t cnt = 1; for (int d = 0; d < nd; ++d) cnt *= (size_t)shape[d];
            CHECK(fcn8s_get_param(m, name, buf, cnt));
            if (fcn8s_set_param(m2, name, buf, cnt) != FCN8S_OK) { fprintf(stderr, "set_param %s: %s\n", name, fcn8s_last_error(m2)); return 9; }
        }
        CHECK(fcn8s_get_opt_state(m, mo, ve, total));
        if (fcn8s_set_opt_state(m2, mo, ve, total) != FCN8S_OK || fcn8s_set_global_step(m2, fcn8s_global_step(m)) != FCN8S_OK) { fprintf(stderr, "state import: %s\n", fcn8s_last_error(m2)); return 9; }
        float la = 0.f, lb = 0.f; int64_t sa = 0, sb = 0;
        CHECK(fcn8s_train_step(m, img, FCN8S_IMG_U8, lab, N, H, W, 1e-4f, 0.5f, 1e-3f, FCN8S_HOST, &la, &sa));
        if (fcn8s_train_step(m2, img, FCN8S_IMG_U8, lab, N, H, W, 1e-4f, 0.5f, 1e-3f, FCN8S_HOST, &lb, &sb) != FCN8S_OK) { fprintf(stderr, "resumed step: %s\n", fcn8s_last_error(m2)); return 9; }
        printf("resume_step=%lld resume_loss=%.9g original_loss=%.9g\n", (long long)sb, (double)lb, (double)la);
        /* (equal up to the summation order of the few small launches that split a reduction over blocks and add the parts atomically) */
        if (sa != sb || fabs((double)la - (double)lb) > 1e-6 * fabs((double)la)) { fprintf(stderr, "the restored model did not continue like the original\n"); return 9; }
        free(buf); free(mo); free(ve);
        fcn8s_destroy(m2);
    }
    CHECK(fcn8s_destroy(m));
    free(img); free(lab); free(pred);
    printf("c_client=ok\n");
    return 0;
}
