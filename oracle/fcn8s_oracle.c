/* Explicit-loop C restatement of the TensorFlow-1.x op semantics the FCN-8s hot
 * path is built from.  TEST INFRASTRUCTURE ONLY: it pins oracle/fcn8s_oracle.py
 * (torch-CPU) with known-answer checks and is never linked into the product.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for these ops (they
 * live in TensorFlow 1.x, un-vendored); each function restates the published
 * definition of the TF op created at the cited reference line.
 *
 * All tensors NHWC, float32 storage, double accumulation.
 * Build: make -C oracle   ->  oracle/libfcn8s_oracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX4(n, h, w, c, H, W, C) ((((size_t)(n) * (H) + (h)) * (W) + (w)) * (C) + (c))

/* K0 [INFERRED, inside the SavedModel]: RGB uint8 -> BGR float minus VGG mean;
 * image fed at fcn8s_tensorflow.py:558. */
void orc_preprocess_u8(const uint8_t* img, float* out, size_t npix)
{
    static const float mean_bgr[3] = {103.939f, 116.779f, 123.68f};
    for (size_t p = 0; p < npix; ++p)
        for (int c = 0; c < 3; ++c)
            out[p * 3 + c] = (float)img[p * 3 + (2 - c)] - mean_bgr[c];
}

/* K1/K3/K5/K7: tf.nn.conv2d stride 1, SAME, HWIO weights, + bias (+ ReLU).
 * VGG layers [INFERRED]; decoder 1x1 heads fcn8s_tensorflow.py:173-200. */
void orc_conv2d_same(const float* x, const float* w, const float* b, float* y,
                     int N, int H, int W, int Cin, int Cout, int K, int relu)
{
    const int p = (K - 1) / 2;
    for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
    for (int ww = 0; ww < W; ++ww)
    for (int co = 0; co < Cout; ++co) {
        double acc = b ? b[co] : 0.0;
        for (int ky = 0; ky < K; ++ky) {
            const int ih = h + ky - p;
            if (ih < 0 || ih >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int iw = ww + kx - p;
                if (iw < 0 || iw >= W) continue;
                for (int ci = 0; ci < Cin; ++ci)
                    acc += (double)x[IDX4(n, ih, iw, ci, H, W, Cin)] *
                           w[(((size_t)ky * K + kx) * Cin + ci) * Cout + co];
            }
        }
        if (relu && acc < 0) acc = 0;
        y[IDX4(n, h, ww, co, H, W, Cout)] = (float)acc;
    }
}

/* Autodiff of the above (AdamOptimizer.minimize, fcn8s_tensorflow.py:257):
 * dx = dgrad, dw = wgrad, db = sum. */
void orc_conv2d_same_bwd(const float* x, const float* w, const float* dy,
                         float* dx, float* dw, float* db,
                         int N, int H, int W, int Cin, int Cout, int K)
{
    const int p = (K - 1) / 2;
    const size_t nx = (size_t)N * H * W * Cin, nw = (size_t)K * K * Cin * Cout;
    double* ax = (double*)malloc(nx * sizeof(double)); double* aw = (double*)malloc(nw * sizeof(double));
    memset(ax, 0, nx * sizeof(double)); memset(aw, 0, nw * sizeof(double));
    for (int co = 0; co < Cout; ++co) {
        double s = 0;
        for (size_t q = 0; q < (size_t)N * H * W; ++q) s += dy[q * Cout + co];
        if (db) db[co] = (float)s;
    }
    for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
    for (int ww = 0; ww < W; ++ww)
    for (int ky = 0; ky < K; ++ky) {
        const int ih = h + ky - p;
        if (ih < 0 || ih >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int iw = ww + kx - p;
            if (iw < 0 || iw >= W) continue;
            for (int ci = 0; ci < Cin; ++ci) {
                const size_t xi = IDX4(n, ih, iw, ci, H, W, Cin);
                for (int co = 0; co < Cout; ++co) {
                    const size_t wi = (((size_t)ky * K + kx) * Cin + ci) * Cout + co;
                    const double g = dy[IDX4(n, h, ww, co, H, W, Cout)];
                    ax[xi] += g * w[wi];
                    aw[wi] += g * x[xi];
                }
            }
        }
    }
    if (dx) for (size_t i = 0; i < nx; ++i) dx[i] = (float)ax[i];
    if (dw) for (size_t i = 0; i < nw; ++i) dw[i] = (float)aw[i];
    free(ax); free(aw);
}

/* K2: tf.nn.max_pool 2x2 stride 2 SAME [INFERRED]; H, W even. */
void orc_maxpool2x2(const float* x, float* y, int N, int H, int W, int C)
{
    const int Ho = H / 2, Wo = W / 2;
    for (int n = 0; n < N; ++n)
    for (int h = 0; h < Ho; ++h)
    for (int w = 0; w < Wo; ++w)
    for (int c = 0; c < C; ++c) {
        float m = x[IDX4(n, 2 * h, 2 * w, c, H, W, C)];
        for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const float v = x[IDX4(n, 2 * h + dy, 2 * w + dx, c, H, W, C)];
            if (v > m) m = v;
        }
        y[IDX4(n, h, w, c, Ho, Wo, C)] = m;
    }
}

/* MaxPoolGrad: gradient goes to the first maximal element of each window. */
void orc_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C)
{
    const int Ho = H / 2, Wo = W / 2;
    memset(dx, 0, (size_t)N * H * W * C * sizeof(float));
    for (int n = 0; n < N; ++n)
    for (int h = 0; h < Ho; ++h)
    for (int w = 0; w < Wo; ++w)
    for (int c = 0; c < C; ++c) {
        int bi = 0; float m = x[IDX4(n, 2 * h, 2 * w, c, H, W, C)];
        for (int q = 1; q < 4; ++q) {
            const float v = x[IDX4(n, 2 * h + q / 2, 2 * w + q % 2, c, H, W, C)];
            if (v > m) { m = v; bi = q; }
        }
        dx[IDX4(n, 2 * h + bi / 2, 2 * w + bi % 2, c, H, W, C)] = dy[IDX4(n, h, w, c, Ho, Wo, C)];
    }
}

/* K8: tf.layers.conv2d_transpose(padding='same'), kernel [kh,kw,Cout,Cin],
 * fcn8s_tensorflow.py:204-233.  y[n,i*s+ky-p,j*s+kx-p,co] += x[n,i,j,ci]*W[ky,kx,co,ci],
 * p = (k-s)/2, out = in*s, then + b[co]. */
void orc_conv2d_transpose_same(const float* x, const float* w, const float* b, float* y,
                               int N, int Hi, int Wi, int Cin, int Cout, int K, int S)
{
    const int p = (K - S) / 2, Ho = Hi * S, Wo = Wi * S;
    const size_t ny = (size_t)N * Ho * Wo * Cout;
    double* ay = (double*)malloc(ny * sizeof(double));
    for (size_t q = 0; q < ny; ++q) ay[q] = b ? b[q % Cout] : 0.0;
    for (int n = 0; n < N; ++n)
    for (int i = 0; i < Hi; ++i)
    for (int j = 0; j < Wi; ++j)
    for (int ky = 0; ky < K; ++ky) {
        const int oy = i * S + ky - p;
        if (oy < 0 || oy >= Ho) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int ox = j * S + kx - p;
            if (ox < 0 || ox >= Wo) continue;
            for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                ay[IDX4(n, oy, ox, co, Ho, Wo, Cout)] +=
                    (double)x[IDX4(n, i, j, ci, Hi, Wi, Cin)] *
                    w[(((size_t)ky * K + kx) * Cout + co) * Cin + ci];
        }
    }
    for (size_t q = 0; q < ny; ++q) y[q] = (float)ay[q];
    free(ay);
}

void orc_conv2d_transpose_same_bwd(const float* x, const float* w, const float* dy,
                                   float* dx, float* dw, float* db,
                                   int N, int Hi, int Wi, int Cin, int Cout, int K, int S)
{
    const int p = (K - S) / 2, Ho = Hi * S, Wo = Wi * S;
    const size_t nx = (size_t)N * Hi * Wi * Cin, nw = (size_t)K * K * Cout * Cin;
    double* ax = (double*)malloc(nx * sizeof(double)); double* aw = (double*)malloc(nw * sizeof(double));
    memset(ax, 0, nx * sizeof(double)); memset(aw, 0, nw * sizeof(double));
    for (int co = 0; co < Cout; ++co) {
        double s = 0;
        for (size_t q = 0; q < (size_t)N * Ho * Wo; ++q) s += dy[q * Cout + co];
        if (db) db[co] = (float)s;
    }
    for (int n = 0; n < N; ++n)
    for (int i = 0; i < Hi; ++i)
    for (int j = 0; j < Wi; ++j)
    for (int ky = 0; ky < K; ++ky) {
        const int oy = i * S + ky - p;
        if (oy < 0 || oy >= Ho) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int ox = j * S + kx - p;
            if (ox < 0 || ox >= Wo) continue;
            for (int co = 0; co < Cout; ++co) {
                const double g = dy[IDX4(n, oy, ox, co, Ho, Wo, Cout)];
                for (int ci = 0; ci < Cin; ++ci) {
                    const size_t wi = (((size_t)ky * K + kx) * Cout + co) * Cin + ci;
                    const size_t xi = IDX4(n, i, j, ci, Hi, Wi, Cin);
                    ax[xi] += g * w[wi];
                    aw[wi] += g * x[xi];
                }
            }
        }
    }
    if (dx) for (size_t q = 0; q < nx; ++q) dx[q] = (float)ax[q];
    if (dw) for (size_t q = 0; q < nw; ++q) dw[q] = (float)aw[q];
    free(ax); free(aw);
}

/* K10: reduce_mean(softmax_cross_entropy_with_logits(labels, logits)),
 * fcn8s_tensorflow.py:253, with class-id labels standing for the one-hot rows.
 * dlogits = (softmax - onehot) / npix.  Returns the mean loss. */
double orc_softmax_xent(const float* logits, const uint8_t* label_ids, float* dlogits,
                        size_t npix, int C)
{
    double total = 0;
    for (size_t p = 0; p < npix; ++p) {
        const float* l = logits + p * C;
        double m = l[0];
        for (int c = 1; c < C; ++c) if (l[c] > m) m = l[c];
        double s = 0;
        for (int c = 0; c < C; ++c) s += exp((double)l[c] - m);
        const double lse = m + log(s);
        total += lse - l[label_ids[p]];
        if (dlogits)
            for (int c = 0; c < C; ++c)
                dlogits[p * C + c] = (float)((exp((double)l[c] - lse) - (c == label_ids[p])) / (double)npix);
    }
    return total / (double)npix;
}

/* K13: tf.nn.softmax then tf.argmax(axis=-1, int64), fcn8s_tensorflow.py:268-269.
 * float32 softmax as TF/Eigen computes it (exp(x-max)/sum); lowest index wins ties. */
void orc_softmax_argmax(const float* logits, float* softmax_out, int64_t* argmax_out,
                        size_t npix, int C)
{
    for (size_t p = 0; p < npix; ++p) {
        const float* l = logits + p * C;
        float m = l[0];
        for (int c = 1; c < C; ++c) if (l[c] > m) m = l[c];
        float e[256]; float s = 0.f;
        for (int c = 0; c < C; ++c) { e[c] = expf(l[c] - m); s += e[c]; }
        int best = 0; float bv = -1.f;
        for (int c = 0; c < C; ++c) {
            const float v = e[c] / s;
            if (softmax_out) softmax_out[p * C + c] = v;
            if (v > bv) { bv = v; best = c; }
        }
        if (argmax_out) argmax_out[p] = best;
    }
}

/* K14: the accumulator inside tf.metrics.mean_iou (fcn8s_tensorflow.py:291-293);
 * identical arithmetic to cityscapesscripts/evaluation/addToConfusionMatrix_impl.c:10-16. */
void orc_confusion(const int64_t* label_ids, const int64_t* pred_ids, size_t npix,
                   int64_t* conf, int C)
{
    for (size_t p = 0; p < npix; ++p) conf[(size_t)C * label_ids[p] + pred_ids[p]] += 1;
}

/* K12: tf.train.AdamOptimizer (fcn8s_tensorflow.py:256): epsilon outside the
 * bias correction.  t = 1-based step.  float32 state like TF. */
void orc_tf_adam(float* theta, const float* g, float* m, float* v, size_t n, int t,
                 float lr, float beta1, float beta2, float eps)
{
    const float lr_t = lr * (float)sqrt(1.0 - pow((double)beta2, t)) / (float)(1.0 - pow((double)beta1, t));
    for (size_t i = 0; i < n; ++i) {
        m[i] = beta1 * m[i] + (1.f - beta1) * g[i];
        v[i] = beta2 * v[i] + (1.f - beta2) * g[i] * g[i];
        theta[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
    }
}
