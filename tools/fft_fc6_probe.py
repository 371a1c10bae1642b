#!/usr/bin/env python
"""Probe for a next-round idea (not in the product): fc6 (7x7, 512 -> 4096 on a 16x32 map) through 16x16 real FFT tiles (10x10 outputs
per tile) with the three-multiplication complex product, instead of F(4x4,4x4) Winograd sub-filters.  Prints the multiply counts of both
algorithms at the training shape and the fp32 error of the FFT route against a float64 direct correlation (numpy keeps complex64 for
float32 input), next to the F(4x4,4x4) figure of tools/winograd_matrices.py (1e-5 of the output range)."""
import numpy as np


def conv_direct64(x, w):
    H, W, Ci = x.shape; K = w.shape[0]; Co = w.shape[3]; p = K // 2
    xp = np.zeros((H + 2 * p, W + 2 * p, Ci)); xp[p:p + H, p:p + W] = x
    y = np.zeros((H, W, Co))
    for ky in range(K):
        for kx in range(K):
            y += xp[ky:ky + H, kx:kx + W] @ w[ky, kx]
    return y


def conv_fft32(x, w, tile=16):
    H, W, Ci = x.shape; K = w.shape[0]; Co = w.shape[3]; p = K // 2; m = tile - K + 1
    th, tw = -(-H // m), -(-W // m)
    xp = np.zeros((th * m + K - 1, tw * m + K - 1, Ci), np.float32); xp[p:p + H, p:p + W] = x
    wf = np.zeros((tile, tile, Ci, Co), np.float32); wf[:K, :K] = w[::-1, ::-1]          # correlation = convolution with the flipped kernel
    Wf = np.fft.rfft2(wf, axes=(0, 1))                                                   # complex64 [tile][tile/2+1][Ci][Co]
    c, d = Wf.real, Wf.imag
    y = np.zeros((th * m, tw * m, Co), np.float32)
    for ty in range(th):
        for tx in range(tw):
            X = np.fft.rfft2(xp[ty * m:ty * m + tile, tx * m:tx * m + tile], axes=(0, 1))
            a, b = X.real, X.imag
            # Gauss: three real GEMMs per frequency
            k1 = np.einsum('uvi,uvio->uvo', (a + b), c); k2 = np.einsum('uvi,uvio->uvo', a, (d - c)); k3 = np.einsum('uvi,uvio->uvo', b, (c + d))
            Y = np.fft.irfft2((k1 - k3) + 1j * (k1 + k2), s=(tile, tile), axes=(0, 1))
            y[ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = Y[K - 1:, K - 1:]
    return y[:H, :W]


def main():
    rng = np.random.default_rng(0)
    H, W, Ci, Co = 16, 32, 256, 16
    x = np.maximum(rng.standard_normal((H, W, Ci)), 0).astype(np.float32)        # ReLU-like input
    w = (rng.standard_normal((7, 7, Ci, Co)) * np.sqrt(2.0 / (49 * Ci))).astype(np.float32)
    ref = conv_direct64(x.astype(np.float64), w.astype(np.float64))
    got = conv_fft32(x, w)
    print("FFT route (complex64), Cin = %d: max |error| / output range = %.2e   (F(4x4,4x4) Winograd at K = 512: 1e-5)" % (Ci, np.abs(got - ref).max() / np.abs(ref).max()))
    N, Cin, Cout = 16, 512, 4096
    wino = 49 * 4 * (N * 4 * 8)                      # positions x sub-filters x tiles
    fft = 144 * 3 * (N * 2 * 4)                      # complex frequencies x 3 real products x tiles
    print("real multiplies per (cin, cout) pair at 16 x 1024x512: Winograd F(4x4,4x4) sub-filters %d, 16x16 FFT tiles + 3-mult %d  (x%.2f)" % (wino, fft, wino / fft))
    print("GEMM flops per pass: %.0f vs %.0f GFLOP; filter bank %.2f vs %.2f GB; GEMM rows per position %d vs %d"
          % (2e-9 * wino * Cin * Cout, 2e-9 * fft * Cin * Cout, 49 * 4 * Cin * Cout * 4e-9, 144 * 3 * Cin * Cout * 4e-9, N * 32, N * 8))


if __name__ == "__main__":
    main()
