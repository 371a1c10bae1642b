#!/usr/bin/env python3
"""Derives (and checks symbolically) the Cook-Toom / Winograd matrices A^T, G, B^T used in csrc/winograd.hip:

    Y = A^T [ (G g G^T) . (B^T d B) ] A        F(m x m, r x r), alpha = m + r - 1 interpolation points (one of them at infinity)

With V_k the alpha x k Vandermonde matrix of the points (row [0 .. 0 1] for infinity):  A^T = V_m^T,  G = V_r,  B^T = V_alpha^-T;
the rows of B^T are scaled to integers and G's rows by the inverse factors.  `numerics()` measures the fp32 error of the 2-D
algorithm against a float64 direct correlation on ReLU-like data (what decided F(6,3) and F(4,4) were usable in fp32).

    python tools/winograd_matrices.py
"""
import numpy as np, sympy as sp
from fractions import Fraction

def vand(points, k):
    rows = []
    for p in points:
        if p is None: rows.append([0] * (k - 1) + [1])
        else: rows.append([sp.Rational(p) ** i for i in range(k)])
    return sp.Matrix(rows)

def winograd(m, r, points):
    al = m + r - 1
    assert len(points) == al
    AT = vand(points, m).T
    G = vand(points, r)
    BT = vand(points, al).inv().T
    # scale rows of BT to integers, G rows inversely
    for i in range(al):
        den = sp.ilcm(*[sp.fraction(x)[1] for x in BT.row(i)])
        num = sp.igcd(*[sp.fraction(x * den)[0] for x in BT.row(i) if x != 0])
        sc = sp.Rational(den, num)
        BT[i, :] = BT.row(i) * sc
        G[i, :] = G.row(i) / sc
    return AT, G, BT

def check(m, r, pts):
    AT, G, BT = winograd(m, r, pts)
    d = sp.Matrix(sp.symbols('d0:%d' % (m + r - 1))); g = sp.Matrix(sp.symbols('g0:%d' % r))
    y = AT * sp.Matrix([a * b for a, b in zip(G * g, BT * d)])
    ref = sp.Matrix([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert sp.simplify(y - ref) == sp.zeros(m, 1)
    return AT, G, BT

if __name__ == "__main__":
    import sys
    for (m, r, pts) in [(4, 3, [0, 1, -1, 2, -2, None]), (4, 4, [0, 1, -1, 2, -2, sp.Rational(1, 2), None]), (4, 4, [0, 1, -1, sp.Rational(1,2), sp.Rational(-1,2), 2, None]),
                        (4, 4, [0, 1, -1, sp.Rational(1,2), sp.Rational(-1,2), -2, None])]:
        AT, G, BT = check(m, r, pts)
        print("F(%d,%d) points" % (m, r), pts)
        print("AT", AT.tolist()); print("G", G.tolist()); print("BT", BT.tolist())
        # numerics: 2-D, K channels, fp32 vs fp64
        rng = np.random.default_rng(0)
        A_ = np.array(AT.tolist(), dtype=np.float64); G_ = np.array(G.tolist(), dtype=np.float64); B_ = np.array(BT.tolist(), dtype=np.float64)
        K = 2048
        d = np.maximum(rng.standard_normal((K, m + r - 1, m + r - 1)), 0)       # relu-like data
        g = rng.standard_normal((K, r, r)) / np.sqrt(K * r * r)
        ref = np.zeros((m, m))
        for i in range(m):
            for j in range(m):
                ref[i, j] = (d[:, i:i + r, j:j + r] * g).sum()
        def run(dt):
            A, Gm, B = A_.astype(dt), G_.astype(dt), B_.astype(dt)
            U = np.einsum('ak,ckl,bl->cab', Gm, g.astype(dt), Gm).astype(dt)
            V = np.einsum('ak,ckl,bl->cab', B, d.astype(dt), B).astype(dt)
            M = (U * V).astype(dt).sum(0, dtype=dt)
            return A @ M @ A.T
        e32 = np.abs(run(np.float32) - ref).max() / np.abs(ref).max()
        e64 = np.abs(run(np.float64) - ref).max() / np.abs(ref).max()
        print("  rel err fp32 %.2e   fp64 %.2e" % (e32, e64))

def numerics(m, r, pts, K, trials=3):
    AT, G, BT = check(m, r, pts)
    A_ = np.array(AT.tolist(), dtype=np.float64); G_ = np.array(G.tolist(), dtype=np.float64); B_ = np.array(BT.tolist(), dtype=np.float64)
    errs = []
    for t in range(trials):
        rng = np.random.default_rng(t)
        n = m + r - 1
        d = np.maximum(rng.standard_normal((K, n, n)), 0); g = rng.standard_normal((K, r, r)) / np.sqrt(K * r * r)
        ref = np.array([[(d[:, i:i + r, j:j + r] * g).sum() for j in range(m)] for i in range(m)])
        A, Gm, B = A_.astype(np.float32), G_.astype(np.float32), B_.astype(np.float32)
        U = np.einsum('ak,ckl,bl->cab', Gm, g.astype(np.float32), Gm).astype(np.float32)
        V = np.einsum('ak,ckl,bl->cab', B, d.astype(np.float32), B).astype(np.float32)
        M = (U * V).astype(np.float32).sum(0, dtype=np.float32)
        errs.append(np.abs(A @ M @ A.T - ref).max() / np.abs(ref).max())
    return max(errs)
