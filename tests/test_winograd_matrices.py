"""The Winograd transform matrices compiled into csrc/winograd.hip are exactly the ones tools/winograd_matrices.py derives
(Cook-Toom, checked symbolically against the correlation they must compute) -- a CPU-side pin of the constants the fast
convolution path rests on."""
import os
import re
import sys
from fractions import Fraction

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _parse_struct(src, m, r):
    """-> {'bt': rows, 'g': rows, 'at': rows} of Fractions from `template <> struct WinoMat<m, r> { ... }`."""
    start = src.index("template <> struct WinoMat<%d, %d>" % (m, r))
    body = src[start:src.index("\n};\n", start)]
    out = {}
    for name in ("bt", "g", "at"):
        i = body.index(" %s(int i, int j)" % name)
        j = body.index("constexpr float m", i)
        k = body.index(";", j)
        lit = body[body.index("=", j) + 1:k]
        rows = re.findall(r"\{([^{}]*)\}", lit)
        mat = []
        for row in rows:
            vals = []
            for tok in row.split(","):
                tok = tok.strip().replace(".f", "").replace("f", "")
                if "/" in tok:
                    a, b = tok.split("/")
                    vals.append(Fraction(a.strip()) / Fraction(b.strip()))
                else:
                    vals.append(Fraction(tok))
            mat.append(vals)
        out[name] = mat
    return out


@pytest.mark.parametrize("m,r,points", [(4, 3, [0, 1, -1, 2, -2, None]),
                                        (6, 3, [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2), None]),
                                        (4, 4, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), -2, None])])
def test_kernel_constants_are_the_derived_matrices(m, r, points):
    import sympy as sp
    from winograd_matrices import check
    pts = [None if p is None else sp.Rational(p.numerator, p.denominator) if isinstance(p, Fraction) else sp.Integer(p) for p in points]
    AT, G, BT = check(m, r, pts)                       # raises if A^T [(G g) . (B^T d)] is not the correlation of d with g
    src = open(os.path.join(ROOT, "fcn8s_tensorflow_amd", "csrc", "winograd.hip")).read()
    k = _parse_struct(src, m, r)

    def same_up_to_row_scaling(kernel_bt, kernel_g, BTd, Gd):
        # B^T rows may be scaled by s_i if G's rows are scaled by 1/s_i (the published F(4,3) matrices use another scaling)
        for i in range(len(kernel_bt)):
            ref = [Fraction(int(x.p), int(x.q)) for x in BTd.row(i)]
            nz = next(j for j, v in enumerate(ref) if v != 0)
            s = kernel_bt[i][nz] / ref[nz]
            assert s != 0 and [v * s for v in ref] == kernel_bt[i], ("bt", i)
            refg = [Fraction(int(x.p), int(x.q)) for x in Gd.row(i)]
            assert [v / s for v in refg] == kernel_g[i], ("g", i)
    same_up_to_row_scaling(k["bt"], k["g"], BT, G)
    assert k["at"] == [[Fraction(int(x.p), int(x.q)) for x in AT.row(i)] for i in range(AT.rows)]
