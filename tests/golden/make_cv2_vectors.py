#!/usr/bin/env python3
"""Known-answer vectors for the OpenCV calls of the reference's BatchGenerator (data_generator/batch_generator.py:328-331, :341, :355,
:367, :377, :387, :469-486).  OpenCV is not installable here and not vendored by the reference, so the expected outputs are produced by a
SCALAR, pixel-by-pixel transcription of OpenCV's published 8-bit algorithms (one Python statement per C++ statement of resize.cpp /
color_hsv / color_yuv), deliberately written without NumPy vector tricks and without looking at fcn8s_tensorflow_amd/cv2_compat.py's
formulation -- two independent restatements have to agree -- plus a handful of values derived by hand (listed in HAND below and
asserted here, so that a slip in the transcription itself is caught).

    python tests/golden/make_cv2_vectors.py        -> tests/golden/cv2_vectors.npz
"""
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def f32(x):
    """round a Python float (double) to float32 and back, like a C `float` variable"""
    return struct.unpack("f", struct.pack("f", x))[0]


def cv_round(x):
    """cvRound / saturate_cast from floating point: nearest, ties to even (lrint under the default rounding mode)"""
    fl = math.floor(x)
    d = x - fl
    if d > 0.5 or (d == 0.5 and fl % 2 == 1):
        return int(fl) + 1
    return int(fl)


def sat_short(x):
    return max(-32768, min(32767, cv_round(x)))


def sat_u8(x):
    return max(0, min(255, cv_round(x)))


# ---- resize.cpp -----------------------------------------------------------------------------------------------------------------------
def resize_nn(src, dh, dw):
    sh, sw = len(src), len(src[0])
    fx, fy = dw / sw, dh / sh                       # inv_scale_x, inv_scale_y
    ifx, ify = 1.0 / fx, 1.0 / fy
    x_ofs = [min(int(math.floor(x * ifx)), sw - 1) for x in range(dw)]
    out = []
    for y in range(dh):
        sy = min(int(math.floor(y * ify)), sh - 1)
        out.append([src[sy][x_ofs[x]] for x in range(dw)])
    return out


def resize_linear(src, dh, dw):
    """src: list of rows of pixels, a pixel = list of channel values"""
    sh, sw, cn = len(src), len(src[0]), len(src[0][0])
    inv_scale_x, inv_scale_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_scale_x, 1.0 / inv_scale_y
    iscale_x, iscale_y = cv_round(scale_x), cv_round(scale_y)
    is_area_fast = abs(scale_x - iscale_x) < 2.220446049250313e-16 and abs(scale_y - iscale_y) < 2.220446049250313e-16
    if is_area_fast and iscale_x == 2 and iscale_y == 2:          # INTER_LINEAR -> INTER_AREA (fast), 2x2 boxes
        return [[[(src[2 * y][2 * x][c] + src[2 * y][2 * x + 1][c] + src[2 * y + 1][2 * x][c] + src[2 * y + 1][2 * x + 1][c] + 2) >> 2
                  for c in range(cn)] for x in range(dw)] for y in range(dh)]
    xofs, ialpha = [], []
    for dx in range(dw):
        fx = f32((dx + 0.5) * scale_x - 0.5)
        sx = int(math.floor(fx))
        fx = f32(fx - sx)
        if sx < 0:
            fx, sx = 0.0, 0
        if sx >= sw - 1:
            fx, sx = 0.0, sw - 1
        xofs.append(sx)
        ialpha.append((sat_short(f32(f32(1.0 - fx) * 2048.0)), sat_short(f32(fx * 2048.0))))
    yofs, ibeta = [], []
    for dy in range(dh):
        fy = f32((dy + 0.5) * scale_y - 0.5)
        sy = int(math.floor(fy))
        fy = f32(fy - sy)
        yofs.append(sy)
        ibeta.append((sat_short(f32(f32(1.0 - fy) * 2048.0)), sat_short(f32(fy * 2048.0))))

    def hrow(r):                                              # HResizeLinear on source row r -> int32 buffer
        S = src[r]
        row = []
        for dx in range(dw):
            sx = xofs[dx]
            a0, a1 = ialpha[dx]
            if sx + 1 < sw:
                row.append([S[sx][c] * a0 + S[sx + 1][c] * a1 for c in range(cn)])
            else:
                row.append([S[sx][c] * 2048 for c in range(cn)])
        return row

    def clip(x, a, b):
        return (x if x < b else b - 1) if x >= a else a
    out = []
    for dy in range(dh):
        S0, S1 = hrow(clip(yofs[dy], 0, sh)), hrow(clip(yofs[dy] + 1, 0, sh))
        b0, b1 = ibeta[dy]
        out.append([[(((b0 * (S0[x][c] >> 4)) >> 16) + ((b1 * (S1[x][c] >> 4)) >> 16) + 2) >> 2 for c in range(cn)] for x in range(dw)])
    return out


# ---- color_hsv ------------------------------------------------------------------------------------------------------------------------
HSV_SHIFT = 12
SDIV = [0] + [cv_round((255 << HSV_SHIFT) / (1.0 * i)) for i in range(1, 256)]
HDIV180 = [0] + [cv_round((180 << HSV_SHIFT) / (6.0 * i)) for i in range(1, 256)]


def rgb2hsv_px(r, g, b):
    v = max(b, g, r)
    vmin = min(b, g, r)
    diff = v - vmin
    vr = -1 if v == r else 0
    vg = -1 if v == g else 0
    s = (diff * SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * HDIV180[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h += 180 if h < 0 else 0
    return max(0, min(255, h)), s, v


SECTOR = [[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]]


def hsv2rgb_px(H, S, V):
    h, s, v = float(H), f32(S * f32(1.0 / 255.0)), f32(V * f32(1.0 / 255.0))
    if s == 0:
        b = g = r = v
    else:
        h = f32(h * f32(6.0 / 180.0))
        while h < 0:
            h = f32(h + 6)
        while h >= 6:
            h = f32(h - 6)
        sector = int(math.floor(h))
        h = f32(h - sector)
        if not 0 <= sector < 6:
            sector, h = 0, 0.0
        tab = [v, f32(v * f32(1.0 - s)), f32(v * f32(1.0 - f32(s * h))), f32(v * f32(1.0 - f32(s * f32(1.0 - h))))]
        b, g, r = tab[SECTOR[sector][0]], tab[SECTOR[sector][1]], tab[SECTOR[sector][2]]
    return sat_u8(f32(r * 255.0)), sat_u8(f32(g * 255.0)), sat_u8(f32(b * 255.0))


def brightness_px(r, g, b, factor):
    h, s, v = rgb2hsv_px(r, g, b)
    vv = v * factor
    v = 255 if vv > 255 else int(vv)                       # float64 -> uint8 store truncates
    return hsv2rgb_px(h, s, v)


def gray_px(r, g, b):
    return (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14


# values derived by hand (DESIGN.md section 2 walks through two of them)
HAND = {
    "hsv(255,0,0)": (rgb2hsv_px(255, 0, 0), (0, 255, 255)),
    "hsv(0,255,0)": (rgb2hsv_px(0, 255, 0), (60, 255, 255)),
    "hsv(0,0,255)": (rgb2hsv_px(0, 0, 255), (120, 255, 255)),
    "hsv(128,64,32)": (rgb2hsv_px(128, 64, 32), (10, 191, 128)),       # s = (96*8160+2048)>>12, h = (32*1280+2048)>>12
    "hsv(7,7,7)": (rgb2hsv_px(7, 7, 7), (0, 0, 7)),
    "rgb(60,255,255)": (hsv2rgb_px(60, 255, 255), (0, 255, 0)),
    "rgb(0,0,200)": (hsv2rgb_px(0, 0, 200), (200, 200, 200)),
    "gray(255,255,255)": (gray_px(255, 255, 255), 255),
    "gray(255,0,0)": (gray_px(255, 0, 0), 76),                          # (255*4899+8192)>>14 = 76
    "linear [0,100] -> 4": ([p[0] for p in resize_linear([[[0], [100]]], 1, 4)[0]], [0, 25, 75, 100]),
    "linear 2x shrink": (resize_linear([[[1], [2], [10], [20]], [[3], [4], [30], [41]]], 1, 2), [[[3], [25]]]),    # (1+2+3+4+2)>>2, (101+2)>>2
    "nearest 5 -> 3": (resize_nn([[0, 1, 2, 3, 4]], 1, 3)[0], [0, 1, 3]),                                          # floor(x * 5/3)
    "nearest 3 -> 7": (resize_nn([[0, 1, 2]], 1, 7)[0], [0, 0, 0, 1, 1, 2, 2]),
}


def main():
    for k, (got, want) in HAND.items():
        assert got == want, (k, got, want)
    rng = np.random.default_rng(2017)
    out = {}
    img = rng.integers(0, 256, (11, 14, 3), dtype=np.uint8)
    lab = rng.integers(0, 34, (11, 14), dtype=np.uint8)
    out["img"], out["lab"] = img, lab
    sizes = [(7, 9), (23, 30), (11, 28), (5, 7), (16, 14), (11, 14)]
    out["sizes"] = np.array(sizes)
    for i, (h, w) in enumerate(sizes):
        out["linear_%d" % i] = np.array(resize_linear(img.tolist(), h, w), dtype=np.uint8)
        out["nearest_%d" % i] = np.array(resize_nn(lab.tolist(), h, w), dtype=np.uint8)
    img2 = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)          # exact 2x shrink: the INTER_AREA branch
    out["img2"] = img2
    out["linear_half"] = np.array(resize_linear(img2.tolist(), 6, 8), dtype=np.uint8)
    out["linear_half_x_only"] = np.array(resize_linear(img2.tolist(), 12, 8), dtype=np.uint8)     # only one axis halves: stays bilinear
    # colour: every grey, the colour cube corners / edges, and random pixels
    px = np.concatenate([np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1),
                         np.array([[r, g, b] for r in (0, 1, 127, 128, 254, 255) for g in (0, 1, 127, 128, 254, 255) for b in (0, 1, 127, 128, 254, 255)], np.uint8),
                         rng.integers(0, 256, (1500, 3), dtype=np.uint8)])
    out["px"] = px
    out["hsv"] = np.array([rgb2hsv_px(*map(int, p)) for p in px], dtype=np.uint8)
    hs = np.concatenate([out["hsv"], np.stack([rng.integers(0, 180, 800), rng.integers(0, 256, 800), rng.integers(0, 256, 800)], 1).astype(np.uint8)])
    out["hsv_in"] = hs
    out["rgb_from_hsv"] = np.array([hsv2rgb_px(*map(int, p)) for p in hs], dtype=np.uint8)
    factors = [0.5, 0.73, 1.0, 1.31, 2.0]
    out["factors"] = np.array(factors)
    for i, f in enumerate(factors):
        out["bright_%d" % i] = np.array([brightness_px(*map(int, p), f) for p in px], dtype=np.uint8)
    out["gray"] = np.array([gray_px(*map(int, p)) for p in px], dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "cv2_vectors.npz"), **out)
    print("wrote cv2_vectors.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
