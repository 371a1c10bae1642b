"""The OpenCV calls of the reference's BatchGenerator, restated in NumPy with OpenCV's own integer / float32 arithmetic.

`data_generator/batch_generator.py` goes through `cv2` for every pixel it resamples or recolours:

    cv2.resize(image, dsize, interpolation=cv2.INTER_LINEAR)        :329, :367   (images: resize, random scale)
    cv2.resize(gt_image, dsize, interpolation=cv2.INTER_NEAREST)    :330, :377   (ground truth)
    cv2.cvtColor(image, cv2.COLOR_RGB2HSV) / COLOR_HSV2RGB          :474, :486   (_brightness)
    cv2.flip(image, 1)                                              :341
    cv2.warpAffine(image, [[1,0,x],[0,1,y]], dsize)                 :355         (integer translation)
    cv2.cvtColor(image, cv2.COLOR_RGB2GRAY)                         :387

OpenCV (module `opencv-python`, version unpinned by the reference -- no requirements file; the code dates from 2017/18, i.e. the
3.x series) is a third-party dependency that is neither vendored in /root/reference nor installable here, so these functions
restate its published algorithms for 8-bit images and are pinned by hand-derived known answers and by an independent
scalar-loop restatement (tests/golden/make_cv2_vectors.py -> tests/golden/cv2_vectors.npz):

  * resize INTER_NEAREST (imgproc/src/resize.cpp, resizeNN): sx = min(floor(x * (1 / (dw / sw))), sw - 1), same for rows;
  * resize INTER_LINEAR, 8-bit (resize.cpp, resizeGeneric_ / HResizeLinear / VResizeLinear<uchar,int,short,FixedPtCast<..,22>>):
    fx = float((dx + 0.5) * scale - 0.5) in double, sx = floor(fx), fx -= sx, clamped at the borders; the two taps are rounded to
    11-bit fixed point (cvRound(w * 2048), saturated to short); horizontal pass in int32, vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  An exact 2x shrink in both directions is INTER_AREA in
    disguise (resize.cpp: "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to INTER_LINEAR"):
    (a + b + c + d + 2) >> 2;
  * RGB2HSV, 8-bit, H in [0,180) (imgproc/src/color_hsv: RGB2HSV_b): integer arithmetic with the 12-bit reciprocal tables
    sdiv_table[v] = cvRound((255 << 12) / v), hdiv_table180[d] = cvRound((180 << 12) / (6 d));
  * HSV2RGB, 8-bit (HSV2RGB_b): float32 -- h * (6/180), sector = floor(h), tab = {v, v(1-s), v(1-s h), v(1-s(1-h))}, outputs
    cvRound(x * 255) saturated (cvRound = round half to even);
  * RGB2GRAY, 8-bit (imgproc/src/color_yuv / color.cpp, RGB2Gray<uchar>, 3.x constants): (4899 R + 9617 G + 1868 B + 8192) >> 14
    (OpenCV >= 4.1 uses the 15-bit constants 9798 / 19235 / 3735; the two differ by at most 1 in rare pixels);
  * flip and integer-translation warpAffine move whole pixels (weights 1 and 0): plain copies.

The GPU kernels fcn8s_op_resample_u8 / fcn8s_op_augment_u8 follow the same definitions (csrc/elementwise.hip) and are tested
bit-exact against this module.
"""
from __future__ import annotations

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


# ---- cv2.resize -------------------------------------------------------------------------------------------------------------------

def nearest_indices(src_size, dst_size):
    """resizeNN: source index of every destination index."""
    inv_scale = float(dst_size) / float(src_size)          # double
    ifx = 1.0 / inv_scale
    idx = np.floor(np.arange(dst_size, dtype=np.float64) * ifx).astype(np.int64)
    return np.minimum(idx, src_size - 1)


def resize_nearest(a, height, width):
    """cv2.resize(a, (width, height), interpolation=cv2.INTER_NEAREST) for any dtype / channel count."""
    a = np.asarray(a)
    return np.ascontiguousarray(a[nearest_indices(a.shape[0], height)][:, nearest_indices(a.shape[1], width)])


def linear_taps(src_size, dst_size):
    """Per destination index: (source index s, 11-bit fixed-point weights of s and s + 1) exactly as resize() tabulates them."""
    scale = 1.0 / (float(dst_size) / float(src_size))       # double: scale_x = 1. / inv_scale_x
    d = np.arange(dst_size, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)        # fx = (float)((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0; s[lo] = 0
    hi = s >= src_size - 1
    f[hi] = 0.0; s[hi] = src_size - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)      # saturate_cast<short>(float) = cvRound
    w1 = np.rint(f * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)
    return s, np.clip(w0, -32768, 32767), np.clip(w1, -32768, 32767)


def resize_linear(a, height, width):
    """cv2.resize(a, (width, height), interpolation=cv2.INTER_LINEAR) for uint8 images [H,W] or [H,W,C]."""
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError("resize_linear restates OpenCV's 8-bit path; got %s" % a.dtype)
    H, W = a.shape[:2]
    if height == H and width == W:
        return a.copy()
    if H == 2 * height and W == 2 * width:                   # INTER_AREA (fast) stands in for INTER_LINEAR at an exact 2x shrink
        x = a.astype(np.int64)
        return ((x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, ax0, ax1 = linear_taps(W, width)
    sy, by0, by1 = linear_taps_rows(H, height)
    x = a.astype(np.int64)
    sx1 = np.minimum(sx + 1, W - 1)                          # (weight 0 wherever sx + 1 would leave the row)
    shape = (1, width) + (1,) * (a.ndim - 2)
    rows = x[:, sx] * ax0.reshape(shape) + x[:, sx1] * ax1.reshape(shape)        # horizontal pass, int32 range
    r0 = np.clip(sy, 0, H - 1); r1 = np.clip(sy + 1, 0, H - 1)
    bshape = (height,) + (1,) * (a.ndim - 1)
    out = (((by0.reshape(bshape) * (rows[r0] >> 4)) >> 16) + ((by1.reshape(bshape) * (rows[r1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)                              # (the result of the fixed-point cast always lies in 0..255)


def linear_taps_rows(src_size, dst_size):
    """Rows: the same table, but the source index is not clamped when it is built (the row loop clamps the two row numbers
    instead and keeps the weights)."""
    scale = 1.0 / (float(dst_size) / float(src_size))
    d = np.arange(dst_size, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)
    w1 = np.rint(f * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)
    return s, w0, w1


# ---- cv2.cvtColor ---------------------------------------------------------------------------------------------------------------------

_HSV_SHIFT = 12
_i = np.arange(1, 256, dtype=np.float64)
SDIV_TABLE = np.concatenate([[0], np.rint((255 << _HSV_SHIFT) / (1.0 * _i))]).astype(np.int64)
HDIV_TABLE180 = np.concatenate([[0], np.rint((180 << _HSV_SHIFT) / (6.0 * _i))]).astype(np.int64)
del _i


def rgb2hsv(image):
    """cv2.cvtColor(image, cv2.COLOR_RGB2HSV) for uint8 [..., 3]: H in [0, 180), S and V in [0, 255]."""
    a = np.asarray(image)
    r, g, b = (a[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * SDIV_TABLE[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * HDIV_TABLE180[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])      # tab index of (b, g, r) per sector


def hsv2rgb(hsv):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2RGB) for uint8 [..., 3] (H in [0, 180))."""
    a = np.asarray(hsv)
    f = np.float32
    h = a[..., 0].astype(f)
    s = a[..., 1].astype(f) * f(1.0 / 255.0)
    v = a[..., 2].astype(f) * f(1.0 / 255.0)
    h = h * f(6.0 / 180.0)
    h = np.where(h < 0, h + f(6), h).astype(f)               # (uint8 input: h lies in [0, 8.5]; one wrap suffices)
    h = np.where(h >= 6, h - f(6), h).astype(f)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector); h = np.where(bad, f(0), h).astype(f)
    one = f(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * h), v * (one - s * (one - h))], -1).astype(f)
    idx = _SECTOR[sector]                                    # [..., 3] = tab indices of b, g, r
    b = np.take_along_axis(tab, idx[..., 0:1], -1)[..., 0]
    g = np.take_along_axis(tab, idx[..., 1:2], -1)[..., 0]
    r = np.take_along_axis(tab, idx[..., 2:3], -1)[..., 0]
    grey = s == 0
    b = np.where(grey, v, b); g = np.where(grey, v, g); r = np.where(grey, v, r)
    out = np.stack([r, g, b], -1).astype(f) * f(255.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)    # saturate_cast<uchar>(float): cvRound (half to even), saturated


def brightness(image, factor):
    """`_brightness` of the reference (data_generator/batch_generator.py:469-486) for a given factor: RGB -> 8-bit HSV, V * factor in
    float64 saturated at 255 and TRUNCATED on the store into the uint8 array (`hsv[:,:,2] = v_channel`), HSV -> RGB."""
    hsv = rgb2hsv(image)
    v = hsv[..., 2] * float(factor)
    hsv[..., 2] = np.where(v > 255, 255, v).astype(np.uint8)
    return hsv2rgb(hsv)


def rgb2gray(image):
    """cv2.cvtColor(image, cv2.COLOR_RGB2GRAY) for uint8 [..., 3] (OpenCV 3.x 14-bit constants)."""
    a = np.asarray(image).astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)


# ---- whole-pixel moves ----------------------------------------------------------------------------------------------------------------

def flip_horizontal(a):
    """cv2.flip(a, 1)"""
    return np.ascontiguousarray(np.asarray(a)[:, ::-1])


def translate(a, x_shift, y_shift, border_value=0):
    """cv2.warpAffine(a, [[1,0,x_shift],[0,1,y_shift]], (W, H), borderValue=border_value) for integer shifts:
    dst(x, y) = src(x - x_shift, y - y_shift), `border_value` where that falls outside."""
    a = np.asarray(a)
    out = np.full_like(a, 0 if border_value is None else border_value)
    h, w = a.shape[:2]
    if abs(y_shift) < h and abs(x_shift) < w:
        ys, yd = (slice(0, h - y_shift), slice(y_shift, h)) if y_shift >= 0 else (slice(-y_shift, h), slice(0, h + y_shift))
        xs, xd = (slice(0, w - x_shift), slice(x_shift, w)) if x_shift >= 0 else (slice(-x_shift, w), slice(0, w + x_shift))
        out[yd, xd] = a[ys, xs]
    return out
