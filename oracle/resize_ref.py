"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the image preprocessing in front of the hot path:

  * ``test.py:207-216``  read_resize_image: cv2.imread -> BGR2GRAY -> ``tw = int(128 * (w / h))`` ->
    ``cv2.resize(src, (tw, 128), interpolation=cv2.INTER_AREA)``
  * ``utils/dataset.py:47-60``  ImageDataset.pil_loader: ``new_width = int(width * (128 / height))`` ->
    ``cv2.resize(img, (new_width, 128), interpolation=cv2.INTER_AREA)``
  * ``utils/dataset.py:118-132``  AlignCollate width cap (max_width=1600 as built at ``test.py:235``)

PARITY UNPINNED. The algorithm lives in a third-party dependency that is absent from /root/reference and
not installed here: ``opencv-python`` (``requirements.txt:5``, no version pin). This file restates the
published OpenCV 4.x algorithm (modules/imgproc/src/resize.cpp: ``resize`` dispatch, ``computeResizeAreaTab``,
``ResizeArea_Invoker``, ``ResizeAreaFast_Invoker``, the ``area_mode`` branch of the linear resizer with its
11-bit fixed-point ``HResizeLinear`` / ``VResizeLinear<uchar>``; modules/imgproc/src/color_yuv: ``RGB2Gray<uchar>``
with 14-bit coefficients). The reference ships no resized fixtures, so nothing here is checked against cv2
output; the known-answer tests are hand-computed from the published formulas. Known build-dependent detail:
for an exact 2x2 decimation cv2's SIMD body rounds half up ((s+2)>>2) while its scalar tail rounds half to
even; the vector form is used for every pixel here.

The HIP kernel (csrc/preprocess.hip) is tested bit-exact against this file.
"""
import math

import numpy as np

COEF_BITS = 11                      # INTER_RESIZE_COEF_BITS
COEF_SCALE = 1 << COEF_BITS
B2Y, G2Y, R2Y, GRAY_SHIFT = 1868, 9617, 4899, 14


def bgr2gray(img, order="bgr"):
    """u8 [H,W,3] -> u8 [H,W]; cv2.cvtColor(src, COLOR_BGR2GRAY) (test.py:209-210)."""
    a = img.astype(np.int64)
    if order == "bgr":
        b, g, r = a[..., 0], a[..., 1], a[..., 2]
    else:
        r, g, b = a[..., 0], a[..., 1], a[..., 2]
    return ((b * B2Y + g * G2Y + r * R2Y + (1 << (GRAY_SHIFT - 1))) >> GRAY_SHIFT).astype(np.uint8)


def target_width(h, w, height=128, rule="test"):
    """rule 'test': test.py:211-213; rule 'dataset': utils/dataset.py:54-56."""
    if rule == "test":
        return int(height * (float(w) / float(h)))
    return int(w * (height / h))


def _round_u8(x32):
    """saturate_cast<uchar>(float): cvRound (round half to even), then clamp."""
    return np.clip(np.rint(x32.astype(np.float64)), 0, 255).astype(np.uint8)


def area_tab(ssize, dsize, scale):
    """computeResizeAreaTab -> per destination index the ordered (source index, float32 weight) list."""
    tab = [[] for _ in range(dsize)]
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1 = math.ceil(f1)
        s2 = math.floor(f2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            tab[d].append((s1 - 1, np.float32((s1 - f1) / cell)))
        for s in range(s1, s2):
            tab[d].append((s, np.float32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            tab[d].append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
    return tab


def _resize_area(src, dw, dh, scale_x, scale_y):
    sh, sw = src.shape
    xt = area_tab(sw, dw, scale_x)
    yt = area_tab(sh, dh, scale_y)
    s32 = src.astype(np.float32)
    buf = np.zeros((sh, dw), np.float32)                  # horizontal pass of every source row
    for d in range(dw):
        acc = np.zeros((sh,), np.float32)
        for s, a in xt[d]:
            acc = acc + s32[:, s] * a                      # float32 multiply, then float32 add (no FMA)
        buf[:, d] = acc
    out = np.zeros((dh, dw), np.uint8)
    for d in range(dh):
        acc = np.zeros((dw,), np.float32)
        for s, b in yt[d]:
            acc = acc + b * buf[s]
        out[d] = _round_u8(acc)
    return out


def _resize_area_fast(src, dw, dh, ix, iy):
    sh, sw = src.shape
    out = np.zeros((dh, dw), np.uint8)
    area = ix * iy
    scale = np.float32(1.0) / np.float32(area)
    for dy in range(dh):
        sy0 = dy * iy
        if sy0 >= sh:
            continue
        for dx in range(dw):
            sx0 = dx * ix
            if sy0 + iy <= sh and sx0 + ix <= sw:
                s = int(src[sy0:sy0 + iy, sx0:sx0 + ix].astype(np.int64).sum())
                if ix == 2 and iy == 2:
                    out[dy, dx] = (s + 2) >> 2
                else:
                    out[dy, dx] = _round_u8(np.float32(s) * scale)
            elif sx0 < sw:
                blk = src[sy0:min(sy0 + iy, sh), sx0:min(sx0 + ix, sw)].astype(np.int64)
                out[dy, dx] = _round_u8(np.float32(np.float32(blk.sum()) / np.float32(blk.size)))
    return out


def linear_area_coeffs(ssize, dsize, scale, inv_scale, clamp_high):
    """area_mode branch of the linear resizer: source index, and the two 11-bit fixed-point weights."""
    ofs = np.zeros(dsize, np.int64)
    c0 = np.zeros(dsize, np.int64)
    c1 = np.zeros(dsize, np.int64)
    for d in range(dsize):
        s = math.floor(d * scale)
        f = np.float32((d + 1) - (s + 1) * inv_scale)
        f = np.float32(0) if f <= 0 else np.float32(f - np.float32(math.floor(f)))
        if clamp_high and s >= ssize - 1:
            f = np.float32(0)
            s = ssize - 1
        ofs[d] = s
        c0[d] = int(np.clip(np.rint(np.float64((np.float32(1) - f) * np.float32(COEF_SCALE))), -32768, 32767))
        c1[d] = int(np.clip(np.rint(np.float64(f * np.float32(COEF_SCALE))), -32768, 32767))
    return ofs, c0, c1


def _resize_linear_area(src, dw, dh, scale_x, scale_y, inv_x, inv_y):
    sh, sw = src.shape
    xo, a0, a1 = linear_area_coeffs(sw, dw, scale_x, inv_x, True)
    yo, b0, b1 = linear_area_coeffs(sh, dh, scale_y, inv_y, False)
    s = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    rows = s[:, xo] * a0[None, :] + s[:, x1] * a1[None, :]       # HResizeLinear, int32 range
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    r0 = rows[y0] >> 4
    r1 = rows[y1] >> 4
    v = (((b0[:, None] * r0) >> 16) + ((b1[:, None] * r1) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)                            # uchar(...) cast (values are 0..255 here)


def resize_area(src, dw, dh):
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA) for one-channel u8 images."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    if dw <= 0 or dh <= 0:
        raise ValueError("empty destination (cv2 asserts !dsize.empty())")
    inv_x = dw / sw
    inv_y = dh / sh
    scale_x = 1.0 / inv_x
    scale_y = 1.0 / inv_y
    if scale_x >= 1 and scale_y >= 1:
        ixr, iyr = int(np.rint(scale_x)), int(np.rint(scale_y))      # saturate_cast<int>(double) rounds
        eps = np.finfo(np.float64).eps
        if abs(scale_x - ixr) < eps and abs(scale_y - iyr) < eps:
            return _resize_area_fast(src, dw, dh, ixr, iyr)
        return _resize_area(src, dw, dh, scale_x, scale_y)
    return _resize_linear_area(src, dw, dh, scale_x, scale_y, inv_x, inv_y)


def read_resize(img, height=128, rule="test", order="bgr"):
    """test.py:207-216 / utils/dataset.py:47-60 on an already decoded u8 array ([H,W] or [H,W,3])."""
    if img.ndim == 3:
        img = bgr2gray(img, order)
    h, w = img.shape
    return resize_area(img, target_width(h, w, height, rule), height)


def align_collate_widths(widths, max_width=1600):
    """utils/dataset.py:118-124: batch width = min(max line width, max_width)."""
    m = max(widths)
    if max_width and m > max_width:
        m = max_width
    return m


def truncate_label(label, w, maxw):
    """utils/dataset.py:139-143: a line wider than the cap keeps a proportional prefix of its label."""
    if w > maxw:
        return label[:max(1, int(len(label) * (maxw / w)))]
    return label
