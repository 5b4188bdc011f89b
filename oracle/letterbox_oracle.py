"""letterbox_oracle.py -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the host step in front of
``preprocess``: ``_letterbox_bgr`` (src/gui_scaling.py:228-244) = aspect-preserving ``cv2.resize`` to fit
(out_w, out_h) -- INTER_AREA when shrinking, INTER_CUBIC when enlarging -- pasted centred on a black canvas.

PARITY UNPINNED.  The arithmetic lives in OpenCV (``opencv-python``, unpinned in
requirements/requirements-common.txt, not under /root/reference and not installed here), so this file restates
OpenCV 4.x's published 8-bit algorithms and is checked by known answers only:
  * INTER_AREA, integer ratio: box sums; 2x2 is ``(a+b+c+d+2)>>2``, other ratios ``cvRound(sum * float(1/area))``
    (imgproc/src/resize.cpp ResizeAreaFastVec / ResizeAreaFast_Invoker);
  * INTER_AREA, fractional ratio: ``computeResizeAreaTab`` weights (float), row buffers accumulated in float in
    table order, ``cvRound`` at the end (ResizeArea_Invoker);
  * INTER_CUBIC: Keys kernel A = -0.75, coefficients quantised to 11 fractional bits (``cvRound(c * 2048)``),
    horizontal then vertical integer passes with replicated borders, ``(v + 2^21) >> 22`` (the scalar
    FixedPtCast path; OpenCV's SIMD path rounds in float and may differ from it by one LSB on ties).
cvRound is round-half-to-even.  Geometry uses Python's round(), as the reference does.
"""
from __future__ import annotations

import math

import numpy as np

AREA, CUBIC = 0, 1


def geometry(w, h, out_w, out_h):
    """-> (new_w, new_h, x0, y0, interp) of gui_scaling.py:234-243."""
    scale = min(out_w / max(w, 1), out_h / max(h, 1))
    new_w = max(1, int(round(w * scale)))
    new_h = max(1, int(round(h * scale)))
    return new_w, new_h, (out_w - new_w) // 2, (out_h - new_h) // 2, (AREA if scale < 1.0 else CUBIC)


def area_tab(ssize, dsize):
    """computeResizeAreaTab: list of (dst index, src index, float32 alpha)."""
    scale = ssize / dsize                      # double
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _cv_round(a):
    return np.rint(a)                           # half to even, like lrint


def resize_area(src, new_w, new_h):
    h, w = src.shape[:2]
    sx, sy = w / new_w, h / new_h
    ix, iy = int(round(sx)), int(round(sy))
    if abs(sx - ix) < np.finfo(np.float64).eps and abs(sy - iy) < np.finfo(np.float64).eps:
        blk = src[: new_h * iy, : new_w * ix].reshape(new_h, iy, new_w, ix, -1).astype(np.int64).sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            return ((blk + 2) >> 2).astype(np.uint8)
        scale = np.float32(1.0) / np.float32(ix * iy)
        return np.clip(_cv_round(blk.astype(np.float32) * scale), 0, 255).astype(np.uint8)
    xt, yt = area_tab(w, new_w), area_tab(h, new_h)
    s = src.astype(np.float32)
    # horizontal: buf[sy][dx] = sum_k S[sy][sx_k] * alpha_k, float32, accumulated in table order from 0
    buf = np.zeros((h, new_w, src.shape[2]), np.float32)
    for dx, sxk, alpha in xt:
        buf[:, dx] = buf[:, dx] + s[:, sxk] * alpha
    out = np.zeros((new_h, new_w, src.shape[2]), np.float32)
    first = [True] * new_h
    for dy, syk, beta in yt:
        if first[dy]:
            out[dy] = beta * buf[syk]
            first[dy] = False
        else:
            out[dy] = out[dy] + beta * buf[syk]
    return np.clip(_cv_round(out), 0, 255).astype(np.uint8)


def _cubic_coeffs(x):
    A = np.float32(-0.75)
    x = np.float32(x)
    one = np.float32(1.0)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return [c0, c1, c2, c3]


def cubic_tab(ssize, dsize):
    """-> (src start index sx-1 per dst index, int16 coefficients [dsize][4])."""
    scale = ssize / dsize
    ofs = np.zeros(dsize, np.int64)
    co = np.zeros((dsize, 4), np.int64)
    for d in range(dsize):
        fx = np.float32((d + 0.5) * scale - 0.5)
        s0 = math.floor(fx)
        fx = np.float32(fx - np.float32(s0))
        ofs[d] = s0 - 1
        co[d] = [int(np.clip(_cv_round(np.float32(c) * np.float32(2048)), -32768, 32767)) for c in _cubic_coeffs(fx)]
    return ofs, co


def resize_cubic(src, new_w, new_h):
    h, w = src.shape[:2]
    xo, xc = cubic_tab(w, new_w)
    yo, yc = cubic_tab(h, new_h)
    s = src.astype(np.int64)
    hor = np.zeros((h, new_w, src.shape[2]), np.int64)
    for k in range(4):
        idx = np.clip(xo + k, 0, w - 1)
        hor += s[:, idx] * xc[:, k][None, :, None]
    out = np.zeros((new_h, new_w, src.shape[2]), np.int64)
    for k in range(4):
        idy = np.clip(yo + k, 0, h - 1)
        out += hor[idy] * yc[:, k][:, None, None]
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def letterbox_bgr(frame, out_w, out_h):
    h, w = frame.shape[:2]
    if w == out_w and h == out_h:
        return frame
    new_w, new_h, x0, y0, interp = geometry(w, h, out_w, out_h)
    if (new_w, new_h) == (w, h):
        resized = frame
    else:
        resized = resize_area(frame, new_w, new_h) if interp == AREA else resize_cubic(frame, new_w, new_h)
    canvas = np.zeros((out_h, out_w, 3), dtype=frame.dtype)
    canvas[y0:y0 + new_h, x0:x0 + new_w] = resized
    return canvas
