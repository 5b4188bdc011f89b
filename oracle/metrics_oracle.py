"""metrics_oracle.py -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's objective metrics core
(src/gui_objective_metrics.py): ``_psnr_bgr`` 438-444, ``_ssim_single_channel`` / ``_ssim_bgr`` 447-474,
``_linear_bgr_to_absolute_rgb`` 477-483, ``_pq_oetf_absolute`` 486-491, ``_linear_rgb_to_itp`` 494-515 and
``_delta_e_itp_absolute_rgb`` 518-528, on unit-range float images ``[3][H][W]`` in R, G, B plane order.

PARITY UNPINNED: the module imports cv2 at load, so it cannot be run here, and its SSIM uses ``cv2.GaussianBlur``
(OpenCV, not in the reference tree).  PSNR and dE-ITP are plain NumPy formulas restated from the text; the blur is
OpenCV's documented one -- ``getGaussianKernel(11, 1.5)`` (exp(-x^2 / 2 sigma^2), normalised), separable, float32,
BORDER_REFLECT_101.  Checked by known answers only."""
from __future__ import annotations

import numpy as np

PQ_M1, PQ_M2 = 2610.0 / 16384.0, 2523.0 / 32.0
PQ_C1, PQ_C2, PQ_C3 = 3424.0 / 4096.0, 2413.0 / 128.0, 2392.0 / 128.0


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float32) - b.astype(np.float32)) ** 2, dtype=np.float64))
    return 99.0 if mse <= 1e-12 else float(10.0 * np.log10(1.0 / mse))


def gaussian_kernel(ksize=11, sigma=1.5):
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _blur(img, k):
    r = len(k) // 2
    p = np.pad(img.astype(np.float32), r, mode="reflect")        # numpy "reflect" = BORDER_REFLECT_101
    h, w = img.shape
    tmp = np.zeros((h + 2 * r, w), np.float32)
    for i, kv in enumerate(k):
        tmp += p[:, i:i + w] * kv
    out = np.zeros((h, w), np.float32)
    for i, kv in enumerate(k):
        out += tmp[i:i + h] * kv
    return out


def ssim(a, b):
    k = gaussian_kernel()
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    vals = []
    for c in range(3):
        x, y = a[c].astype(np.float32), b[c].astype(np.float32)
        mu_a, mu_b = _blur(x, k), _blur(y, k)
        sa = _blur(x * x, k) - mu_a * mu_a
        sb = _blur(y * y, k) - mu_b * mu_b
        sab = _blur(x * y, k) - mu_a * mu_b
        num = (2.0 * mu_a * mu_b + c1) * (2.0 * sab + c2)
        den = (mu_a * mu_a + mu_b * mu_b + c1) * (sa + sb + c2)
        vals.append(float(np.mean(num / (den + 1e-12), dtype=np.float64)))
    return float(np.mean(vals))


def pq_oetf_absolute(lum):
    y = np.clip(lum.astype(np.float32) / 10000.0, 0.0, 1.0)
    y_m1 = np.power(y, PQ_M1).astype(np.float32)
    num = PQ_C1 + PQ_C2 * y_m1
    den = 1.0 + PQ_C3 * y_m1
    return np.power(num / np.maximum(den, 1e-12), PQ_M2).astype(np.float32)


def itp(rgb_abs):
    r, g, b = (rgb_abs[i].astype(np.float32) for i in range(3))
    l = (1688.0 * r + 2146.0 * g + 262.0 * b) / 4096.0
    m = (683.0 * r + 2951.0 * g + 462.0 * b) / 4096.0
    s = (99.0 * r + 309.0 * g + 3688.0 * b) / 4096.0
    lp, mp, sp = pq_oetf_absolute(l), pq_oetf_absolute(m), pq_oetf_absolute(s)
    i = 0.5 * lp + 0.5 * mp
    ct = (6610.0 * lp - 13613.0 * mp + 7003.0 * sp) / 4096.0
    cp = (17933.0 * lp - 17390.0 * mp - 543.0 * sp) / 4096.0
    return i, 0.5 * ct, cp


def delta_e_itp(a, b, peak_nits=1000.0):
    pa = np.clip(a.astype(np.float32), 0.0, 1.0) * float(peak_nits)
    pb = np.clip(b.astype(np.float32), 0.0, 1.0) * float(peak_nits)
    i1, t1, p1 = itp(pa)
    i2, t2, p2 = itp(pb)
    de = 720.0 * np.sqrt((i1 - i2) ** 2 + (t1 - t2) ** 2 + (p1 - p2) ** 2 + 1e-12)
    return float(np.mean(de, dtype=np.float64))


def metrics(a, b, peak_nits=1000.0):
    """-> dict(psnr_db, sssim, delta_e_itp): the keys of the reference's metrics dict (runtime_metrics.py:192-194)."""
    return {"psnr_db": psnr(a, b), "sssim": ssim(a, b), "delta_e_itp": delta_e_itp(a, b, peak_nits)}
