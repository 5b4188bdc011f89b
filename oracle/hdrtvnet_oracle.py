"""hdrtvnet_oracle.py -- TEST INFRASTRUCTURE ONLY (the parity oracle), never shipped.

CPU fp32 restatement of the reference's SDR->HDR per-frame path: the network
graphs of AGCM, LE and HG composed from the plain-C operators in
``hdrtv_oracle.c`` (loaded with ctypes), plus the pre/post quantisers.  Each
function cites the reference lines it follows.  PARITY PINNED: every function
is checked against golden vectors produced by running the reference itself
(``tests/golden/gen_golden.py``; see ``tests/test_oracle_golden.py``) -- except
``post_pq_rgb48`` / ``gamut709_2020``, which have no implementation on the
reference's playback path and are "parity unpinned"
(known-answer tests only).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (hdr-realtime-video-pipeline_amd/)
never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
F32P = ctypes.POINTER(ctypes.c_float)


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hdrtv_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    lib = ctypes.CDLL(so)
    lib.orc_pq_oetf.restype = ctypes.c_float
    lib.orc_pq_oetf.argtypes = [ctypes.c_float]
    _LIB = lib
    return lib


def set_threads(n: int) -> None:
    """OpenMP thread count for the C operators (cpu_baseline reports it as ``cores``)."""
    omp = ctypes.CDLL("libgomp.so.1")
    omp.omp_set_num_threads(int(n))


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(F32P)


# ------------------------------------------------------------- INT8 fake-quant (configs[4])
class QWeight(np.ndarray):
    """A dequantised weight that remembers its layer's activation quantiser (W8A8 layers only)."""
    x_scale = None
    x_zero = None


def _fq(x, w):
    """Activation fake-quant in front of a W8A8 layer, fp32 as the reference's CPU path computes it
    (W8A8Conv2d.forward / W8A8Linear.forward, hdrtvnet_torch.py:351-364, 398-409): asymmetric
    ``round((x - zero) / scale).clamp(0, 255) * scale + zero``, symmetric ``round(x / scale).clamp(-128, 127) * scale``;
    round = half to even.  Every other weight passes x through."""
    sc = getattr(w, "x_scale", None)
    if sc is None:
        return x
    x = np.asarray(x, np.float32)
    sc = np.float32(sc)
    if w.x_zero is not None:
        z = np.float32(w.x_zero)
        q = np.clip(np.rint((x - z) / sc), np.float32(0), np.float32(255))
        return (q * sc + z).astype(np.float32)
    q = np.clip(np.rint(x / sc), np.float32(-128), np.float32(127))
    return (q * sc).astype(np.float32)


def w8a8_state(state):
    """INT8 runtime checkpoint -> a state dict the graphs below run as the reference's fake-quant execution
    (``predequantize`` off): ``<layer>.weight`` = weight_int8 * scale in fp32 (W8*.forward), and for W8A8 layers the
    returned weight carries x_scale / x_zero so that conv2d / linear quantise their input first."""
    out = {}
    for k, v in state.items():
        a = np.asarray(v)
        if k.endswith(".weight_int8"):
            base = k[: -len(".weight_int8")]
            w_scale = state.get(base + ".w_scale", state.get(base + ".scale"))
            shape = (-1,) + (1,) * (a.ndim - 1)
            w = (a.astype(np.float32) * np.asarray(w_scale, np.float32).reshape(shape)).astype(np.float32).view(QWeight)
            if base + ".x_scale" in state:
                w.x_scale = float(np.asarray(state[base + ".x_scale"], np.float32).reshape(-1)[0])
                if base + ".x_zero" in state:
                    w.x_zero = float(np.asarray(state[base + ".x_zero"], np.float32).reshape(-1)[0])
            out[base + ".weight"] = w
        elif k.endswith((".w_scale", ".scale", ".x_scale", ".x_zero")):
            continue
        else:
            out[k] = a.astype(np.float32) if a.dtype.kind == "f" else a
    return out


# --------------------------------------------------------------------------- ops
def conv2d(x, w, b=None, stride=1, pad=0):
    x, xp = _f(_fq(x, w))
    w, wp = _f(w)
    co, ci, k, _ = w.shape
    assert x.shape[0] == ci, (x.shape, w.shape)
    h, wd = x.shape[1:]
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    y = np.empty((co, ho, wo), np.float32)
    if b is not None:
        b, bp = _f(b)
    else:
        bp = None
    _lib().orc_conv2d(xp, ci, h, wd, wp, bp, co, k, stride, pad, y.ctypes.data_as(F32P))
    return y


def linear(v, w, b):
    return (np.asarray(w, np.float32) @ np.asarray(_fq(v, w), np.float32) + np.asarray(b, np.float32)).astype(np.float32)


def avgpool3s2p1(x):
    x, xp = _f(x)
    c, h, w = x.shape
    y = np.empty((c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), np.float32)
    _lib().orc_avgpool3s2p1(xp, c, h, w, y.ctypes.data_as(F32P))
    return y


def instnorm(x, gamma, beta, eps=1e-5):
    x, xp = _f(x)
    g, gp = _f(gamma)
    b, bp = _f(beta)
    y = np.empty_like(x)
    _lib().orc_instnorm(xp, x.shape[0], x.shape[1] * x.shape[2], gp, bp, ctypes.c_float(eps),
                        y.ctypes.data_as(F32P))
    return y


def batchnorm(x, gamma, beta, mean, var, eps=1e-5):
    x, xp = _f(x)
    g, gp = _f(gamma)
    b, bp = _f(beta)
    m, mp = _f(mean)
    v, vp = _f(var)
    y = np.empty_like(x)
    _lib().orc_batchnorm(xp, x.shape[0], x.shape[1] * x.shape[2], gp, bp, mp, vp, ctypes.c_float(eps),
                         y.ctypes.data_as(F32P))
    return y


def maxpool2(x):
    x, xp = _f(x)
    c, h, w = x.shape
    y = np.empty((c, h // 2, w // 2), np.float32)
    _lib().orc_maxpool2(xp, c, h, w, y.ctypes.data_as(F32P))
    return y


def pixelshuffle2(x):
    x, xp = _f(x)
    c, h, w = x.shape
    y = np.empty((c // 4, 2 * h, 2 * w), np.float32)
    _lib().orc_pixelshuffle2(xp, c, h, w, y.ctypes.data_as(F32P))
    return y


def relu(x):
    return np.maximum(x, np.float32(0))


def leaky(x, slope):
    return np.where(x >= 0, x, x * np.float32(slope)).astype(np.float32)


def bicubic_aa_quarter(x):
    """hdrtvnet_torch.py:2278-2285 (F.interpolate 0.25x bicubic antialias)."""
    x, xp = _f(x)
    c, h, w = x.shape
    y = np.empty((c, max(1, h // 4), max(1, w // 4)), np.float32)
    _lib().orc_bicubic_aa_quarter(xp, c, h, w, y.ctypes.data_as(F32P))
    return y


def bilinear_quarter(x):
    """hdrtvnet_torch.py:2269-2276 (``fast_condition_resize``): F.interpolate(0.25, bilinear, align_corners=False,
    recompute_scale_factor=False).  Source coordinate 4 d + 1.5: both lambdas are 0.5, the taps are (4d+1, 4d+2) clamped to
    the last row / column; arithmetic order of ATen's upsample_bilinear2d."""
    x = np.asarray(x, np.float32)
    c, h, w = x.shape
    ho, wo = max(1, h // 4), max(1, w // 4)
    y1 = 4 * np.arange(ho) + 1
    x1 = 4 * np.arange(wo) + 1
    y2 = np.minimum(y1 + 1, h - 1)
    x2 = np.minimum(x1 + 1, w - 1)
    half = np.float32(0.5)
    top = half * x[:, y1][:, :, x1] + half * x[:, y1][:, :, x2]
    bot = half * x[:, y2][:, :, x1] + half * x[:, y2][:, :, x2]
    return (half * top + half * bot).astype(np.float32)


_C_OPS = {k: globals()[k] for k in ("conv2d", "avgpool3s2p1", "instnorm", "batchnorm", "maxpool2", "pixelshuffle2",
                                    "bicubic_aa_quarter", "relu", "leaky")}


def use_backend(name: str) -> None:
    """"c": the plain-C operators above (default).  "aten": PyTorch's CPU kernels (oracle/aten_backend.py) under the
    same graphs, i.e. the network as the reference runs it eagerly on a CPU."""
    if name == "c":
        globals().update(_C_OPS)
    elif name == "aten":
        from . import aten_backend
        globals().update(aten_backend.OPS)
        aten_conv = aten_backend.OPS["conv2d"]
        globals()["conv2d"] = lambda x, w, b=None, stride=1, pad=0: aten_conv(_fq(x, w), w, b, stride, pad)
    else:
        raise ValueError(name)


# --------------------------------------------------------------- pre / post stages
def preprocess(frame_bgr):
    """HDRTVNetTorch.preprocess, hdrtvnet_torch.py:2238-2296 -> (tensor[3,H,W], cond[3,H//4,W//4])."""
    f = np.ascontiguousarray(frame_bgr, dtype=np.uint8)
    h, w = f.shape[:2]
    t = np.empty((3, h, w), np.float32)
    _lib().orc_pre_unpack(f.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), h, w, t.ctypes.data_as(F32P))
    return t, bicubic_aa_quarter(t)


def postprocess_u8(out_chw):
    """HDRTVNetTorch.postprocess, hdrtvnet_torch.py:2351-2368 -> u8 [H,W,3] BGR."""
    x, xp = _f(out_chw)
    _, h, w = x.shape
    y = np.empty((h, w, 3), np.uint8)
    _lib().orc_post_u8(xp, h, w, y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return y


def post_rgb48(out_chw):
    """_tensor_to_rgb48_bytes, gui_pipeline_worker_feeders.py:193-229 -> u16 [H,W,3] RGB."""
    x, xp = _f(out_chw)
    _, h, w = x.shape
    y = np.empty((h, w, 3), np.uint16)
    _lib().orc_post_rgb48(xp, h, w, y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    return y


def pq_oetf(nits: float) -> float:
    """gui_objective_metrics.py:486-491."""
    return float(_lib().orc_pq_oetf(ctypes.c_float(nits)))


def gamut709_2020(rgb_chw):
    x, xp = _f(rgb_chw)
    y = np.empty_like(x)
    _lib().orc_gamut709_2020(xp, x.shape[1], x.shape[2], y.ctypes.data_as(F32P))
    return y


def post_pq_rgb48(rgb_chw, peak_nits=1000.0):
    x, xp = _f(rgb_chw)
    _, h, w = x.shape
    y = np.empty((h, w, 3), np.uint16)
    _lib().orc_post_pq_rgb48(xp, h, w, ctypes.c_float(peak_nits), y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    return y


# ------------------------------------------------------------------------- AGCM
def agcm_classifier(sd, cond):
    """Color_Condition.forward, Condition_arch.py:19-35 -> 6-vector."""
    x = cond
    for i, idx in enumerate((0, 4, 8, 12, 16)):
        p = f"AGCM.classifier.model.{idx}"
        x = conv2d(x, sd[p + ".weight"], sd[p + ".bias"])
        x = avgpool3s2p1(x)
        x = leaky(x, 0.2)
        if i < 4:
            q = f"AGCM.classifier.model.{idx + 3}"
            x = instnorm(x, sd[q + ".weight"], sd[q + ".bias"], 1e-5)
    x = conv2d(x, sd["AGCM.classifier.model.20.weight"], sd["AGCM.classifier.model.20.bias"])
    return x.mean(axis=(1, 2), dtype=np.float64).astype(np.float32)


def agcm(sd, tensor, cond, taps=None):
    """ConditionNet.forward dynamic branch, Condition_arch.py:559-585 -> agcm_out[3,H,W]."""
    fea = agcm_classifier(sd, cond)
    if taps is not None:
        taps["fea6"] = fea

    def gfm(stage):
        s = linear(fea, sd[f"AGCM.cond_scale_{stage}.weight"], sd[f"AGCM.cond_scale_{stage}.bias"])
        t = linear(fea, sd[f"AGCM.cond_shift_{stage}.weight"], sd[f"AGCM.cond_shift_{stage}.bias"])
        return s[:, None, None], t[:, None, None]

    out = conv2d(tensor, sd["AGCM.conv_first.weight"], sd["AGCM.conv_first.bias"])
    s, t = gfm("first")
    out = relu(out * s + t + out)
    out = conv2d(out, sd["AGCM.HRconv.weight"], sd["AGCM.HRconv.bias"])
    s, t = gfm("HR")
    out = relu(out * s + t + out)
    out = conv2d(out, sd["AGCM.conv_last.weight"], sd["AGCM.conv_last.bias"])
    s, t = gfm("last")
    return (out * s + t + out).astype(np.float32)


# --------------------------------------------------------------------------- LE
def _c(sd, name, x, stride=1, pad=None):
    w = sd[name + ".weight"]
    if pad is None:
        pad = w.shape[2] // 2
    return conv2d(x, w, sd[name + ".bias"], stride, pad)


def sft(sd, name, x, cond):
    """SFTLayer.forward, arch_util.py:68-72."""
    scale = _c(sd, f"{name}.SFT_scale_conv1", leaky(_c(sd, f"{name}.SFT_scale_conv0", cond), 0.1))
    shift = _c(sd, f"{name}.SFT_shift_conv1", leaky(_c(sd, f"{name}.SFT_shift_conv0", cond), 0.1))
    return (x * (scale + np.float32(1)) + shift).astype(np.float32)


def resblock_sft(sd, name, x, cond):
    """ResBlock_with_SFT.forward, arch_util.py:89-95."""
    fea = sft(sd, name + ".sft1", x, cond)
    fea = relu(_c(sd, name + ".conv1", fea))
    fea = sft(sd, name + ".sft2", fea, cond)
    fea = _c(sd, name + ".conv2", fea)
    return x + fea


def align_to(x, ref_hw):
    """HDRUNet3T1._align_to, HDRUNet3T1_arch.py:79-104: centre-crop then replicate-pad."""
    rh, rw = ref_hw
    xh, xw = x.shape[-2:]
    if xh > rh:
        top = (xh - rh) // 2
        x = x[..., top:top + rh, :]
    if xw > rw:
        left = (xw - rw) // 2
        x = x[..., :, left:left + rw]
    xh, xw = x.shape[-2:]
    ph, pw = rh - xh, rw - xw
    if ph > 0 or pw > 0:
        pt, pl = ph // 2, pw // 2
        x = np.pad(x, ((0, 0), (pt, ph - pt), (pl, pw - pl)), mode="edge")
    return np.ascontiguousarray(x)


def le(sd, img, taps=None):
    """HDRUNet3T1._forward_safe_aligned, HDRUNet3T1_arch.py:152-206, with
    x = [img, img] (Ensemble_AGCM_LE_arch.py:890-893) and weighting_network=False."""
    def tap(k, v):
        if taps is not None:
            taps[k] = v
        return v

    L = "LE."
    cond = _c(sd, L + "cond_first.0", img)
    cond = leaky(cond, 0.1)
    cond = leaky(_c(sd, L + "cond_first.2", cond), 0.1)
    cond = tap("LE.cond_first", leaky(_c(sd, L + "cond_first.4", cond), 0.1))

    def condnet(i, strides):
        n = f"{L}CondNet{i}"
        y = leaky(_c(sd, n + ".0", cond, strides[0]), 0.1)
        y = leaky(_c(sd, n + ".2", y, strides[1]), 0.1)
        return tap(f"LE.CondNet{i}", _c(sd, n + ".4", y, strides[2]))

    cond1 = condnet(1, (1, 1, 1))
    cond2 = condnet(2, (2, 1, 1))
    cond3 = condnet(3, (2, 2, 1))
    cond4 = condnet(4, (2, 2, 2))

    fea0 = relu(tap("LE.conv_first", _c(sd, L + "conv_first", img)))
    fea0 = tap("LE.SFT_layer1", sft(sd, L + "SFT_layer1", fea0, cond1))
    fea0 = relu(tap("LE.HR_conv1", _c(sd, L + "HR_conv1", fea0)))

    fea1 = relu(tap("LE.down_conv1", _c(sd, L + "down_conv1", fea0, 2)))
    fea1 = tap("LE.recon_trunk1", resblock_sft(sd, L + "recon_trunk1.0", fea1, cond2))
    fea2 = relu(tap("LE.down_conv2", _c(sd, L + "down_conv2", fea1, 2)))
    fea2 = tap("LE.recon_trunk2", resblock_sft(sd, L + "recon_trunk2.0", fea2, cond3))
    fea3 = relu(tap("LE.down_conv3", _c(sd, L + "down_conv3", fea2, 2)))
    out = fea3
    for b in range(4):
        out = resblock_sft(sd, f"{L}recon_trunk3.{b}", out, cond4)
    tap("LE.recon_trunk3", out)
    out = out + fea3

    up = relu(tap("LE.up_conv1", pixelshuffle2(_c(sd, L + "up_conv1.0", out))))
    if up.shape[-2:] != fea2.shape[-2:]:
        up = align_to(up, fea2.shape[-2:])
    out = tap("LE.recon_trunk4", resblock_sft(sd, L + "recon_trunk4.0", up + fea2, cond3))

    up = relu(tap("LE.up_conv2", pixelshuffle2(_c(sd, L + "up_conv2.0", out))))
    if up.shape[-2:] != fea1.shape[-2:]:
        up = align_to(up, fea1.shape[-2:])
    out = tap("LE.recon_trunk5", resblock_sft(sd, L + "recon_trunk5.0", up + fea1, cond2))

    up = relu(tap("LE.up_conv3", pixelshuffle2(_c(sd, L + "up_conv3.0", out))))
    if up.shape[-2:] != fea0.shape[-2:]:
        up = align_to(up, fea0.shape[-2:])
    out = tap("LE.SFT_layer2", sft(sd, L + "SFT_layer2", up + fea0, cond1))
    out = relu(tap("LE.HR_conv2", _c(sd, L + "HR_conv2", out)))
    out = tap("LE.conv_last", _c(sd, L + "conv_last", out))
    if out.shape[-2:] != img.shape[-2:]:
        out = align_to(out, img.shape[-2:])
    return (img + out).astype(np.float32)


def hr_forward(sd, tensor, cond, taps=None):
    """Ensemble_AGCM_LE.forward, Ensemble_AGCM_LE_arch.py:889-897 -> (out, agcm_out)."""
    a = agcm(sd, tensor, cond, taps)
    return le(sd, a, taps), a


# --------------------------------------------------------------------------- HG
def hg_mask(base, r=0.75, thresh=0.1):  # noqa: D401  (r = HG_Composite(mask_r))
    """HG_Composite._make_mask, HG_Composite_arch.py:78-84."""
    m = base.max(axis=0, keepdims=True)
    m = ((m - np.float32(r)) / np.float32(1.0 - r)).astype(np.float32)
    m = np.clip(m, 0.0, 1.0)
    return (m > np.float32(thresh)).astype(np.float32)


def _hg_block(hg, name, x):
    """conv_block: conv3x3 + BatchNorm2d(eval) + ReLU, Hallucination_arch.py:24-29."""
    y = conv2d(x, hg[name + ".0.weight"], hg[name + ".0.bias"], 1, 1)
    y = batchnorm(y, hg[name + ".1.weight"], hg[name + ".1.bias"], hg[name + ".1.running_mean"],
                  hg[name + ".1.running_var"], 1e-5)
    return relu(y)


def _hg_up(hg, name, x):
    """up_block: conv3x3 -> PixelShuffle(2) -> ReLU, Hallucination_arch.py:32-36."""
    return relu(pixelshuffle2(conv2d(x, hg[name + ".0.weight"], hg[name + ".0.bias"], 1, 1)))


def hg_generator(hg, img, mask, taps=None):
    """Hallucination_Generator.forward, Hallucination_arch.py:97-137."""
    def tap(k, v):
        if taps is not None:
            taps[k] = v
        return v

    c1 = tap("hg.conv1", _hg_block(hg, "conv1", img))
    c2 = tap("hg.conv2", _hg_block(hg, "conv2", tap("hg.p1", maxpool2(c1))))
    c3 = tap("hg.conv3_2", _hg_block(hg, "conv3_2", maxpool2(_hg_block(hg, "conv3_1", c2))))
    c4 = tap("hg.conv4_2", _hg_block(hg, "conv4_2", maxpool2(_hg_block(hg, "conv4_1", c3))))
    c5 = tap("hg.conv5_2", _hg_block(hg, "conv5_2", maxpool2(_hg_block(hg, "conv5_1", c4))))
    code = tap("hg.conv_code2", _hg_block(hg, "conv_code2", maxpool2(_hg_block(hg, "conv_code1", c5))))

    def fuse(name, a, b):
        return conv2d(np.concatenate((a, b), axis=0), hg[name + ".weight"], hg[name + ".bias"])

    c6 = tap("hg.conv6", fuse("conv6", _hg_up(hg, "Up_conv1", code), c5))
    c7 = tap("hg.conv7", fuse("conv7", _hg_up(hg, "Up_conv2", c6), c4))
    c8 = tap("hg.conv8", fuse("conv8", _hg_up(hg, "Up_conv3", c7), c3))
    c9 = tap("hg.conv9", fuse("conv9", _hg_up(hg, "Up_conv4", c8), c2))
    up5 = tap("hg.up5", _hg_up(hg, "Up_conv5", c9))
    c10 = tap("hg.conv10", fuse("conv10", up5, c1))
    out = tap("hg.tail", fuse("conv_last", c10, img))          # before the mask blend: lets a test re-blend with any mask
    return (mask * out + img).astype(np.float32)


def hg_composite(sd, hg, tensor, cond, taps=None):
    """HG_Composite.forward, HG_Composite_arch.py:86-107 -> (hg_out, agcm_out)."""
    base, a = hr_forward(sd, tensor, cond, taps)
    if taps is not None:
        taps["base"] = base
    mask = hg_mask(base)
    _, h, w = base.shape
    ph, pw = (32 - h % 32) % 32, (32 - w % 32) % 32
    if ph or pw:
        bp = np.pad(base, ((0, 0), (0, ph), (0, pw)), mode="reflect")
        mp = np.pad(mask, ((0, 0), (0, ph), (0, pw)), mode="reflect")
        out = hg_generator(hg, bp, mp, taps)[:, :h, :w]
    else:
        out = hg_generator(hg, base, mask, taps)
    if taps is not None:
        taps["mask"] = mask
    return np.ascontiguousarray(out), a


def process(sd, frame_bgr, hg=None):
    """HDRTVNetTorch.process, hdrtvnet_torch.py:2373-2377 -> u8 BGR (CPU fp32 semantics)."""
    t, c = preprocess(frame_bgr)
    out = hg_composite(sd, hg, t, c)[0] if hg is not None else hr_forward(sd, t, c)[0]
    return postprocess_u8(out)
