"""Weight pack (``.hdrw``) writer/reader and the seeded HG weight generator.

The C-ABI library (``include/hdrtv_mi355x.h``: ``hdrtv_create``) takes one flat
blob; this module turns a reference ``state_dict`` (name -> array) into that blob
and back.  Layout (little endian):

    0   char[8]  magic "HDRW1\\0\\0\\0"
    8   u32      n_entries
    12  u32      reserved
    16  n_entries x { char name[96]; u32 dtype(0=f32 1=f16 2=i8 3=i64);
                      u32 ndim; u32 dims[4]; u64 offset; u64 nbytes }   (136 B each)
    ..  data, every tensor 64-byte aligned; offsets are from the blob start

HG weights are not shipped with the reference: timing and
parity use ``seeded_hg_state`` below, a numpy-only generator both the golden
script (feeding the reference's own ``Hallucination_Generator``) and the GPU box
can reproduce bit for bit.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

from . import arch

MAGIC = b"HDRW1\0\0\0"
_ENTRY = struct.Struct("<96sII4IQQ")
_DT_CODE = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.int8): 2,
            np.dtype(np.int64): 3}
_DT_FROM = {v: k for k, v in _DT_CODE.items()}


def pack_state(state) -> bytes:
    """name -> ndarray (or torch tensor)  ->  .hdrw blob."""
    items = []
    for name, value in state.items():
        if hasattr(value, "detach"):
            value = value.detach().cpu().numpy()
        a = np.ascontiguousarray(value)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        if a.dtype not in _DT_CODE:
            raise ValueError(f"unsupported dtype {a.dtype} for {name}")
        if a.ndim > 4:
            raise ValueError(f"rank > 4 for {name}")
        if len(name.encode()) >= 96:
            raise ValueError(f"name too long: {name}")
        items.append((name, a))
    head = 16 + _ENTRY.size * len(items)
    off = (head + 63) & ~63
    table, blobs = [], []
    for name, a in items:
        dims = list(a.shape) + [1] * (4 - a.ndim)
        nbytes = a.nbytes
        table.append(_ENTRY.pack(name.encode(), _DT_CODE[a.dtype], a.ndim, *dims, off, nbytes))
        blobs.append((off, a.tobytes()))
        off = (off + nbytes + 63) & ~63
    out = bytearray(off)
    out[0:8] = MAGIC
    struct.pack_into("<II", out, 8, len(items), 0)
    pos = 16
    for t in table:
        out[pos:pos + _ENTRY.size] = t
        pos += _ENTRY.size
    for o, b in blobs:
        out[o:o + len(b)] = b
    return bytes(out)


def unpack_state(blob: bytes) -> "OrderedDict[str, np.ndarray]":
    if blob[:8] != MAGIC:
        raise ValueError("not an HDRW1 weight pack")
    n, _ = struct.unpack_from("<II", blob, 8)
    out = OrderedDict()
    pos = 16
    for _ in range(n):
        name, dt, ndim, d0, d1, d2, d3, off, nbytes = _ENTRY.unpack_from(blob, pos)
        pos += _ENTRY.size
        name = name.rstrip(b"\0").decode()
        shape = (d0, d1, d2, d3)[:ndim]
        out[name] = np.frombuffer(blob, dtype=_DT_FROM[dt], count=nbytes // _DT_FROM[dt].itemsize,
                                  offset=off).reshape(shape)
    return out


def load_pack(path) -> "OrderedDict[str, np.ndarray]":
    with open(path, "rb") as f:
        return unpack_state(f.read())


def save_pack(path, state) -> None:
    with open(path, "wb") as f:
        f.write(pack_state(state))


def check_hr_state(state) -> None:
    """Raise ValueError unless ``state`` has exactly the HR (AGCM+LE) tensors; an INT8 runtime layer may hold
    ``weight_int8`` (+ ``scale`` / ``w_scale``, ``x_scale``, ``x_zero``) in place of ``weight``."""
    want = dict(arch.hr_params())
    for k in list(want):
        if k.endswith(".weight") and k not in state and k[:-len(".weight")] + ".weight_int8" in state:
            want[k[:-len(".weight")] + ".weight_int8"] = want.pop(k)
    missing = [k for k in want if k not in state]
    if missing:
        raise ValueError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[0]}")
    for k, shp in want.items():
        if tuple(state[k].shape) != tuple(shp):
            raise ValueError(f"bad shape for {k}: {tuple(state[k].shape)} != {shp}")


def is_int8_state(state) -> bool:
    return any(k.endswith(".weight_int8") for k in state)


def is_hg_w8a8_layout(state) -> bool:
    """True when a quantised HG state has exactly the W8A8 layers the int8 HG kernels serve (HG_W8A8_GROUPS below)."""
    want = {name for layers in HG_W8A8_GROUPS.values() for name in layers}
    have = {k[: -len(".x_scale")] for k in state if k.endswith(".x_scale")}
    return have == want


def normalize_int8_state(state) -> "OrderedDict[str, np.ndarray]":
    """INT8 runtime checkpoint (hdrtvnet_torch.py:1755-1883) as plain arrays for the weight pack, quantised layers kept:
    ``weight_int8`` int8, per-channel ``scale`` / ``w_scale`` and ``bias`` as stored, ``x_scale`` / ``x_zero`` as the fp32
    scalars the reference promotes them to (``W8A8Conv2d._apply``, 339-349)."""
    out = OrderedDict()
    for k, v in state.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if k.endswith(".weight_int8"):
            out[k] = np.ascontiguousarray(a.astype(np.int8))
        elif k.endswith((".x_scale", ".x_zero")):
            out[k] = a.astype(np.float32).reshape(1)
        elif a.dtype.kind == "f":
            out[k] = a.astype(np.float32) if a.dtype == np.float64 else a
        elif a.dtype.kind in "iu" and a.ndim == 0:
            continue            # e.g. num_batches_tracked
        else:
            out[k] = a
    return out


def dequantize_int8_state(state, compute: str = "fp16") -> "OrderedDict[str, np.ndarray]":
    """INT8 runtime checkpoint (W8Conv2d / W8A8Conv2d / W8Linear / W8A8Linear state,
    hdrtvnet_torch.py:233-410) -> plain ``<layer>.weight`` / ``<layer>.bias`` tensors, the way the
    reference itself runs these checkpoints on ROCm: ``predequantize="auto"`` replaces every
    quantised layer by a native conv with ``w = weight_int8.to(cd) * scale`` and DROPS the
    activation fake-quant (``_predequantize_conv`` 444-462, auto rule 1893-1899).  ``compute``
    is the dtype of that product: "fp16" on a GPU, "fp32" on CPU (1766-1773)."""
    cd = np.float16 if compute == "fp16" else np.float32
    out = OrderedDict()
    for k, v in state.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if k.endswith(".weight_int8"):
            base = k[: -len(".weight_int8")]
            sc = state.get(base + ".w_scale", state.get(base + ".scale"))
            if sc is None:
                raise ValueError(f"INT8 checkpoint has no scale for {base}")
            sc = sc.detach().cpu().numpy() if hasattr(sc, "detach") else np.asarray(sc)
            shape = (-1,) + (1,) * (a.ndim - 1)
            out[base + ".weight"] = (a.astype(cd) * sc.astype(cd).reshape(shape)).astype(cd).astype(np.float32)
        elif k.endswith((".w_scale", ".scale", ".x_scale", ".x_zero")):
            continue
        else:
            out[k] = a.astype(np.float32)
    return out


def seeded_hg_state(seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """Deterministic stand-in for the absent HG.pt (keys as Hallucination_Generator's).

    Conv weights: N(0, 2/fan_in) (the reference's own kaiming fan-in init,
    Hallucination_arch.py:13-21), biases N(0, 0.02); BatchNorm gamma N(1, 0.02),
    beta N(0, 0.02), running_mean N(0, 0.1), running_var U(0.5, 1.5) so that BN
    folding is exercised with non-trivial statistics.  The 1x1 fuse convs and
    conv_last are scaled down so the seeded head stays O(1) in fp16.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in arch.hg_params(with_counters=True):
        if name.endswith("num_batches_tracked"):
            out[name] = np.array(1, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif ".1.weight" in name:
            out[name] = (1.0 + 0.02 * rng.standard_normal(shape)).astype(np.float32)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            std = np.sqrt(2.0 / fan_in)
            if shape[2] == 1:          # 1x1 fuse convs see un-normalised concat inputs
                std *= 0.5
            out[name] = (std * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (0.02 * rng.standard_normal(shape)).astype(np.float32)
    return out


def synthetic_frame(h: int, w: int, seed: int = 1234, kind: str = "noise") -> np.ndarray:
    """u8 BGR HWC test frame.  ``noise``: BASELINE.md section 3 protocol
    (``default_rng(seed).integers(0,256)``); ``gradient``: smooth ramps plus a few
    bright blobs so the HG highlight mask is a 0/1 mixture."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind != "gradient":
        raise ValueError(kind)
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32),
                         np.linspace(0, 1, w, dtype=np.float32), indexing="ij")
    img = np.stack([0.15 + 0.5 * xx, 0.1 + 0.45 * yy, 0.2 + 0.3 * (1 - xx) * yy], axis=2)
    for _ in range(6):
        cy, cx = rng.uniform(0.1, 0.9, 2)
        r = rng.uniform(0.04, 0.15)
        d2 = ((yy - cy) ** 2 + (xx - cx) ** 2) / (r * r)
        img += (0.9 * np.exp(-d2))[:, :, None] * rng.uniform(0.7, 1.0, 3).astype(np.float32)
    img += rng.normal(0, 0.01, img.shape).astype(np.float32)
    return np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ W8A8 HG (BASELINE configs[4], int8 MFMA)
# The HG layers that run on int8 MFMA, with the activation tensor(s) each one reads.  Tensors that are concatenated
# (pixel-shuffled Up_conv output + encoder skip) or read by two layers share ONE quantiser, so that each activation
# exists once in HBM as int8: in the reference's checkpoint format (per-layer x_scale / x_zero,
# hdrtvnet_torch.py:296-364) this is simply equal values on the layers of a group.  conv1, conv10 and conv_last
# (3-channel input / output) stay fp16, as the first and last layers do in the reference's mixed recipes
# (configs/qat_layouts/original_hg_composite_mixed_w8a8.txt keeps 24 layers fp16).
HG_W8A8_GROUPS = OrderedDict([
    # group (= activation tensors)      layers reading it
    ("p1", ("conv2.0",)),
    ("conv2+up4", ("conv3_1.0", "conv9")),
    ("p3", ("conv3_2.0",)),
    ("conv3_2+up3", ("conv4_1.0", "conv8")),
    ("p4", ("conv4_2.0",)),
    ("conv4_2+up2", ("conv5_1.0", "conv7")),
    ("p5", ("conv5_2.0",)),
    ("conv5_2+up1", ("conv_code1.0", "conv6")),
    ("pc", ("conv_code2.0",)),
    ("conv_code2", ("Up_conv1.0",)),
    ("conv6", ("Up_conv2.0",)),
    ("conv7", ("Up_conv3.0",)),
    ("conv8", ("Up_conv4.0",)),
    ("conv9", ("Up_conv5.0",)),
])


def activation_qparams(lo: float, hi: float, integer_zero: bool = True):
    """Range -> (x_scale, x_zero) of the reference's asymmetric u8 activation quantiser
    (``x_q = round((x - x_zero) / x_scale).clamp(0, 255)``).  ``integer_zero`` (default): x_zero = -k * x_scale,
    k in 0..255, x_scale fp16-representable so that k * x_scale is exact in fp32.  Zero padding (applied after
    dequantisation in the reference) is then the code k exactly, which lets an integer kernel pad with a constant.
    ``integer_zero=False``: the reference's own ``calibrate_w8a8(method="max")`` rule (hdrtvnet_torch.py:1001-1099):
    x_zero = running minimum, x_scale = (max - min) / 255 -- a float zero point."""
    if not integer_zero:
        lo, hi = float(lo), float(hi)
        s = np.float32(max(hi - lo, 1e-6) / 255.0)
        return float(s), float(np.float32(lo))
    lo, hi = min(float(lo), 0.0), max(float(hi), 0.0)
    s = np.float32(np.float16(max(hi - lo, 1e-6) / 255.0 * 1.0005))       # never round the range down
    k = int(np.clip(np.rint(-lo / float(s)), 0, 255))
    return float(s), float(np.float32(-k) * s)


def hg_w8a8_state(hg_state, act_ranges, integer_zero: bool = True) -> "OrderedDict[str, np.ndarray]":
    """fp HG state + ``{group: (lo, hi)}`` calibration ranges -> runtime W8A8 state in the reference's key layout
    (``<layer>.weight_int8`` int8, ``.w_scale`` per output channel, ``.bias``, ``.x_scale``, ``.x_zero``;
    W8A8Conv2d.__init__, hdrtvnet_torch.py:309-337: w_scale = max|w| / 127, round, clamp).  BatchNorm tensors and the
    fp16 layers pass through unchanged."""
    layer_q = {}
    for group, layers in HG_W8A8_GROUPS.items():
        qp = activation_qparams(*act_ranges[group], integer_zero=integer_zero)
        for name in layers:
            layer_q[name] = qp
    out = OrderedDict()
    for k, v in hg_state.items():
        a = np.asarray(v)
        base = k[: -len(".weight")] if k.endswith(".weight") else None
        if base in layer_q:
            w = a.astype(np.float32)
            w_scale = np.maximum(np.abs(w.reshape(w.shape[0], -1)).max(axis=1), np.float32(1e-8)) / np.float32(127.0)
            q = np.clip(np.rint(w / w_scale.reshape(-1, 1, 1, 1)), -128, 127).astype(np.int8)
            out[base + ".weight_int8"] = q
            out[base + ".w_scale"] = w_scale.astype(np.float32)
            out[base + ".x_scale"] = np.array(layer_q[base][0], np.float32)
            out[base + ".x_zero"] = np.array(layer_q[base][1], np.float32)
        else:
            out[k] = a
    return out


def seeded_hg_w8a8_state(seed: int = 1234, integer_zero: bool = True) -> "OrderedDict[str, np.ndarray]":
    """The seeded HG stand-in quantised with the calibration table shipped in ``data/`` (ranges measured with the fp32
    network on the synthetic gradient frames; tests/golden/gen_golden_hg_w8a8.py writes it)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"hg_w8a8_calib_seed{seed}.json")
    with open(path) as f:
        ranges = {k: tuple(v) for k, v in json.load(f)["ranges"].items()}
    return hg_w8a8_state(seeded_hg_state(seed), ranges, integer_zero=integer_zero)
