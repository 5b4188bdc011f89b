"""HDRTVNetMI355X -- drop-in processor for the reference's ``HDRTVNetTorch``.

Mirrors the call surface of ``src/models/hdrtvnet_torch.py:1513-2472`` (constructor
arguments, ``preprocess`` / ``infer`` / ``postprocess`` / ``process`` /
``process_timed`` / ``warmup_compile`` / ``end_profiling`` and the attributes callers
read: ``gui_pipeline_worker_model.py:52,238,242,274``, ``main.py:68,96,104,487``,
``compile_kernels.py:332-345``), the same way the reference's own
``HDRTVNetTensorRT(HDRTVNetTorch)`` is a second backend behind that surface.

All arithmetic runs in ``libhdrtv_mi355x.so`` (hand-written gfx950 HIP kernels) through
the C ABI in ``include/hdrtv_mi355x.h``.  torch is used for device buffers, pinned host
buffers and streams only.  There is no CPU path and no eager fallback: a missing
library, a non-gfx950 device or ``device="cpu"`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch

from . import lib as _L
from . import weights as _W


def _split_composite(state):
    """An ``HG_Composite`` checkpoint (keys ``base.AGCM.* / base.LE.* / hg.*``, hdrtvnet_torch.py:1797-1830 and
    HG_Composite_arch.py:30-76) -> (HR state, HG state or None); any other state passes through as (state, None)."""
    if not any(k.startswith("base.") for k in state):
        return state, None
    hr = {k[5:]: v for k, v in state.items() if k.startswith("base.")}
    hg = {k[3:]: v for k, v in state.items() if k.startswith("hg.")}
    return hr, (hg or None)


def _load_state(path_or_state, what):
    """Accepts a mapping, an ``.hdrw`` pack, or a torch checkpoint (raw state_dict or the
    reference's ``{"state_dict":..., "architecture":...}`` wrapper, hdrtvnet_torch.py:1491)."""
    if isinstance(path_or_state, dict):
        return path_or_state
    path = str(path_or_state)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{what} not found: {path}")
    if path.endswith(".hdrw"):
        return _W.load_pack(path)
    payload = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(payload, dict) and "state_dict" in payload:
        payload = payload["state_dict"] or {}
    if not isinstance(payload, dict):
        raise ValueError(f"{what}: unsupported checkpoint payload in {path}")
    return {(k[7:] if k.startswith("module.") else k): v for k, v in payload.items()}


def kernel_arithmetic(tag):
    """The arithmetic class of a launch by its profile tag (csrc/api_graph.hip writes the tags): 'int8' = v_mfma_i32_*_i8,
    'fq-f16' = activation quantiser in registers + fp16 MFMA on dequantised weights, 'fq-f32' = fp32 fake-quant, 'f16'."""
    t = tag.lower()
    if "rows<fq>" in t:
        return "fq-f16"
    if t.startswith("cls_block<fq>"):
        return "fq-f32"
    if "_i8" in t or ",i8" in t or "<i8" in t or "q8" in t or "-i8" in t:
        return "int8"
    return "f16"


def summarize_profile(profile):
    """{class: {"launches", "gmac", "ms", "kernels"}} + "text": one sentence for a bench label, taken from what ran."""
    out = {k: {"launches": 0, "gmac": 0.0, "ms": 0.0, "kernels": set()} for k in ("int8", "fq-f16", "fq-f32", "f16")}
    for _layer, kern, ms, macs, _bytes in profile:
        c = out[kernel_arithmetic(kern)]
        c["launches"] += 1
        c["gmac"] += macs / 1e9
        c["ms"] += ms
        c["kernels"].add(kern.split("<")[0])
    total = sum(c["gmac"] for c in out.values()) or 1.0
    names = {"int8": "on int8 MFMA", "fq-f16": "as fake-quant on fp16 MFMA inside the fused LE row kernels",
             "fq-f32": "as fp32 fake-quant", "f16": "on fp16 MFMA / vector units"}
    parts = [f"{100 * c['gmac'] / total:.1f} % of the MACs in {c['launches']} launches {names[k]}" for k, c in out.items() if c["launches"]]
    res = {k: {"launches": c["launches"], "gmac": round(c["gmac"], 2), "ms": round(c["ms"], 3), "kernels": sorted(c["kernels"])}
           for k, c in out.items()}
    res["text"] = "; ".join(parts)
    return res


class HDRTVNetMI355X:
    """MI355X-native backend with the ``HDRTVNetTorch`` surface.

    ``model_path``: HR checkpoint (``HR.pt``, an ``.hdrw`` pack, or a state mapping).
    ``hg_weights``: HG checkpoint path / mapping, or ``"seeded:<int>"`` for the deterministic
    stand-in of the un-shipped ``HG.pt`` (weights.seeded_hg_state); ``"seeded-w8a8:<int>"`` is that
    stand-in as a W8A8 checkpoint (weights.seeded_hg_w8a8_state).  An HG checkpoint that carries
    ``weight_int8`` / ``x_scale`` / ``x_zero`` tensors in the layout of weights.HG_W8A8_GROUPS runs its
    18 quantised layers on int8 MFMA (BASELINE configs[4]); any other layout is rejected.  As in the reference
    (hdrtvnet_torch.py:2065-2086): an explicit but missing path raises ``FileNotFoundError``;
    with no path given and ``use_hg=True`` the model silently continues without HG.
    An ``HG_Composite`` checkpoint (``base.*`` + ``hg.*`` keys, the layout of the reference's
    ``pytorch_int8/hg/HR_HG_*.pt``) is split into its two halves; with ``use_hg=True`` and no HG tensors anywhere the
    reference's search order applies (``_resolve_hg_weights``, 2016-2042: explicit path, ``HG.pt`` next to the model,
    ``<cwd>/src/models/weights/original/HG.pt``).
    ``compile_*`` and ``force_channels_last`` are accepted and ignored: there is nothing to JIT (kernels are precompiled
    for gfx950).  ``predequantize``: "auto" / True run an INT8 checkpoint as fp16 convs of the dequantised weights (the
    reference's ROCm behaviour), "off" / False keep its W8A8 layers and run them on int8 MFMA.
    ``use_cuda_graphs=True`` replays ``infer`` from a captured hipGraph (2306-2331).  ``fast_condition_resize=True`` (or
    ``HDRTVNET_FAST_COND_RESIZE=1``) derives the condition map with the bilinear 0.25x resize, ``HDRTVNET_ZERO_COND=1``
    zeroes it (1539-1543, 2262-2276).
    ``lanes`` (no reference counterpart; 1 or 2, default 1): frames in flight on the device.  Each lane has its own activation
    workspace, boundary tensors and HIP stream (``enqueue_frame``; the fp32 preset takes one lane only); the reference-shaped calls (``process`` / ``preprocess`` /
    ``infer`` / ``postprocess``) always run on lane 0 and the caller's current stream.
    """

    def __init__(self, model_path, device="auto", precision="auto",
                 compile_model=True, force_compile=False, compile_mode="auto",
                 use_cuda_graphs=False, force_channels_last=False,
                 predequantize="auto", hg_weights=None, use_hg=True,
                 warmup_passes=3, fast_condition_resize=False, lanes=1, _ab_library=False):
        self.model_path = model_path
        self._lanes = int(lanes)
        if not 1 <= self._lanes <= (4 if os.environ.get("HDRTV_LANES_ANY") == "1" else 2):
            raise ValueError("lanes must be 1 or 2")
        self._warmup_passes = int(warmup_passes)
        env_true = lambda n: str(os.environ.get(n, "")).strip().lower() in ("1", "true", "yes", "on")   # noqa: E731
        self._fast_condition_resize = bool(fast_condition_resize) or env_true("HDRTVNET_FAST_COND_RESIZE")
        self._fast_zero_condition = env_true("HDRTVNET_ZERO_COND")
        self._use_cuda_graphs = bool(use_cuda_graphs)
        self._graphs = {}
        self._profiling = False
        self.device = self._resolve_device(device)
        self.precision = self._resolve_precision(precision)
        self._use_cuda = True
        # fp32: the reference's maximum-precision preset -- the same graph on fp32 tensors (csrc/fp32_ops.hip, fp32_graph.hip)
        self._fp32 = self.precision == "fp32"
        self._dtype = torch.float32 if self._fp32 else torch.float16
        self._compiled = False          # nothing is JIT-compiled; warmup_compile() is a no-op
        self._compile_mode = None
        self._trt_engine = None
        self._is_w8_model = False
        self._memory_format_name = "nhwc-internal"
        self.engine_path = _L.LIB_PATH
        self.model = None               # no nn.Module exists; callers treat None as "0 MB"
        self._lib = _L.load(ab=bool(_ab_library))     # _ab_library: tests only (superseded kernels as bit-identity yardsticks)
        self._ctx = C.c_void_p()

        hr_state, hg_from_ckpt = _split_composite(_load_state(model_path, "model weights"))
        if self.precision.startswith("int8"):
            # hdrtvnet_torch.py:1748-1963: an INT8 runtime checkpoint.  On ROCm the reference
            # pre-dequantizes it to native fp16 convs at load time ("auto", 1893-1899): INT8 is
            # compressed storage, compute is fp16 -- exactly what is done here.
            if not _W.is_int8_state(hr_state):
                raise ValueError(f"precision '{self.precision}' needs an INT8 checkpoint (weight_int8 tensors)")
            if str(predequantize).lower() in ("off", "false", "0", "no"):
                # the quantised layers stay in place (hdrtvnet_torch.py:1893-1917 with predequantize off): W8A8 layers run
                # on int8 MFMA with the reference's activation quantisers, W8 (weight-only) layers as fp16 convs of the
                # dequantised weights, exactly what W8Conv2d.forward computes.  hdrtv_create rejects a checkpoint with a
                # W8A8 layer it has no int8 kernel for.
                hr_state = _W.normalize_int8_state(hr_state)
                self._is_w8_model = True
            else:
                hr_state = _W.dequantize_int8_state(hr_state, "fp16")
                self._is_w8_model = False          # as the reference after pre-dequantization (1919)
        elif _W.is_int8_state(hr_state):
            raise ValueError("INT8 checkpoint given with a floating-point precision; use precision='int8-full'/'int8-mixed'")
        try:
            _W.check_hr_state(hr_state)
        except ValueError as exc:
            raise ValueError(f"Unsupported checkpoint for the MI355X backend: {exc}") from exc
        hg_state = None
        self._use_hg = bool(use_hg)
        if self._use_hg:
            if isinstance(hg_weights, str) and hg_weights.startswith("seeded:"):
                hg_state = _W.seeded_hg_state(int(hg_weights.split(":", 1)[1]))
            elif isinstance(hg_weights, str) and hg_weights.startswith("seeded-w8a8:"):
                hg_state = _W.seeded_hg_w8a8_state(int(hg_weights.split(":", 1)[1]))
            elif isinstance(hg_weights, str) and hg_weights.startswith("seeded-w8a8-minmax:"):
                # the same stand-in calibrated the reference's way (x_zero = running minimum: float zero points)
                hg_state = _W.seeded_hg_w8a8_state(int(hg_weights.split(":", 1)[1]), integer_zero=False)
            elif hg_weights is not None:
                hg_state = _split_composite(_load_state(hg_weights, "HG weights"))     # FileNotFoundError if missing
                hg_state = hg_state[1] if hg_state[1] is not None else hg_state[0]
            elif hg_from_ckpt is not None:
                hg_state = hg_from_ckpt                                # HG_Composite checkpoint: its own hg.* half
            else:
                found, searched = self._resolve_hg_weights(model_path)
                if found:
                    hg_state = _load_state(found, "HG weights")
                    hg_weights = found
                elif self.precision.startswith("int8"):
                    # hdrtvnet_torch.py:1928-1937: an INT8 no-HG checkpoint with HG requested needs split HG weights
                    raise FileNotFoundError("INT8 HG weights were requested but not found.\n  Searched paths:\n" +
                                            "\n".join(f"  - {p}" for p in searched) + "\n  Pass hg_weights or disable HG.")
                else:
                    print("WARNING: HG weights not found; continuing with no-HG model.\n  Searched paths:\n" +
                          "\n".join(f"  - {p}" for p in searched))
                    self._use_hg = False
            if hg_state is not None and _W.is_int8_state(hg_state) and not self._is_w8_model and self.precision.startswith("int8") \
                    and not _W.is_hg_w8a8_layout(hg_state):
                # a quantised HG half in a layout the int8 HG kernels do not serve, with predequantize on: fp16 convs
                hg_state = _W.dequantize_int8_state(hg_state, "fp16")
        self._hg_weights = hg_weights if self._use_hg else None
        self._hg_int8 = hg_state is not None and _W.is_int8_state(hg_state)
        self._hg_state_fp = hg_state if (hg_state is not None and not self._hg_int8) else None
        hr_blob = _W.pack_state(hr_state if self._is_w8_model else {k: hr_state[k] for k, _ in _arch_hr()})
        hg_blob = _W.pack_state({k: v for k, v in hg_state.items()
                                 if not k.endswith("num_batches_tracked")}) if hg_state is not None else b""
        rc = self._lib.hdrtv_create_ex(hr_blob, len(hr_blob), hg_blob if hg_blob else None, len(hg_blob),
                                       self.device.index or 0, _L.PREC_F32 if self._fp32 else _L.PREC_F16, C.byref(self._ctx))
        if rc < 0:
            msg = self._lib.hdrtv_last_error(self._ctx).decode() if self._ctx else "allocation failed"
            self._lib.hdrtv_destroy(self._ctx)
            self._ctx = C.c_void_p()
            if rc == _L.EWEIGHTS:
                raise ValueError(f"model backend failed - {msg}")
            raise RuntimeError(f"model backend failed - {msg}")

        if self._lanes > 1:
            # (HDRTV_EINVAL for the fp32 preset: one lane only there, include/hdrtv_mi355x.h)
            try:
                self._chk(self._lib.hdrtv_set_lanes(self._ctx, self._lanes), "hdrtv_set_lanes")
            except Exception:
                self._lib.hdrtv_destroy(self._ctx)
                self._ctx = C.c_void_p()
                raise
        self._lane_bufs, self._lane_streams = [], []
        if self._fast_zero_condition or self._fast_condition_resize:
            self._chk(self._lib.hdrtv_set_cond_mode(self._ctx, 2 if self._fast_zero_condition else 1), "hdrtv_set_cond_mode")
        self._buf_hw = None
        self._gpu_input = self._gpu_cond = self._gpu_raw = None
        self._pin_input = self._pin_output = None
        self._gpu_out = self._gpu_agcm = self._gpu_u8 = None
        print(f"MI355X device : {self.device}")
        # which arithmetic a W8A8 layer runs in depends on the kernel the launch sequence picks for it per resolution
        # (int8 MFMA, or the fake-quant form inside a fused LE row kernel): `execution_summary()` reports it from a profile
        print(f"MI355X precision: {self.precision}{' (W8A8 layers kept quantised: predequantize off)' if self._is_w8_model else ''}  "
              f"(HG {('W8A8 on int8 MFMA' if self._hg_int8 else 'on') if self._use_hg else 'off'})")
        if self._warmup_passes > 0:
            self._warmup()

    # ------------------------------------------------------------------ helpers
    def _resolve_device(self, device):
        mode = str(device).lower()
        if mode not in ("auto", "cuda", "cpu") and not mode.startswith("cuda:"):
            raise ValueError("device must be one of: auto, cuda, cpu")
        if mode == "cpu":
            raise RuntimeError("the MI355X backend has no CPU path; use the reference's HDRTVNetTorch for device='cpu'")
        if not torch.cuda.is_available():
            raise RuntimeError("CUDA/ROCm device not available for the MI355X backend.")
        if mode.startswith("cuda:"):
            return torch.device(mode)
        return torch.device("cuda", torch.cuda.current_device())

    def _resolve_precision(self, precision):
        p = str(precision).lower()
        if p not in {"auto", "fp16", "fp32", "int8-full", "int8-mixed"}:
            raise ValueError("precision must be one of: auto, fp16, fp32, int8-full, int8-mixed")
        if p in ("auto", "fp16"):
            return "fp16"
        return p              # int8-*: INT8 storage, fp16 compute (the reference's own ROCm behaviour); fp32: the fp32 graph

    def _resolve_hg_weights(self, model_path):
        """hdrtvnet_torch.py:2016-2042 (the user override is handled by the caller): HG.pt next to the checkpoint, then
        the repo-default location relative to the working directory; ``.hdrw`` packs are accepted beside ``.pt``."""
        cands = []
        if not isinstance(model_path, dict):
            d = os.path.dirname(os.path.abspath(str(model_path)))
            cands += [os.path.join(d, "HG.pt"), os.path.join(d, "HG.hdrw")]
        cands.append(os.path.join(os.getcwd(), "src", "models", "weights", "original", "HG.pt"))
        for p in cands:
            if os.path.isfile(p):
                return p, cands
        return None, cands

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, rc, what):
        return _L.check(self._lib, self._ctx, rc, what)

    def _warmup(self):
        h, w = 1080, 1920
        dummy = np.zeros((h, w, 3), dtype=np.uint8)
        for _ in range(self._warmup_passes):
            self.process(dummy)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ buffers
    def _ensure_buffers(self, h, w):
        """hdrtvnet_torch.py:2198-2233."""
        if self._buf_hw == (h, w):
            return
        self._graphs.clear()                      # captured graphs point into the buffers replaced below
        with torch.cuda.device(self.device):
            self._chk(self._lib.hdrtv_reserve(self._ctx, h, w), "hdrtv_reserve")
            ch, cw = max(1, h // 4), max(1, w // 4)
            dev = self.device
            self._gpu_input = torch.empty((1, 3, h, w), dtype=self._dtype, device=dev)
            self._gpu_cond = torch.empty((1, 3, ch, cw), dtype=self._dtype, device=dev)
            self._gpu_raw = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
            self._gpu_u8 = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
            self._gpu_out = torch.empty((1, 3, h, w), dtype=torch.float32 if (self._use_hg or self._fp32) else torch.float16, device=dev)
            self._gpu_agcm = torch.empty((1, 3, h, w), dtype=self._dtype, device=dev)
            self._pin_input = torch.empty((h, w, 3), dtype=torch.uint8, pin_memory=True)
            self._pin_output = torch.empty((h, w, 3), dtype=torch.uint8, pin_memory=True)
            # lane l > 0: its own boundary tensors (input, cond, out, agcm) and stream; lane 0 is the set above
            self._lane_bufs = [(self._gpu_input, self._gpu_cond, self._gpu_out, self._gpu_agcm)]
            for _ in range(1, self._lanes):
                self._lane_bufs.append((torch.empty_like(self._gpu_input), torch.empty_like(self._gpu_cond),
                                        torch.empty_like(self._gpu_out), torch.empty_like(self._gpu_agcm)))
            if len(self._lane_streams) != self._lanes:
                self._lane_streams = [torch.cuda.Stream(dev) for _ in range(self._lanes)]
        self._buf_hw = (h, w)

    # ------------------------------------------------------------------ lanes
    @property
    def lanes(self):
        return self._lanes

    def lane_stream(self, lane):
        """The HIP stream lane ``lane``'s frames are enqueued on (a ``torch.cuda.Stream``; valid after the first
        ``_ensure_buffers``)."""
        return self._lane_streams[lane]

    def enqueue_frame(self, lane, src_bgr_ptr, h, w, dst_rgb48_ptr, stream=None):
        """One frame of the hot path on lane ``lane``, stream-ordered and without any host synchronisation: u8 BGR frame in
        device memory at ``src_bgr_ptr`` -> hdrtv_preprocess -> hdrtv_infer_lane -> hdrtv_post_rgb48 -> u16 RGB48 at the device
        address ``dst_rgb48_ptr``.  ``stream``: a ``torch.cuda.Stream`` (default: the lane's own).  Frames enqueued on different
        lanes may overlap on the device; the bytes written do not depend on the lane (tests/test_gpu_lanes.py).  The caller
        orders the use of ``src`` / ``dst`` against the stream (events), as with any asynchronous launch."""
        if not 0 <= lane < self._lanes:
            raise ValueError(f"lane {lane} of {self._lanes}")
        self._ensure_buffers(h, w)
        st = C.c_void_p((stream if stream is not None else self._lane_streams[lane]).cuda_stream)
        tin, tcond, tout, tagcm = self._lane_bufs[lane]
        dt = _L.F32 if (self._use_hg or self._fp32) else _L.F16
        self._chk(self._lib.hdrtv_preprocess(self._ctx, st, src_bgr_ptr, h, w, tin.data_ptr(), tcond.data_ptr()), "hdrtv_preprocess")
        self._chk(self._lib.hdrtv_infer_lane(self._ctx, lane, st, tin.data_ptr(), tcond.data_ptr(), h, w, tout.data_ptr(), dt,
                                             tagcm.data_ptr()), "hdrtv_infer_lane")
        self._chk(self._lib.hdrtv_post_rgb48(self._ctx, st, tout.data_ptr(), dt, h, w, dst_rgb48_ptr), "hdrtv_post_rgb48")

    # ------------------------------------------------------------------ API
    @torch.inference_mode()
    def preprocess(self, frame_bgr):
        """hdrtvnet_torch.py:2238-2296.  Returns processor-owned persistent tensors."""
        if frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3 or frame_bgr.dtype != np.uint8:
            raise ValueError("frame_bgr must be uint8 [H,W,3]")
        h, w = frame_bgr.shape[:2]
        self._ensure_buffers(h, w)
        # plain memcpy into the pinned slot (GIL released).  Not tensor.copy_: ATen parallelises a 25 MB copy
        # over every OpenMP thread, whose spin-wait afterwards starves the HIP runtime's completion handling
        # (measured on the GPU box: every third 4K frame stalled ~55 ms behind a 128-thread copy).
        staged = getattr(frame_bgr, "pinned_tensor", None)
        dev_copy, ready = getattr(frame_bgr, "device_tensor", None), getattr(frame_bgr, "ready_event", None)
        if dev_copy is not None and ready is not None and dev_copy.device == self.device and tuple(dev_copy.shape) == (h, w, 3):
            # already uploaded by the prefetcher on its own stream (hipMemcpyAsync + hipEvent handoff): wait for that
            # event in stream order and unpack straight from its buffer
            torch.cuda.current_stream(self.device).wait_event(ready)
            self._chk(self._lib.hdrtv_preprocess(self._ctx, self._stream(), dev_copy.data_ptr(), h, w,
                                                 self._gpu_input.data_ptr(), self._gpu_cond.data_ptr()), "hdrtv_preprocess")
            return self._gpu_input, self._gpu_cond
        if staged is not None and staged.is_pinned() and tuple(staged.shape) == (h, w, 3):
            # the frame already lives in page-locked memory (playback.PinnedPrefetch filled it on its own thread while
            # the previous frame was on the GPU): upload it as it is
            self._gpu_raw.copy_(staged, non_blocking=True)
        else:
            src = np.ascontiguousarray(frame_bgr)
            C.memmove(self._pin_input.data_ptr(), src.ctypes.data, src.nbytes)
            self._gpu_raw.copy_(self._pin_input, non_blocking=True)
        self._chk(self._lib.hdrtv_preprocess(self._ctx, self._stream(), self._gpu_raw.data_ptr(), h, w,
                                             self._gpu_input.data_ptr(), self._gpu_cond.data_ptr()), "hdrtv_preprocess")
        return self._gpu_input, self._gpu_cond

    @torch.inference_mode()
    def preprocess_letterboxed(self, frame_bgr, out_w, out_h):
        """``preprocess(_letterbox_bgr(frame, out_w, out_h))`` (gui_scaling.py:228-244 followed by
        hdrtvnet_torch.py:2238-2296) with the resize on the device: the source frame is uploaded at ITS size and
        ``hdrtv_letterbox_u8`` writes the letterboxed u8 frame the ordinary unpack/condition kernels read.  The resize
        arithmetic is oracle/letterbox_oracle.py's restatement of OpenCV (parity with cv2 unpinned)."""
        if frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3 or frame_bgr.dtype != np.uint8:
            raise ValueError("frame_bgr must be uint8 [H,W,3]")
        sh, sw = frame_bgr.shape[:2]
        out_w, out_h = int(out_w), int(out_h)
        if (sw, sh) == (out_w, out_h):
            return self.preprocess(frame_bgr)
        self._ensure_buffers(out_h, out_w)
        if getattr(self, "_lb_shape", None) != (sh, sw):
            self._lb_pin = torch.empty((sh, sw, 3), dtype=torch.uint8, pin_memory=True)
            self._lb_dev = torch.empty((sh, sw, 3), dtype=torch.uint8, device=self.device)
            self._lb_shape = (sh, sw)
        src = np.ascontiguousarray(frame_bgr)
        C.memmove(self._lb_pin.data_ptr(), src.ctypes.data, src.nbytes)
        self._lb_dev.copy_(self._lb_pin, non_blocking=True)
        self._chk(self._lib.hdrtv_letterbox_u8(self._ctx, self._stream(), self._lb_dev.data_ptr(), sh, sw,
                                               self._gpu_raw.data_ptr(), out_h, out_w), "hdrtv_letterbox_u8")
        self._chk(self._lib.hdrtv_preprocess(self._ctx, self._stream(), self._gpu_raw.data_ptr(), out_h, out_w,
                                             self._gpu_input.data_ptr(), self._gpu_cond.data_ptr()), "hdrtv_preprocess")
        return self._gpu_input, self._gpu_cond

    @torch.inference_mode()
    def infer(self, input_cond):
        """hdrtvnet_torch.py:2301-2346.  Returns ``(out, agcm_out)`` like the eager model; both are
        processor-owned and overwritten by the next call (as HDRTVNetTensorRT's output is)."""
        tensor, cond = input_cond
        if tensor.dtype != self._dtype or cond.dtype != self._dtype or not tensor.is_cuda:
            raise ValueError(f"infer expects the {self.precision if self._fp32 else 'fp16'} CUDA tensors returned by preprocess()")
        h, w = int(tensor.shape[2]), int(tensor.shape[3])
        self._ensure_buffers(h, w)
        tensor = tensor.contiguous()
        cond = cond.contiguous()
        if tuple(cond.shape[2:]) != (max(1, h // 4), max(1, w // 4)):
            raise ValueError("cond must be [1,3,H//4,W//4]")
        if self._use_cuda_graphs and not self._profiling:
            return self._infer_graph(tensor, cond, h, w)
        self._chk(self._lib.hdrtv_infer(self._ctx, self._stream(), tensor.data_ptr(), cond.data_ptr(), h, w,
                                        self._gpu_out.data_ptr(), _L.F32 if (self._use_hg or self._fp32) else _L.F16,
                                        self._gpu_agcm.data_ptr()), "hdrtv_infer")
        return self._gpu_out, self._gpu_agcm

    def _infer_graph(self, tensor, cond, h, w):
        """hdrtvnet_torch.py:2306-2331: static input buffers, one capture per shape, replay afterwards.  The ~66 launches of
        hdrtv_infer are captured into a hipGraph on torch's capture stream (the C ABI is stream-ordered and allocation-free
        after hdrtv_reserve); a capture failure falls back to eager launches as the reference does."""
        if tensor.data_ptr() != self._gpu_input.data_ptr():
            self._gpu_input.copy_(tensor)
        if cond.data_ptr() != self._gpu_cond.data_ptr():
            self._gpu_cond.copy_(cond)

        def launch():
            self._chk(self._lib.hdrtv_infer(self._ctx, self._stream(), self._gpu_input.data_ptr(), self._gpu_cond.data_ptr(), h, w,
                                            self._gpu_out.data_ptr(), _L.F32 if (self._use_hg or self._fp32) else _L.F16,
                                            self._gpu_agcm.data_ptr()), "hdrtv_infer")

        g = self._graphs.get((h, w))
        if g is None:
            launch()                                   # eager once: one-time kernel attributes are set outside the capture
            torch.cuda.synchronize(self.device)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    launch()
            except Exception as exc:  # noqa: BLE001
                print(f"WARNING: hipGraph capture failed ({exc}); continuing with eager launches.")
                self._use_cuda_graphs = False
                launch()
                return self._gpu_out, self._gpu_agcm
            self._graphs[(h, w)] = g
        g.replay()
        return self._gpu_out, self._gpu_agcm

    @torch.inference_mode()
    def objective_metrics(self, pred, ref, peak_nits=1000.0):
        """PSNR / SSIM / dE-ITP between two ``[1,3,H,W]`` (or ``[3,H,W]``) CUDA tensors in unit range, as the reference's
        ``_psnr_bgr`` / ``_ssim_bgr`` / ``_delta_e_itp_bgr`` compute them (gui_objective_metrics.py:438-528) -> a dict
        with the reference's metric keys (``psnr_db``, ``sssim``, ``delta_e_itp``).  Parity unpinned (cv2 module)."""
        a = pred[0] if isinstance(pred, (tuple, list)) else pred
        b = ref[0] if isinstance(ref, (tuple, list)) else ref
        if a.shape != b.shape or a.shape[-3] != 3:
            raise ValueError("objective_metrics expects two tensors of the same [.., 3, H, W] shape")
        dt = torch.float32 if (a.dtype == torch.float32 or b.dtype == torch.float32) else torch.float16
        a = a.to(device=self.device, dtype=dt).contiguous()
        b = b.to(device=self.device, dtype=dt).contiguous()
        h, w = int(a.shape[-2]), int(a.shape[-1])
        out = (C.c_double * 3)()
        self._chk(self._lib.hdrtv_metrics(self._ctx, self._stream(), a.data_ptr(), b.data_ptr(),
                                          _L.F32 if dt == torch.float32 else _L.F16, h, w, float(peak_nits), out),
                  "hdrtv_metrics")
        return {"psnr_db": float(out[0]), "sssim": float(out[1]), "delta_e_itp": float(out[2])}

    @torch.inference_mode()
    def postprocess(self, output):
        """hdrtvnet_torch.py:2351-2368.  Returns a zero-copy view of processor-owned pinned memory."""
        if isinstance(output, (tuple, list)):
            output = output[0]
        h, w = int(output.shape[-2]), int(output.shape[-1])
        self._ensure_buffers(h, w)
        output = output.contiguous()
        dt = _L.F32 if output.dtype == torch.float32 else _L.F16
        if output.dtype not in (torch.float16, torch.float32):
            raise ValueError("postprocess expects an fp16 or fp32 tensor")
        self._chk(self._lib.hdrtv_post_u8(self._ctx, self._stream(), output.data_ptr(), dt, h, w,
                                          self._gpu_u8.data_ptr()), "hdrtv_post_u8")
        self._pin_output.copy_(self._gpu_u8, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._pin_output.numpy()

    @torch.inference_mode()
    def process(self, frame_bgr):
        tensor, cond = self.preprocess(frame_bgr)
        return self.postprocess(self.infer((tensor, cond)))

    @torch.inference_mode()
    def process_timed(self, frame_bgr):
        """hdrtvnet_torch.py:2379-2395 -> (output, pre_ms, run_ms, post_ms)."""
        t0 = time.perf_counter()
        tensor, cond = self.preprocess(frame_bgr)
        torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        out = self.infer((tensor, cond))
        torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
        output = self.postprocess(out)
        t3 = time.perf_counter()
        return output, (t1 - t0) * 1000.0, (t2 - t1) * 1000.0, (t3 - t2) * 1000.0

    def warmup_compile(self, width=1920, height=1080):
        """hdrtvnet_torch.py:2400-2469: only meaningful when torch.compile is active; never here."""
        return None

    def end_profiling(self):
        return None

    # ------------------------------------------------------------------ extras (tests / bench)
    def tap(self, name):
        """Internal activation by name (after infer) as a torch tensor copy: NHWC taps come back
        as [C,H,W] float32 for direct comparison with the oracle."""
        p, c, h, w, lay = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(self._lib.hdrtv_get_tap(self._ctx, name.encode(), C.byref(p), C.byref(c), C.byref(h), C.byref(w),
                                          C.byref(lay)), "hdrtv_get_tap")
        n = c.value * h.value * w.value
        torch.cuda.synchronize(self.device)
        if lay.value in (0, 1):
            buf = torch.empty(n, dtype=torch.float16, device=self.device)
        elif lay.value == 4:
            buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        elif lay.value == 5:
            buf = torch.empty(n, dtype=torch.int8, device=self.device)
        else:
            buf = torch.empty(n, dtype=torch.float32, device=self.device)
        _hip_memcpy_d2d(buf.data_ptr(), p.value, buf.numel() * buf.element_size())
        torch.cuda.synchronize(self.device)
        if lay.value in (0, 5):          # NHWC; int8 taps are codes q - 128 of the reading layer's quantiser
            return buf.view(h.value, w.value, c.value).permute(2, 0, 1).float().cpu()
        return buf.view(c.value, h.value, w.value).float().cpu()

    def _tap_device(self, name):
        """Flat device copy of an internal activation (no layout conversion)."""
        p, c, h, w, lay = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(self._lib.hdrtv_get_tap(self._ctx, name.encode(), C.byref(p), C.byref(c), C.byref(h), C.byref(w),
                                          C.byref(lay)), "hdrtv_get_tap")
        dt = {0: torch.float16, 1: torch.float16, 4: torch.uint8, 5: torch.int8}.get(lay.value, torch.float32)
        buf = torch.empty(c.value * h.value * w.value, dtype=dt, device=self.device)
        torch.cuda.synchronize(self.device)
        _hip_memcpy_d2d(buf.data_ptr(), p.value, buf.numel() * buf.element_size())
        torch.cuda.synchronize(self.device)
        return buf

    @torch.inference_mode()
    def calibrate_hg_w8a8(self, frames):
        """Activation ranges for a W8A8 HG checkpoint, the reference's ``calibrate_w8a8(method="max")``
        (hdrtvnet_torch.py:1001-1099: running min / max of every quantised layer's input over the calibration set),
        taken from this fp16 model's own activations on ``frames`` (u8 BGR).  Returns ``{group: (lo, hi)}`` for
        ``weights.hg_w8a8_state`` / ``hg_w8a8_checkpoint``; layers that read one tensor or a concatenation share a range."""
        if self._hg_state_fp is None:
            raise RuntimeError("calibrate_hg_w8a8 needs a model loaded with floating-point HG weights")
        members = {"p1": ("hg.p1",), "conv2+up4": ("hg.conv2", "hg.up4"), "p3": ("hg.p3",), "conv3_2+up3": ("hg.conv3_2", "hg.up3"), "p4": ("hg.p4",),
                   "conv4_2+up2": ("hg.conv4_2", "hg.up2"), "p5": ("hg.p5",), "conv5_2+up1": ("hg.conv5_2", "hg.up1"),
                   "pc": ("hg.pc",), "conv_code2": ("hg.conv_code2",), "conv6": ("hg.conv6",), "conv7": ("hg.conv7",),
                   "conv8": ("hg.conv8",), "conv9": ("hg.conv9",)}
        assert list(members) == list(_W.HG_W8A8_GROUPS)
        ranges = {g: [0.0, 0.0] for g in members}
        for f in frames:
            self.infer(self.preprocess(f))
            for g, taps in members.items():
                for t in taps:
                    v = self._tap_device(t)
                    ranges[g][0] = min(ranges[g][0], float(v.amin()))
                    ranges[g][1] = max(ranges[g][1], float(v.amax()))
        return {g: (lo, hi) for g, (lo, hi) in ranges.items()}

    def hg_w8a8_checkpoint(self, ranges):
        """This model's HG weights as a W8A8 state (reference key layout) for ``hg_weights=``; see calibrate_hg_w8a8."""
        if self._hg_state_fp is None:
            raise RuntimeError("hg_w8a8_checkpoint needs a model loaded with floating-point HG weights")
        return _W.hg_w8a8_state(self._hg_state_fp, ranges)

    def set_hg_mask_r(self, r=0.75):
        """``HG_Composite(mask_r=...)`` (HG_Composite_arch.py:21): threshold base of the highlight mask."""
        self._chk(self._lib.hdrtv_set_hg_mask_r(self._ctx, float(r)), "hdrtv_set_hg_mask_r")
        self._graphs.clear()          # a captured hipGraph has the old threshold baked into its kernel arguments

    def set_variant(self, name, value):
        """Developer / test switch (``hdrtv_set_variant``): which of several equivalent kernels a layer runs on."""
        self._chk(self._lib.hdrtv_set_variant(self._ctx, name.encode(), int(value)), "hdrtv_set_variant")
        self._graphs.clear()          # a captured hipGraph replays the launches it was captured with

    def get_variant(self, name):
        v = C.c_int()
        self._chk(self._lib.hdrtv_get_variant(self._ctx, name.encode(), C.byref(v)), "hdrtv_get_variant")
        return v.value

    def profile_enable(self, on=True):
        """Per-launch HIP-event timing of subsequent infer() calls (bench.py roofline).  While it is on, infer() launches
        eagerly even with ``use_cuda_graphs``: a graph replay records no events, and a graph captured with profiling on
        would carry the event records."""
        self._chk(self._lib.hdrtv_profile_enable(self._ctx, 1 if on else 0), "hdrtv_profile_enable")
        self._profiling = bool(on)

    def profile_read(self):
        """[(layer, kernel, ms, macs, bytes)] of the last infer(); synchronises on its events."""
        n = self._lib.hdrtv_profile_get(self._ctx, -1, None, None, None, None, None)
        out = []
        for i in range(max(n, 0)):
            layer, kern, ms, macs, nbytes = C.c_char_p(), C.c_char_p(), C.c_float(), C.c_double(), C.c_double()
            self._chk(self._lib.hdrtv_profile_get(self._ctx, i, C.byref(layer), C.byref(kern), C.byref(ms),
                                                  C.byref(macs), C.byref(nbytes)), "hdrtv_profile_get")
            out.append((layer.value.decode(), kern.value.decode(), ms.value, macs.value, nbytes.value))
        return out

    def execution_summary(self, profile=None):
        """What the last profiled ``infer()`` ran on, by kernel tag: MACs (and launches) on int8 MFMA, as fake-quant on fp16 MFMA
        (W8A8 layers inside ``le_*_rows<fq>``), as fp32 fake-quant (the AGCM classifier of a full-QAT checkpoint) and on fp16
        MFMA.  ``profile`` = ``profile_read()`` of a run with profiling enabled (read here when omitted)."""
        return summarize_profile(profile if profile is not None else self.profile_read())

    def infer_stats(self):
        n, m = C.c_int(), C.c_double()
        self._lib.hdrtv_infer_stats(self._ctx, C.byref(n), C.byref(m))
        return n.value, m.value

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.hdrtv_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _arch_hr():
    from . import arch
    return arch.hr_params()


_hip = None


def _hip_memcpy_d2d(dst, src, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    rc = _hip.hipMemcpy(dst, src, nbytes, 3)   # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed: {rc}")
