"""Headless counterparts of the reference's pipeline-worker mixins for the hot path.

  _load_model / _silent_warmup   src/gui_pipeline_worker_model.py:39-257   (AMD branch 202-228)
  _process_frame                 src/gui_pipeline_worker_frame_processing.py:168-331
  _stage_hdr_display_tensor      src/gui_pipeline_worker_frame_processing.py:118-156
  _tensor_to_rgb48_bytes + ring  src/gui_pipeline_worker_feeders.py:38-70, 125-249
  _hdr_feeder_fn (core)          src/gui_pipeline_worker_feeders.py:440-496

Same names, argument meaning and error behaviour (``_load_model`` swallows backend exceptions
into a status message and returns False), without PyQt/cv2/mpv: status messages go to a list
or callback, the display sink is any callable taking a ``PinnedFrame``.  Frames that are not at
the processing size are letterboxed on the device (``preprocess_letterboxed``).
"""
from __future__ import annotations

import ctypes as C
import os
import queue as _queue
import sys
import threading
import time

import numpy as np
import torch

from . import lib as _L
from .processor import HDRTVNetMI355X

_RING_FRAMES = max(2, min(8, int(os.environ.get("HDRTVNET_FEEDER_GPU_RGB48_RING_FRAMES", "3") or 3)))

# preset table: the subset of gui_config.PRECISIONS (src/gui_config.py:19-160) this backend serves
def _int8_preset(recipe, suffix):
    return {"precision": f"int8-{recipe}", "model": f"original/pytorch_int8/hg/HR_HG_original_int8_{recipe}{suffix}.pt",
            "model_nohg": f"original/pytorch_int8/hr/HR_original_int8_{recipe}{suffix}.pt"}


PRECISIONS = {
    "FP16": {"precision": "fp16", "model": "original/HR.pt", "model_nohg": "original/HR.pt",
             "hg_weights": "original/HG.pt"},
    "FP32": {"precision": "fp32", "model": "original/HR.pt", "model_nohg": "original/HR.pt",
             "hg_weights": "original/HG.pt"},
    # the six INT8 presets (all of gui_config.py's); the TensorRT-only FP8 presets are not served by this backend
    "INT8 Mixed (PTQ)": _int8_preset("mixed", ""),
    "INT8 Mixed (QAT)": _int8_preset("mixed", "_qat"),
    "INT8 Mixed (QAT) (Film)": _int8_preset("mixed", "_qat_film"),
    "INT8 Full (PTQ)": _int8_preset("full", ""),
    "INT8 Full (QAT)": _int8_preset("full", "_qat"),
    "INT8 Full (QAT) (Film)": _int8_preset("full", "_qat_film"),
}


class PinnedFrame:
    """_PinnedMpvFrame (feeders.py:38-70): a pinned host RGB48 frame whose ring slot is released
    after the sink has consumed it."""

    def __init__(self, worker, slot, host_ptr, shape):
        self._w, self._slot, self._host, self._shape = worker, slot, host_ptr, shape
        self._ready_waited = self._released = False

    def wait_ready(self):
        if not self._ready_waited:
            p = self._w._processor
            p._chk(p._lib.hdrtv_ring_wait(p._ctx, self._slot), "hdrtv_ring_wait")
            self._ready_waited = True

    def buffer_view(self):
        self.wait_ready()
        n = int(np.prod(self._shape))
        return memoryview((C.c_uint16 * n).from_address(self._host)).cast("B")

    def numpy(self):
        self.wait_ready()
        n = int(np.prod(self._shape))
        return np.ctypeslib.as_array((C.c_uint16 * n).from_address(self._host)).reshape(self._shape)

    def release(self):
        if self._released:
            return
        self._released = True
        try:
            self.wait_ready()
        finally:
            p = self._w._processor
            if p is not None:
                p._lib.hdrtv_ring_release(p._ctx, self._slot)


class HostFrame:
    """What ``_tensor_to_rgb48_bytes`` returns when no ring slot frees up within 250 ms (feeders.py:209-235): the frame
    in the single blocking pinned buffer, already complete; same surface as ``PinnedFrame`` (the reference returns
    ``bytes`` there, which its sink treats like a released frame)."""

    def __init__(self, array):
        self._a = array

    def wait_ready(self):
        return None

    def buffer_view(self):
        return memoryview(self._a).cast("B")

    def numpy(self):
        return self._a

    def release(self):
        return None


class HeadlessPipelineWorker:
    def __init__(self, weights_dir, use_hg=True, proc_w=1920, proc_h=1080, hg_weights=None,
                 status_cb=None, buffer_frames=1):
        self._weights_dir = weights_dir
        self._use_hg = bool(use_hg)
        self._hg_override = hg_weights
        self._proc_w, self._proc_h = int(proc_w), int(proc_h)
        self._processor = None
        self._precision_key = None
        self.status_messages = []
        self._status_cb = status_cb
        self._video_playback_buffer_frames = int(buffer_frames)
        self._pool, self._pool_key, self._pool_idx = None, None, 0
        self._timing_events = None
        self._hdr_queue = None
        self._hdr_thread = None
        self._hdr_stop = threading.Event()
        self._ring_shape = None
        self._hdr_sink = None              # remembered so that a hot-swap (_load_model) can restart the feeder
        self._hdr_error = None             # exception that ended the feeder thread, re-raised by _process_frame
        self._fallback = None              # (pinned host u16 tensor, device u16 tensor) of the ring-exhaustion fallback
        self.ring_fallbacks = 0

    # ---------------------------------------------------------------- status
    def _emit(self, msg):
        self.status_messages.append(msg)
        if self._status_cb:
            self._status_cb(msg)

    # ---------------------------------------------------------------- model load / swap
    @staticmethod
    def _silent_warmup(processor, w, h):
        """model.py:39-70: prime the runtime once (no compiled graph exists here)."""
        if getattr(processor, "_compiled", False):
            processor.warmup_compile(w, h)
        else:
            processor.process(np.zeros((max(1, int(h)), max(1, int(w)), 3), dtype=np.uint8))
            torch.cuda.synchronize()

    def _load_model(self, key, announce_ready=True, *, compile_model=None, force_compile=False,
                    compile_mode=None, warmup=True):
        """model.py:72-257.  Returns True on success; on any backend failure emits
        ``ERROR: model backend failed - ...`` and returns False (model.py:229-235)."""
        cfg = PRECISIONS.get(key, {})
        if not cfg:
            self._emit(f"ERROR: precision preset is not defined - {key}")
            return False
        path = cfg["model"] if self._use_hg else cfg["model_nohg"]
        if not (isinstance(path, dict) or os.path.isabs(str(path))):
            path = os.path.join(self._weights_dir, path)
        if not isinstance(path, dict):
            if not os.path.isfile(path) and os.path.isfile(os.path.splitext(path)[0] + ".hdrw"):
                path = os.path.splitext(path)[0] + ".hdrw"
            if not os.path.isfile(path):
                self._emit(f"ERROR: weights not found - {path}")
                return False
        cw, ch = self._proc_w, self._proc_h
        if announce_ready:
            self._emit(f"Loading model: {key} ...")
        sink = self._hdr_sink                  # a running feeder is restarted on the new processor (hot-swap)
        if self._processor is not None:
            if not self._stop_hdr_feeder(keep_sink=True):
                self._emit("ERROR: HDR feeder did not stop; keeping the current model")
                return False
            self._processor.close()
            self._processor = None
            self._ring_shape = None            # the pinned ring lived in the closed context
            self._fallback = None
            torch.cuda.empty_cache()
        try:
            hg = self._hg_override
            if hg is None and self._use_hg:
                cand = os.path.join(self._weights_dir, cfg.get("hg_weights", ""))
                hg = cand if os.path.isfile(cand) else None        # missing -> no-HG, as model.py:208-209
            self._processor = HDRTVNetMI355X(
                path, device="auto", precision=cfg["precision"], compile_model=bool(compile_model),
                force_compile=bool(force_compile), compile_mode=compile_mode or "default",
                hg_weights=hg, use_hg=self._use_hg, warmup_passes=0)
        except Exception as exc:  # noqa: BLE001  (the reference catches everything here)
            self._processor = None
            self._emit(f"ERROR: model backend failed - {exc}")
            print(f"ERROR: model backend failed: {exc}", file=sys.stderr)
            torch.cuda.empty_cache()
            return False
        if announce_ready and warmup:
            self._emit(f"Priming model for {cw}x{ch} ({key}) ...")
        if warmup:
            self._silent_warmup(self._processor, cw, ch)
        self._precision_key = key
        if sink is not None:
            self._start_hdr_feeder(sink)
        if announce_ready:
            self._emit(f"Ready - {key} [MI355X]")
        return True

    # ---------------------------------------------------------------- per-frame
    def _prepare_hdr_output_tensor(self, raw_out, lower_res_processing=False):
        return raw_out[0] if isinstance(raw_out, (tuple, list)) else raw_out

    def _stage_hdr_display_tensor(self, prepared_out, use_cuda=True):
        """frame_processing.py:118-156: copy the backend-owned output into a rotating pool so the
        feeder never reads a tensor the next infer() overwrites."""
        pool_size = max(2, min(16, self._video_playback_buffer_frames + 2))
        key = (tuple(prepared_out.shape), str(prepared_out.device), str(prepared_out.dtype), pool_size)
        if self._pool_key != key or not self._pool:
            self._pool = [torch.empty_like(prepared_out) for _ in range(pool_size)]
            self._pool_key, self._pool_idx = key, 0
        staged = self._pool[self._pool_idx % len(self._pool)]
        self._pool_idx = (self._pool_idx + 1) % len(self._pool)
        staged.copy_(prepared_out, non_blocking=True)
        return staged

    def _cuda_timing_events(self):
        if self._timing_events is None:
            self._timing_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        return self._timing_events

    def _process_frame(self, *, frame, frame_idx, present_t=None, out_w=None, out_h=None, proc_w=None, proc_h=None,
                       lower_res_processing=False, mpv_w=True, use_cuda=True, need_compare_frame=False):
        """frame_processing.py:168-331 (SDR->HDR branch).  Returns the reference's 5-tuple
        ``(display_frame, output, prepared_out, need_hdr_cpu, model_latency_ms)``; with a sink
        attached (``mpv_w`` truthy and the HDR feeder running) the staged tensor and its ready
        event are queued as ``(present_t, tensor, event)``."""
        if self._processor is None:
            raise RuntimeError("no model loaded")
        if self._hdr_error is not None:        # the feeder thread died: surface it instead of silently dropping frames
            err, self._hdr_error = self._hdr_error, None
            raise RuntimeError(f"HDR feeder failed: {err!r}") from err
        start, end = self._cuda_timing_events()
        start.record(torch.cuda.current_stream())
        pw, ph = int(proc_w or self._proc_w), int(proc_h or self._proc_h)
        with torch.inference_mode():
            if frame.shape[1] != pw or frame.shape[0] != ph:
                # the reference letterboxes on the host with cv2 before preprocess (gui_export.py:1080,
                # gui_scaling.py:228-244); here the resize runs on the device
                tensor, cond = self._processor.preprocess_letterboxed(frame, pw, ph)
            else:
                tensor, cond = self._processor.preprocess(frame)
            raw_out = self._processor.infer((tensor, cond))
        end.record(torch.cuda.current_stream())
        end.synchronize()
        model_latency_ms = max(0.0, float(start.elapsed_time(end)))
        prepared_out = self._prepare_hdr_output_tensor(raw_out, lower_res_processing)
        if mpv_w and self._hdr_queue is not None:
            staged = self._stage_hdr_display_tensor(prepared_out, use_cuda)
            ready = torch.cuda.Event(enable_timing=False)
            ready.record(torch.cuda.current_stream())
            self._queue_display_item(self._hdr_queue, (present_t, staged, ready))
        need_hdr_cpu = not mpv_w
        output = self._processor.postprocess(prepared_out) if need_hdr_cpu else frame
        return None, output, prepared_out, need_hdr_cpu, model_latency_ms

    @staticmethod
    def _queue_display_item(q, item):
        try:
            q.put(item, timeout=0.25)
        except _queue.Full:
            try:
                q.get_nowait()          # drop the oldest frame rather than stall inference
            except _queue.Empty:
                pass
            q.put_nowait(item)

    # ---------------------------------------------------------------- RGB48 ring + feeder
    def _tensor_to_rgb48_bytes(self, tensor, stream=None):
        """feeders.py:193-249, GPU branch: quantise to RGB48 directly into a pinned ring slot and
        return a ``PinnedFrame`` guarded by the slot's ready event.  Ring exhaustion (no slot free
        within 250 ms, feeders.py:166-167) falls back to one blocking pinned buffer as the reference
        does (209-235): convert, copy, synchronise the stream, hand the finished frame over."""
        p = self._processor
        t = tensor[0] if isinstance(tensor, (tuple, list)) else tensor
        h, w = int(t.shape[-2]), int(t.shape[-1])
        if self._ring_shape != (h, w):
            p._chk(p._lib.hdrtv_ring_create(p._ctx, _RING_FRAMES, h, w), "hdrtv_ring_create")
            self._ring_shape = (h, w)
        host, dev = C.c_void_p(), C.c_void_p()
        st = stream or torch.cuda.current_stream(p.device)
        sp = C.c_void_p(st.cuda_stream)
        dt = _L.F32 if t.dtype == torch.float32 else _L.F16
        slot = p._lib.hdrtv_ring_acquire(p._ctx, 250, C.byref(host), C.byref(dev))
        if slot == _L.ESTATE:
            self.ring_fallbacks += 1
            if self._fallback is None or tuple(self._fallback[0].shape) != (h, w, 3):
                self._fallback = (torch.empty((h, w, 3), dtype=torch.uint16, pin_memory=True),
                                  torch.empty((h, w, 3), dtype=torch.uint16, device=p.device))
            fb_host, fb_dev = self._fallback
            p._chk(p._lib.hdrtv_post_rgb48(p._ctx, sp, t.contiguous().data_ptr(), dt, h, w, fb_dev.data_ptr()), "hdrtv_post_rgb48")
            with torch.cuda.stream(st):
                fb_host.copy_(fb_dev, non_blocking=True)
            st.synchronize()
            return HostFrame(fb_host.numpy().copy())       # the reference's host_np.tobytes(): a private copy
        p._chk(slot, "hdrtv_ring_acquire")
        p._chk(p._lib.hdrtv_post_rgb48(p._ctx, sp, t.contiguous().data_ptr(), dt, h, w, dev), "hdrtv_post_rgb48")
        p._chk(p._lib.hdrtv_ring_commit(p._ctx, slot, sp), "hdrtv_ring_commit")
        return PinnedFrame(self, slot, host.value, (h, w, 3))

    def _start_hdr_feeder(self, sink):
        """feeders.py:632-657 + 440-496: a thread that waits for each frame's ready event,
        converts on a side stream into the pinned ring and hands the frame to ``sink``."""
        self._stop_hdr_feeder()
        self._hdr_sink = sink
        self._hdr_error = None
        if self._processor is not None and self._ring_shape != (self._proc_h, self._proc_w):
            # pin the ring now (3 x 50 MB at 4K takes tens of ms) rather than inside the first frame's deadline
            p = self._processor
            p._chk(p._lib.hdrtv_ring_create(p._ctx, _RING_FRAMES, self._proc_h, self._proc_w), "hdrtv_ring_create")
            self._ring_shape = (self._proc_h, self._proc_w)
        # feeders.py:634-644: queue depth = buffer_frames (1..3).  With the staging pool of buffer_frames + 2
        # (_stage_hdr_display_tensor) that is exactly enough: queued + the one in the feeder's hands + the one
        # being staged; one more queued frame and the main stream overwrites a tensor the side stream still reads.
        self._hdr_queue = _queue.Queue(maxsize=min(3, max(1, self._video_playback_buffer_frames)))
        self._hdr_stop.clear()
        dev = self._processor.device

        hq = self._hdr_queue

        def run():
            try:
                side = torch.cuda.Stream(device=dev)
                while not self._hdr_stop.is_set():
                    try:
                        item = hq.get(timeout=0.05)
                    except _queue.Empty:
                        continue
                    if item is None:
                        break
                    present_t, tensor, ready = item
                    ready.synchronize()                      # cross-thread device sync (feeders.py:469-473)
                    with torch.cuda.stream(side):
                        payload = self._tensor_to_rgb48_bytes(tensor, side)
                    if present_t is not None:
                        delay = present_t - time.perf_counter()
                        if delay > 0:
                            time.sleep(delay)
                    sink(payload)
            except BaseException as exc:  # noqa: BLE001  (a dead daemon thread must not be silent: _process_frame re-raises)
                self._hdr_error = exc

        self._hdr_thread = threading.Thread(target=run, name="hdr-feeder", daemon=True)
        self._hdr_thread.start()

    def _stop_hdr_feeder(self, keep_sink=False):
        """Stops the feeder thread.  Returns False (and leaves everything in place) if the thread is still running after
        10 s: the processor must not be closed under a thread that may be inside the C library."""
        if self._hdr_thread is not None:
            self._hdr_stop.set()
            try:
                self._hdr_queue.put_nowait(None)
            except Exception:  # noqa: BLE001
                pass
            self._hdr_thread.join(timeout=10.0)
            if self._hdr_thread.is_alive():
                # the stop flag and the sentinel are already set: the thread exits as soon as it unblocks and nothing will
                # drain the queue afterwards -- make the next _process_frame raise instead of queueing into the void
                self._hdr_error = RuntimeError("HDR feeder did not stop within 10 s; frames are no longer being delivered")
                return False
        self._hdr_thread, self._hdr_queue = None, None
        if not keep_sink:
            self._hdr_sink = None
        return True

    def close(self):
        if not self._stop_hdr_feeder():
            raise RuntimeError("HDR feeder thread did not stop; not destroying the processor under it")
        if self._processor is not None:
            self._processor.close()
            self._processor = None


def shard_frames(n_frames: int, rank: int, world_size: int):
    """Frame-parallel partition of BASELINE.json configs[3]: frame i -> GPU (rank) i mod N.
    No data-path collective; order is restored on the host by frame index."""
    return list(range(rank, n_frames, world_size))
