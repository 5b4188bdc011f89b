"""Headless realtime playback around ``HeadlessPipelineWorker``.

  pacing loop, catch-up drop, fps limiter   src/gui_pipeline_worker.py:860-936, 1024-1029 (constants 38-40)
  hot-swap of precision / resolution        src/gui_pipeline_worker.py:737-759
  display prebuffer                         src/gui_pipeline_worker.py:516-530, 1082-1088
  live metrics dict                         src/gui_pipeline_worker_runtime_metrics.py:16-26, 120-199
  CSV schema, 1 % low fps                   src/cli_playback_benchmark.py:278-313, 1123-1129
  rawvideo rgb48le wire format              src/gui_mpv_widget.py:950-960, 998 ; src/gui_export.py:940-1015

No PyQt / cv2 / mpv: Qt signals become plain callbacks, the frame source is any object with
``read() -> (ok, frame_bgr_u8)`` and ``fps``, the HDR sink any callable taking a ``PinnedFrame``.
The clock and sleep functions are injectable so the scheduling logic is testable without a GPU.
"""
from __future__ import annotations

import csv
import os
import queue
import threading
import time

import numpy as np

from .weights import synthetic_frame

# gui_pipeline_worker.py:38-40
_REALTIME_CATCHUP_ENABLED = True
_REALTIME_SKIP_LAG_FRAMES = 1.1
_REALTIME_MAX_CATCHUP_SKIP = 6

# cli_playback_benchmark.py:278-313
CSV_FIELDS = [
    "elapsed_s", "logged_at_local", "fps", "latency_ms", "model_latency_ms", "live_video_latency_ms", "frame",
    "cpu_mb", "gpu_mb", "model_mb", "model_size_label", "precision", "proc_res", "psnr_db", "sssim", "delta_e_itp",
    "hdr_vdp3", "objective_enabled", "objective_note", "hdr_vdp3_note", "is_live_capture", "decode_ms", "resize_ms",
    "infer_ms", "pre_ms", "run_ms", "post_ms", "render_ms", "fps_1p_low", "dropped_frames", "catchup_dropped_frames",
    "fps_limiter_dropped_frames", "source_loops", "playback_mode",
]


# ------------------------------------------------------------------------------- sources
class SyntheticSource:
    """Seeded u8 BGR frames (a small pool cycled, so decode cost stays out of the loop)."""

    def __init__(self, width, height, fps=60.0, n_frames=120, seed=1234, pool=4, kind="gradient"):
        self.width, self.height, self.fps, self.frame_count = int(width), int(height), float(fps), int(n_frames)
        self._pool = [synthetic_frame(self.height, self.width, seed + i, kind) for i in range(max(1, pool))]
        self._i = 0

    def read(self):
        if self._i >= self.frame_count:
            return False, None
        f = self._pool[self._i % len(self._pool)]
        self._i += 1
        return True, f

    def release(self):
        pass


class PinnedFrame(np.ndarray):
    """A u8 BGR frame that lives in page-locked memory; ``pinned_tensor`` is the torch tensor sharing it
    (HDRTVNetMI355X.preprocess uploads such a frame without the host-side copy into its own pinned slot).  When the
    prefetcher has already uploaded it, ``device_tensor`` is the device copy and ``ready_event`` the event recorded
    behind that upload on the prefetcher's stream."""
    pinned_tensor = None
    device_tensor = None
    ready_event = None


class PinnedPrefetch:
    """Source wrapper: a reader thread pulls frames from ``source`` and copies them into a small pool of page-locked
    buffers while the previous frame is on the GPU, so the 25 MB host copy of a 4K frame (2-3 ms) leaves the
    per-frame critical path.  The reference copies into its pinned slot inside preprocess, on the caller's thread
    (hdrtvnet_torch.py:2262-2266); here the HIP path hands the buffer over by hipMemcpyAsync + the stream order.
    At most one frame is queued and one is being filled, so a pool of three is never overwritten while in use
    (``_process_frame`` returns only after the frame's work has completed)."""

    def __init__(self, source, pool=3, upload=True):
        self._src = source
        self._upload = bool(upload)       # also hipMemcpyAsync the frame to the device on a side stream (+ hipEvent)
        self._dev = {}
        self._stream = None
        self._error = None
        self.width, self.height, self.fps = source.width, source.height, source.fps
        self.frame_count = getattr(source, "frame_count", 0)
        self._pool_n = max(3, int(pool))
        self._bufs = {}
        self._q = queue.Queue(maxsize=1)
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, name="hdrtv-pinned-prefetch", daemon=True)
        self._t.start()

    def _stage(self, frame, i):
        import ctypes
        import torch
        key = frame.shape
        pool = self._bufs.get(key)
        if pool is None:
            try:
                pool = [torch.empty(key, dtype=torch.uint8, pin_memory=True) for _ in range(self._pool_n)]
            except RuntimeError:          # no HIP runtime in this process (CPU-only host): frames pass through unstaged
                pool = []
            self._bufs[key] = pool
        if not pool:
            return frame
        t = pool[i % self._pool_n]
        src = np.ascontiguousarray(frame)
        ctypes.memmove(t.data_ptr(), src.ctypes.data, src.nbytes)      # plain memcpy, GIL released
        out = t.numpy().view(PinnedFrame)
        out.pinned_tensor = t
        if self._upload and torch.cuda.is_available():
            dpool = self._dev.get(key)
            if dpool is None:
                dpool = self._dev[key] = [torch.empty(key, dtype=torch.uint8, device="cuda") for _ in range(self._pool_n)]
                self._stream = self._stream or torch.cuda.Stream()
            d = dpool[i % self._pool_n]
            with torch.cuda.stream(self._stream):
                d.copy_(t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            out.device_tensor, out.ready_event = d, ev
        return out

    def _run(self):
        try:
            self._loop()
        except BaseException as exc:  # noqa: BLE001  (handed to the consumer: read() re-raises it)
            self._error = exc

    def _loop(self):
        i = 0
        while not self._stop.is_set():
            ret, frame = self._src.read()
            item = (False, None)
            if ret and frame is not None and frame.dtype == np.uint8 and frame.ndim == 3:
                item = (True, self._stage(frame, i))
                i += 1
            elif ret:
                item = (ret, frame)
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue
            if not item[0]:
                return

    def read(self):
        while True:
            try:
                return self._q.get(timeout=0.05 if not self._t.is_alive() else 0.5)
            except queue.Empty:
                if not self._t.is_alive() and self._q.empty():
                    if self._error is not None:
                        err, self._error = self._error, None
                        raise RuntimeError("frame source failed in the prefetch thread") from err
                    return False, None

    def release(self):
        self._stop.set()
        self._t.join(timeout=2.0)
        self._src.release()


class RawVideoSource:
    """Headerless ``bgr24`` rawvideo file (what ``ffmpeg -f rawvideo -pix_fmt bgr24`` writes), memory-mapped."""

    def __init__(self, path, width, height, fps):
        self.width, self.height, self.fps = int(width), int(height), float(fps)
        fb = self.width * self.height * 3
        size = os.path.getsize(path)
        if size < fb or size % fb:
            raise ValueError(f"{path}: size {size} is not a whole number of {self.width}x{self.height} bgr24 frames")
        self.frame_count = size // fb
        self._mm = np.memmap(path, dtype=np.uint8, mode="r", shape=(self.frame_count, self.height, self.width, 3))
        self._i = 0

    def read(self):
        if self._i >= self.frame_count:
            return False, None
        f = np.ascontiguousarray(self._mm[self._i])
        self._i += 1
        return True, f

    def release(self):
        self._mm = None


# ------------------------------------------------------------------------------- rgb48le sink
class Rgb48leSink:
    """Writes each RGB48 frame as headerless little-endian ``rgb48le`` rawvideo -- the byte stream the
    reference pipes into mpv (gui_mpv_widget.py:950-960) and into ffmpeg on export (gui_export.py:957-975)
    -- to a file object / path / pipe, then releases the frame's ring slot (gui_mpv_widget.py:663-668)."""

    def __init__(self, target, width, height, fps):
        self.width, self.height, self.fps = int(width), int(height), float(fps)
        self._own = isinstance(target, (str, os.PathLike))
        self._f = open(target, "wb") if self._own else target
        self.frames = 0
        self.bytes = 0

    def __call__(self, payload):
        try:
            view = payload.buffer_view()
            if len(view) != self.width * self.height * 6:
                raise ValueError(f"frame of {len(view)} bytes does not match {self.width}x{self.height} rgb48le")
            self._f.write(view)
            self.frames += 1
            self.bytes += len(view)
        finally:
            payload.release()

    def close(self):
        if self._f is not None:
            self._f.flush()
            if self._own:
                self._f.close()
            self._f = None

    def mpv_args(self):
        """Command-line form of the demuxer options of gui_mpv_widget.py:955-960 and the HDR tag of 998."""
        return ["--demuxer=rawvideo", f"--demuxer-rawvideo-w={self.width}", f"--demuxer-rawvideo-h={self.height}",
                "--demuxer-rawvideo-mp-format=rgb48le", f"--demuxer-rawvideo-fps={self.fps:g}",
                "--vf=format=colorlevels=full:primaries=bt.2020:gamma=pq"]

    def ffmpeg_input_args(self):
        """Input half of the export command (gui_export.py:957-977): full-range BT.2020 / PQ rgb48le on stdin."""
        return ["-f", "rawvideo", "-pix_fmt", "rgb48le", "-s:v", f"{self.width}x{self.height}", "-r", f"{self.fps:.6f}",
                "-color_range", "pc", "-colorspace", "bt2020nc", "-color_trc", "smpte2084", "-color_primaries", "bt2020",
                "-i", "-"]


# ------------------------------------------------------------------------------- metrics
def trimmed_latency_average(values) -> float:
    """runtime_metrics.py:16-26."""
    vals = [float(v) for v in values if float(v) > 0.0]
    if not vals:
        return 0.0
    if len(vals) < 8:
        return sum(vals) / len(vals)
    vals.sort()
    trim = max(1, len(vals) // 10)
    kept = vals[trim:-trim] if len(vals) > (trim * 2) else vals
    return sum(kept) / len(kept)


def one_percent_low(fps_samples, avg_fps):
    """cli_playback_benchmark.py:1123-1129."""
    s = sorted(fps_samples)
    if not s:
        return avg_fps
    k = max(1, int(len(s) * 0.01))
    return sum(s[:k]) / k


class CsvLogger:
    def __init__(self, path):
        self._f = open(path, "w", newline="")
        self._w = csv.DictWriter(self._f, fieldnames=CSV_FIELDS, extrasaction="ignore")
        self._w.writeheader()
        self._t0 = time.perf_counter()

    def log(self, metrics):
        row = {k: metrics.get(k, "") for k in CSV_FIELDS}
        row["elapsed_s"] = f"{time.perf_counter() - self._t0:.3f}"
        row["logged_at_local"] = time.strftime("%Y-%m-%d %H:%M:%S")
        self._w.writerow(row)

    def close(self):
        self._f.close()


# ------------------------------------------------------------------------------- the loop
class RealtimePlayback:
    """The worker's ``run()`` loop for a file-like source (not live capture), headless.

    ``worker`` needs ``_process_frame(frame=, frame_idx=, present_t=, proc_w=, proc_h=, mpv_w=)`` returning the
    reference's 5-tuple, ``_load_model(key)``, ``_silent_warmup(processor, w, h)``, ``_processor``, ``_precision_key``,
    ``_proc_w/_proc_h`` -- i.e. a ``HeadlessPipelineWorker`` (or a stand-in in tests)."""

    def __init__(self, worker, source, *, sink=True, frame_stride=1, realtime=True, metrics_cb=None, csv_path=None,
                 metrics_interval_s=0.20, metrics_window=120, clock=time.perf_counter, sleep=None, status_cb=None,
                 gt_source=None, objective_every=10):
        self.w, self.source = worker, source
        self.sink = sink
        self.frame_stride = max(1, int(frame_stride))
        self.realtime = bool(realtime)
        self.metrics_cb, self.status_cb = metrics_cb, status_cb
        self._csv = CsvLogger(csv_path) if csv_path else None
        self._metrics_interval_s = max(0.05, float(metrics_interval_s))
        self._window = int(metrics_window)
        self._clock = clock
        self._sleep = sleep or self._sleep_until_default
        # optional ground-truth HDR frames ([3,H,W] or [1,3,H,W] unit-range tensors from ``gt_source.read()``): every
        # ``objective_every``-th processed frame is scored on the device (gui_pipeline_worker_objective.py role)
        self.gt_source, self.objective_every = gt_source, max(1, int(objective_every))
        self._objective = {"psnr_db": None, "sssim": None, "delta_e_itp": None}
        self._pending_precision = None
        self._pending_resolution = None
        self._display_prebuffer_target = 0
        self._display_prebuffer_count = 0
        self.prebuffer_ready = None          # callback(frame_idx, buffered)
        self._stop = False
        # counters (CSV columns)
        self.realtime_drop_frames = 0
        self.fps_limiter_dropped_frames = 0
        self.frames_processed = 0
        self.last_metrics = None

    # -- control surface (gui_pipeline_worker.py:516-530 and the pending_* slots)
    def request_precision(self, key):
        self._pending_precision = key

    def request_resolution(self, w, h):
        self._pending_resolution = (int(w), int(h))

    def request_display_prebuffer(self, frames):
        self._display_prebuffer_target = max(0, int(frames))
        self._display_prebuffer_count = 0

    def cancel_display_prebuffer(self):
        self._display_prebuffer_target = 0
        self._display_prebuffer_count = 0

    def stop(self):
        self._stop = True

    def _sleep_until_default(self, t):
        while True:
            d = t - self._clock()
            if d <= 0:
                return
            time.sleep(min(d, 0.002) if d < 0.004 else d - 0.002)

    def _status(self, msg):
        if self.status_cb:
            self.status_cb(msg)

    # -- the loop
    def run(self, max_frames=None):
        w = self.w
        frame_interval_s = 1.0 / max(1e-6, float(self.source.fps))
        proc_w, proc_h = int(w._proc_w), int(w._proc_h)
        frame_idx = 0
        frame_times, model_times, presented_times, fps_samples = [], [], [], []
        next_frame_t = self._clock()
        t_start = next_frame_t
        last_emit_t = 0.0
        while not self._stop:
            # hot-swap precision / processing resolution between frames (737-759)
            pending = self._pending_precision
            if pending and pending != w._precision_key:
                self._pending_precision = None
                if not w._load_model(pending):
                    continue
            elif pending:
                self._pending_precision = None
            if self._pending_resolution is not None:
                new_pw, new_ph = self._pending_resolution
                self._pending_resolution = None
                if (new_pw, new_ph) != (proc_w, proc_h):
                    self._status(f"Switching to {new_pw}x{new_ph} ...")
                    w._proc_w, w._proc_h = new_pw, new_ph
                    proc_w, proc_h = new_pw, new_ph
                    w._silent_warmup(w._processor, proc_w, proc_h)
                    self._status(f"Ready - {w._precision_key} @ {proc_w}x{proc_h}")

            now = self._clock()
            lag_s = 0.0
            if self.realtime:
                if now < next_frame_t:
                    self._sleep(next_frame_t)
                    now = self._clock()
                else:
                    lag_s = now - next_frame_t

            ret, frame = self.source.read()
            if not ret:
                break
            frame_idx += 1

            # real-time catch-up: behind the wall clock -> drop decoded frames, process the newest (899-936)
            if self.realtime and _REALTIME_CATCHUP_ENABLED and lag_s > frame_interval_s * _REALTIME_SKIP_LAG_FRAMES:
                skip_n = min(_REALTIME_MAX_CATCHUP_SKIP, max(0, int(lag_s / frame_interval_s)))
                while skip_n > 0:
                    ret_skip, frame_skip = self.source.read()
                    if not ret_skip:
                        ret = False
                        break
                    frame = frame_skip
                    frame_idx += 1
                    self.realtime_drop_frames += 1
                    next_frame_t += frame_interval_s
                    skip_n -= 1
                if not ret:
                    break

            # fps limiter via frame skipping (keeps wall-clock speed)
            if self.frame_stride > 1 and (frame_idx % self.frame_stride) != 0:
                next_frame_t += frame_interval_s
                self.fps_limiter_dropped_frames += 1
                continue

            present_t = max(next_frame_t, self._clock()) if self.realtime else None
            t0 = self._clock()
            _, _, prepared, _, model_latency_ms = w._process_frame(frame=frame, frame_idx=frame_idx, present_t=present_t,
                                                                    proc_w=proc_w, proc_h=proc_h, mpv_w=self.sink)
            if self.gt_source is not None:
                ok_gt, gt = self.gt_source.read()
                if ok_gt and prepared is not None and (self.frames_processed % self.objective_every) == 0:
                    self._objective = w._processor.objective_metrics(prepared, gt)
            t1 = self._clock()
            next_frame_t += frame_interval_s
            frame_ms = (t1 - t0) * 1000.0
            frame_times.append(frame_ms)
            if frame_ms > 0:
                fps_samples.append(1000.0 / frame_ms)
            if model_latency_ms > 0.0:
                model_times.append(float(model_latency_ms))
            presented_times.append(max(t1, present_t) if present_t is not None else t1)
            for lst in (frame_times, model_times, presented_times):
                if len(lst) > self._window:
                    del lst[: len(lst) - self._window]
            self.frames_processed += 1

            if self._display_prebuffer_target > 0:
                self._display_prebuffer_count += 1
                if self._display_prebuffer_count >= self._display_prebuffer_target:
                    buffered = int(self._display_prebuffer_count)
                    self._display_prebuffer_target = 0
                    self._display_prebuffer_count = 0
                    if self.prebuffer_ready:
                        self.prebuffer_ready(frame_idx, buffered)

            now_t = self._clock()
            if last_emit_t <= 0.0 or (now_t - last_emit_t) >= self._metrics_interval_s:
                last_emit_t = now_t
                self._emit_metrics(frame_idx, frame_times, model_times, presented_times, fps_samples, proc_w, proc_h)
            if max_frames is not None and self.frames_processed >= max_frames:
                break
        if frame_times:
            self._emit_metrics(frame_idx, frame_times, model_times, presented_times, fps_samples, proc_w, proc_h)
        if self._csv:
            self._csv.close()
            self._csv = None
        elapsed = self._clock() - t_start
        return {"frames_processed": self.frames_processed, "frames_read": frame_idx, "elapsed_s": elapsed,
                "catchup_dropped_frames": self.realtime_drop_frames,
                "fps_limiter_dropped_frames": self.fps_limiter_dropped_frames,
                "fps": (self.frames_processed / elapsed) if elapsed > 0 else 0.0, "last_metrics": self.last_metrics}

    def _emit_metrics(self, frame_idx, frame_times, model_times, presented_times, fps_samples, proc_w, proc_h):
        avg = sum(frame_times) / len(frame_times)
        model_avg = (sum(model_times) / len(model_times)) if model_times else 0.0
        if len(presented_times) >= 2:
            dt = presented_times[-1] - presented_times[0]
            fps = ((len(presented_times) - 1) / dt) if dt > 0 else 0.0
        else:
            fps = 1000.0 / avg if avg > 0 else 0.0
        cpu_mb = gpu_mb = 0.0
        try:
            import psutil
            cpu_mb = psutil.Process().memory_info().rss / (1024 * 1024)
        except Exception:  # noqa: BLE001
            pass
        try:
            import torch
            if torch.cuda.is_available():
                gpu_mb = float(torch.cuda.memory_reserved() / (1024 * 1024)) or float(torch.cuda.memory_allocated() / (1024 * 1024))
        except Exception:  # noqa: BLE001
            pass
        m = {
            "fps": fps, "latency_ms": avg, "model_latency_ms": float(model_avg),
            "model_latency_display_ms": float(trimmed_latency_average(model_times)), "live_video_latency_ms": 0.0,
            "is_live_capture": False, "frame": frame_idx, "cpu_mb": cpu_mb, "gpu_mb": gpu_mb,
            "model_mb": float(getattr(self.w, "_model_mb", 0.0) or 0.0), "model_size_label": "Checkpoint",
            "precision": self.w._precision_key, "proc_res": f"{proc_w}x{proc_h}",
            "psnr_db": self._objective["psnr_db"], "sssim": self._objective["sssim"],
            "delta_e_itp": self._objective["delta_e_itp"], "hdr_vdp3": None, "objective_enabled": self.gt_source is not None,
            "objective_note": "", "hdr_vdp3_note": "",
            # CSV-only columns (cli_playback_benchmark.py)
            "infer_ms": float(model_avg), "fps_1p_low": one_percent_low(fps_samples, fps),
            "dropped_frames": self.realtime_drop_frames + self.fps_limiter_dropped_frames,
            "catchup_dropped_frames": self.realtime_drop_frames,
            "fps_limiter_dropped_frames": self.fps_limiter_dropped_frames, "source_loops": 0,
            "playback_mode": "realtime" if self.realtime else "max-throughput",
        }
        self.last_metrics = m
        if self.metrics_cb:
            self.metrics_cb(m)
        if self._csv:
            self._csv.log(m)


def main(argv=None):
    """``python -m hdrtv_mi355x.playback``: synthetic (or bgr24 rawvideo) source -> worker -> rgb48le sink."""
    import argparse
    import json

    from .worker import HeadlessPipelineWorker

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--weights-dir", required=True, help="directory holding original/HR.pt (or .hdrw) [and original/HG.pt]")
    ap.add_argument("--precision", default="FP16")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--fps", type=float, default=60.0)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--input", help="bgr24 rawvideo file at --size (default: synthetic frames)")
    ap.add_argument("--out", help="rgb48le rawvideo output (file or fifo); omit to run without a display sink")
    ap.add_argument("--no-hg", action="store_true")
    ap.add_argument("--hg-weights", default=None, help="HG weight file, or seeded:<n>")
    ap.add_argument("--max-throughput", action="store_true", help="do not pace to the source clock")
    ap.add_argument("--no-prefetch", action="store_true", help="copy each frame into pinned memory on the processing thread, as the reference does")
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--csv")
    a = ap.parse_args(argv)
    wd, ht = (int(v) for v in a.size.lower().split("x", 1))
    src = RawVideoSource(a.input, wd, ht, a.fps) if a.input else SyntheticSource(wd, ht, a.fps, a.frames)
    if not a.no_prefetch:
        src = PinnedPrefetch(src)
    worker = HeadlessPipelineWorker(a.weights_dir, use_hg=not a.no_hg, proc_w=wd, proc_h=ht, hg_weights=a.hg_weights,
                                    status_cb=lambda m: print(m, flush=True))
    if not worker._load_model(a.precision):
        return 1
    sink = None
    if a.out:
        sink = Rgb48leSink(a.out, wd, ht, a.fps)
        worker._start_hdr_feeder(sink)
    pb = RealtimePlayback(worker, src, sink=bool(sink), frame_stride=a.stride, realtime=not a.max_throughput, csv_path=a.csv)
    res = pb.run(max_frames=a.frames)
    if sink:
        deadline = time.perf_counter() + 5.0
        while sink.frames < res["frames_processed"] and time.perf_counter() < deadline:
            time.sleep(0.01)
        worker._stop_hdr_feeder()
        sink.close()
        res["sink_frames"], res["sink_bytes"] = sink.frames, sink.bytes
    worker.close()
    lm = res.pop("last_metrics") or {}
    res.update({k: lm.get(k) for k in ("latency_ms", "model_latency_ms", "fps_1p_low", "proc_res", "precision")})
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
