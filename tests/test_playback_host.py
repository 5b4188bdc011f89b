"""Host logic of the headless realtime loop (hdrtv_mi355x/playback.py) with a stand-in worker and a fake
clock: pacing, catch-up drop, fps limiter, hot-swap, prebuffer, metrics dict / CSV schema, sources and the
rgb48le sink.  No GPU.  Reference behaviour: gui_pipeline_worker.py:860-936, 737-759, 1082-1088;
gui_pipeline_worker_runtime_metrics.py:16-26, 178-199; cli_playback_benchmark.py:278-313, 1123-1129."""
import csv
import io

import numpy as np
import pytest

from hdrtv_mi355x import playback as P


class FakeClock:
    def __init__(self):
        self.t = 100.0
        self.sleeps = 0

    def __call__(self):
        return self.t

    def sleep_until(self, t):
        if t > self.t:
            self.t = t
            self.sleeps += 1


class FakeWorker:
    def __init__(self, clock, cost_s, w=96, h=64):
        self._clock, self.cost_s = clock, cost_s
        self._proc_w, self._proc_h = w, h
        self._precision_key = "FP16"
        self._processor = object()
        self.loaded, self.warmups, self.frames = [], [], []

    def _load_model(self, key):
        self.loaded.append(key)
        if key == "broken":
            return False
        self._precision_key = key
        return True

    def _silent_warmup(self, processor, w, h):
        self.warmups.append((w, h))

    def _process_frame(self, *, frame, frame_idx, present_t=None, proc_w=None, proc_h=None, mpv_w=True, **_):
        self.frames.append(frame_idx)
        self._clock.t += self.cost_s
        return None, frame, None, False, self.cost_s * 1000.0 * 0.9


def _run(cost_s, n=120, fps=60.0, **kw):
    clk = FakeClock()
    w = FakeWorker(clk, cost_s)
    src = P.SyntheticSource(96, 64, fps=fps, n_frames=n, pool=2, kind="noise")
    pb = P.RealtimePlayback(w, src, clock=clk, sleep=clk.sleep_until, **kw)
    return pb, w, clk, pb.run()


def test_faster_than_source_sleeps_and_drops_nothing():
    pb, w, clk, r = _run(cost_s=0.004, n=120, fps=60.0)
    assert r["frames_processed"] == 120 and r["catchup_dropped_frames"] == 0 and w.frames == list(range(1, 121))
    assert clk.sleeps >= 118                                   # paced to the source clock
    assert abs(r["elapsed_s"] - 120 / 60.0) < 2 / 60.0        # wall-clock speed of the source
    assert abs(r["last_metrics"]["fps"] - 60.0) < 0.5


def test_slower_than_source_catches_up_by_dropping():
    # 40 ms per frame against a 16.7 ms source: lag > 1.1 intervals -> skip min(6, int(lag / interval)) frames
    pb, w, clk, r = _run(cost_s=0.040, n=600, fps=60.0)
    assert r["catchup_dropped_frames"] > 0
    # (the frame in hand when the source ends inside a catch-up skip is neither processed nor a drop)
    assert 0 <= r["frames_read"] - r["frames_processed"] - r["catchup_dropped_frames"] <= 1
    gaps = np.diff(w.frames)
    assert gaps.max() <= 1 + P._REALTIME_MAX_CATCHUP_SKIP
    # cadence is preserved: the run takes about as long as the source clip, not 600 * 40 ms
    assert abs(r["elapsed_s"] - 600 / 60.0) < 0.25
    assert r["last_metrics"]["catchup_dropped_frames"] == r["catchup_dropped_frames"]
    assert r["last_metrics"]["playback_mode"] == "realtime"


def test_lag_just_under_threshold_is_not_dropped():
    # 1.05 intervals of work: lag settles at 0.05 interval per frame and accumulates; the first frames must not drop
    clk = FakeClock()
    w = FakeWorker(clk, (1 / 60.0) * 1.0)
    src = P.SyntheticSource(96, 64, fps=60.0, n_frames=30, pool=1, kind="noise")
    r = P.RealtimePlayback(w, src, clock=clk, sleep=clk.sleep_until).run()
    assert r["catchup_dropped_frames"] == 0 and r["frames_processed"] == 30


def test_max_throughput_mode_never_sleeps():
    pb, w, clk, r = _run(cost_s=0.004, n=50, realtime=False)
    assert clk.sleeps == 0 and r["frames_processed"] == 50
    assert abs(r["last_metrics"]["fps"] - 250.0) < 5.0 and r["last_metrics"]["playback_mode"] == "max-throughput"


def test_fps_limiter_stride():
    pb, w, clk, r = _run(cost_s=0.002, n=100, frame_stride=2)
    assert r["frames_processed"] == 50 and r["fps_limiter_dropped_frames"] == 50
    assert all(i % 2 == 0 for i in w.frames)
    assert abs(r["elapsed_s"] - 100 / 60.0) < 2 / 60.0        # still the source's wall-clock speed


def test_hot_swap_precision_resolution_and_prebuffer():
    clk = FakeClock()
    w = FakeWorker(clk, 0.002)
    src = P.SyntheticSource(96, 64, fps=60.0, n_frames=10, pool=1, kind="noise")
    status, ready = [], []
    pb = P.RealtimePlayback(w, src, clock=clk, sleep=clk.sleep_until, status_cb=status.append)
    pb.prebuffer_ready = lambda f, n: ready.append((f, n))
    pb.request_precision("INT8 Mixed (QAT)")
    pb.request_display_prebuffer(3)
    r = pb.run()
    assert w.loaded == ["INT8 Mixed (QAT)"] and r["last_metrics"]["precision"] == "INT8 Mixed (QAT)"
    assert ready == [(3, 3)]
    # a resolution switch warms the backend up; later frames are processed at the new size (the worker letterboxes
    # frames that arrive at another size on the device)
    src2 = P.SyntheticSource(96, 64, fps=60.0, n_frames=10, pool=1, kind="noise")
    pb2 = P.RealtimePlayback(w, src2, clock=clk, sleep=clk.sleep_until, status_cb=status.append)
    pb2.request_resolution(128, 96)
    r2 = pb2.run()
    assert w.warmups == [(128, 96)] and status[-2:] == ["Switching to 128x96 ...", "Ready - INT8 Mixed (QAT) @ 128x96"]
    assert r2["frames_processed"] == 10 and r2["last_metrics"]["proc_res"] == "128x96" and (w._proc_w, w._proc_h) == (128, 96)


def test_metrics_dict_and_csv_schema(tmp_path):
    path = tmp_path / "log.csv"
    got = []
    pb, w, clk, r = _run(cost_s=0.004, n=60, csv_path=str(path), metrics_cb=got.append, metrics_interval_s=0.1)
    ref_keys = {"fps", "latency_ms", "model_latency_ms", "model_latency_display_ms", "live_video_latency_ms",
                "is_live_capture", "frame", "cpu_mb", "gpu_mb", "model_mb", "model_size_label", "precision", "proc_res",
                "psnr_db", "sssim", "delta_e_itp", "hdr_vdp3", "objective_enabled", "objective_note", "hdr_vdp3_note"}
    assert got and ref_keys <= set(got[-1])
    assert got[-1]["proc_res"] == "96x64" and abs(got[-1]["latency_ms"] - 4.0) < 1e-6
    rows = list(csv.DictReader(open(path)))
    assert list(rows[0].keys()) == P.CSV_FIELDS and len(rows) == len(got)
    assert len(P.CSV_FIELDS) == 34 and P.CSV_FIELDS[0] == "elapsed_s" and P.CSV_FIELDS[-1] == "playback_mode"


def test_latency_statistics():
    assert P.trimmed_latency_average([]) == 0.0
    assert P.trimmed_latency_average([2.0, 4.0, 0.0, -1.0]) == 3.0                 # non-positive ignored, < 8 -> mean
    vals = [1000.0] + [10.0] * 18 + [0.001]
    assert P.trimmed_latency_average(vals) == 10.0                                  # 10 % trimmed at both ends
    assert P.one_percent_low([], 42.0) == 42.0
    assert P.one_percent_low([60.0] * 198 + [20.0, 30.0], 0.0) == 25.0              # k = max(1, int(200 * 0.01)) = 2


class _Payload:
    def __init__(self, arr):
        self._a, self.released = arr, 0

    def buffer_view(self):
        return memoryview(self._a).cast("B")

    def release(self):
        self.released += 1


def test_rgb48le_sink_and_rawvideo_source(tmp_path):
    buf = io.BytesIO()
    sink = P.Rgb48leSink(buf, 4, 2, 59.94)
    a = np.arange(4 * 2 * 3, dtype=np.uint16).reshape(2, 4, 3) * 1000
    p = _Payload(a)
    sink(p)
    assert p.released == 1 and sink.frames == 1 and buf.getvalue() == a.astype("<u2").tobytes()
    bad = _Payload(np.zeros((2, 2, 3), np.uint16))
    with pytest.raises(ValueError):
        sink(bad)
    assert bad.released == 1                                   # the ring slot is returned even on error
    assert "--demuxer-rawvideo-mp-format=rgb48le" in sink.mpv_args() and "--demuxer-rawvideo-w=4" in sink.mpv_args()
    assert sink.mpv_args()[-1] == "--vf=format=colorlevels=full:primaries=bt.2020:gamma=pq"
    ff = sink.ffmpeg_input_args()
    assert ff[ff.index("-pix_fmt") + 1] == "rgb48le" and ff[ff.index("-color_trc") + 1] == "smpte2084" and ff[-2:] == ["-i", "-"]
    # bgr24 rawvideo source
    frames = np.random.default_rng(0).integers(0, 256, (3, 6, 8, 3), dtype=np.uint8)
    path = tmp_path / "clip.bgr24"
    frames.tofile(path)
    src = P.RawVideoSource(str(path), 8, 6, 24.0)
    assert src.frame_count == 3
    for i in range(3):
        ok, f = src.read()
        assert ok and np.array_equal(f, frames[i])
    assert src.read() == (False, None)
    with pytest.raises(ValueError):
        P.RawVideoSource(str(path), 7, 6, 24.0)


def test_objective_metrics_hook():
    """With a ground-truth source attached every Nth processed frame is scored and the metrics dict carries it."""
    clk = FakeClock()
    w = FakeWorker(clk, 0.002)

    class Proc:
        calls = 0

        def objective_metrics(self, pred, gt):
            Proc.calls += 1
            return {"psnr_db": 40.0 + Proc.calls, "sssim": 0.99, "delta_e_itp": 1.5}

    w._processor = Proc()
    w._process_frame = lambda **kw: (clk.__setattr__("t", clk.t + 0.002), (None, None, "pred", False, 1.8))[1]

    class Gt:
        def read(self):
            return True, "gt"

    src = P.SyntheticSource(96, 64, fps=60.0, n_frames=21, pool=1, kind="noise")
    got = []
    r = P.RealtimePlayback(w, src, clock=clk, sleep=clk.sleep_until, gt_source=Gt(), objective_every=10, metrics_cb=got.append).run()
    assert Proc.calls == 3 and r["last_metrics"]["objective_enabled"] is True          # frames 0, 10, 20
    assert r["last_metrics"]["psnr_db"] == 43.0 and r["last_metrics"]["delta_e_itp"] == 1.5


def test_pinned_prefetch_source_preserves_order_and_marks_frames():
    """PinnedPrefetch: frames come back in order, as page-locked arrays that carry their tensor, and EOF is sticky."""
    import torch
    from hdrtv_mi355x import playback as P
    src = P.SyntheticSource(96, 64, fps=30.0, n_frames=7, pool=4, kind="noise")
    ref = P.SyntheticSource(96, 64, fps=30.0, n_frames=7, pool=4, kind="noise")
    pf = P.PinnedPrefetch(src)
    assert (pf.width, pf.height, pf.frame_count) == (96, 64, 7)
    got = []
    while True:
        ok, f = pf.read()
        if not ok:
            break
        if torch.cuda.is_available():               # page-locked staging needs a HIP runtime; without one frames pass through
            assert isinstance(f, P.PinnedFrame) and isinstance(f.pinned_tensor, torch.Tensor) and f.pinned_tensor.shape == (64, 96, 3)
            assert f.ctypes.data == f.pinned_tensor.data_ptr()
        got.append(np.array(f))
    assert pf.read() == (False, None)
    pf.release()
    assert len(got) == 7
    for g in got:
        ok, r = ref.read()
        assert ok and np.array_equal(g, r)


def test_pinned_prefetch_surfaces_source_errors():
    """An exception in the wrapped source is not swallowed by the reader thread: read() raises."""
    import pytest
    from hdrtv_mi355x import playback as P

    class Bad:
        width, height, fps, frame_count = 8, 8, 30.0, 3

        def __init__(self):
            self.n = 0

        def read(self):
            self.n += 1
            if self.n == 2:
                raise OSError("decoder died")
            return True, np.zeros((8, 8, 3), np.uint8)

        def release(self):
            pass

    pf = P.PinnedPrefetch(Bad())
    ok, f = pf.read()
    assert ok and f.shape == (8, 8, 3)
    with pytest.raises(RuntimeError, match="prefetch thread"):
        pf.read()
    assert pf.read() == (False, None)
    pf.release()
