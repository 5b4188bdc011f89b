"""The constructor surface of HDRTVNetTorch beyond the defaults: condition-map shortcuts, hipGraph
replay behind use_cuda_graphs, HG_Composite checkpoints, and the worker's failure paths (ring exhaustion fallback,
feeder errors, hot-swap with a sink attached)."""
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUT_MAX, OUT_MEAN = 3.7e-3, 2.5e-4      # test_gpu_parity.py's bars (1.5 x measured)


def _hr(golden_dir, **kw):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    return HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=kw.pop("use_hg", False), warmup_passes=0, **kw)


def test_cond_modes_vs_reference_runs(golden_dir, monkeypatch):
    d = np.load(os.path.join(golden_dir, "cond_modes_61x103_gradient_s7.npz"))
    p = _hr(golden_dir, fast_condition_resize=True)
    try:
        t, c = p.preprocess(d["frame"])
        dc = np.abs(c.float().cpu().numpy()[0] - d["cond_bilinear"]).max()
        out, agcm = p.infer((t, c))
        do = np.abs(out.float().cpu().numpy()[0] - d["out_bilinear"])
        print(f"  bilinear cond: max_abs={dc:.3e}; out max_abs={do.max():.3e} mean_abs={do.mean():.3e}")
        assert dc <= 1.2e-3 and do.max() <= OUT_MAX and do.mean() <= OUT_MEAN
    finally:
        p.close()
    monkeypatch.setenv("HDRTVNET_ZERO_COND", "1")
    p = _hr(golden_dir)
    try:
        t, c = p.preprocess(d["frame"])
        assert not c.any()
        out, agcm = p.infer((t, c))
        do = np.abs(out.float().cpu().numpy()[0] - d["out_zero"])
        print(f"  zero cond: out max_abs={do.max():.3e} mean_abs={do.mean():.3e}")
        assert do.max() <= OUT_MAX and do.mean() <= OUT_MEAN
    finally:
        p.close()


def test_pre_fused_cond_exact_at_4k(golden_dir):
    """The fused preprocess kernel at BASELINE size: planes exact, condition map against the oracle's separable
    antialiased bicubic fed the same fp16-rounded tensor."""
    from hdrtv_mi355x import weights as W
    from oracle import hdrtvnet_oracle as O
    p = _hr(golden_dir)
    try:
        for hw in ((2160, 3840), (1081, 1923)):
            f = W.synthetic_frame(hw[0], hw[1], seed=5, kind="noise")
            t, c = p.preprocess(f)
            tn = t.float().cpu().numpy()[0]
            assert np.array_equal(tn, ((f[:, :, ::-1].astype(np.float32) * np.float32(1 / 255.0)).astype(np.float16)
                                       .astype(np.float32)).transpose(2, 0, 1))
            ref = O.bicubic_aa_quarter(tn).astype(np.float16).astype(np.float32)
            dd = np.abs(c.float().cpu().numpy()[0] - ref)
            print(f"  cond {hw}: max_abs={dd.max():.3e} differing={int((dd != 0).sum())} of {dd.size}")
            assert dd.max() <= 1e-3
    finally:
        p.close()


def test_hip_graph_replay_is_bit_identical(golden_dir):
    import torch
    from hdrtv_mi355x import weights as W
    f = W.synthetic_frame(272, 480, seed=3, kind="gradient")
    outs, p50 = [], []
    for graphs in (False, True):
        p = _hr(golden_dir, use_hg=True, hg_weights="seeded:1234", use_cuda_graphs=graphs)
        try:
            t, c = p.preprocess(f)
            for _ in range(3):
                out, agcm = p.infer((t, c))
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                out, agcm = p.infer((t, c))
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            p50.append(sorted(ts)[10])
            assert p._use_cuda_graphs is graphs
            outs.append((out.clone(), agcm.clone()))
        finally:
            p.close()
    print(f"  infer p50 at 272x480: eager {p50[0]:.3f} ms, hipGraph replay {p50[1]:.3f} ms")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_hg_composite_checkpoint_layout(golden_dir):
    """An HG_Composite state (base.* + hg.*), the layout of pytorch_int8/hg/HR_HG_*.pt, fp16 and INT8."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    d = np.load(os.path.join(golden_dir, "hg_96x128_gradient_s3.npz"))
    hr = W.load_pack(os.path.join(golden_dir, "hr_weights.hdrw"))
    comp = {"base." + k: v for k, v in hr.items()}
    comp.update({"hg." + k: v for k, v in W.seeded_hg_state(1234).items()})
    p = HDRTVNetMI355X(comp, use_hg=True, warmup_passes=0)
    try:
        assert p._use_hg
        out, _ = p.infer(p.preprocess(d["frame"]))
        dd = np.abs(out.cpu().numpy()[0] - d["out"])
        assert np.percentile(dd, 99.9) <= 8e-3
    finally:
        p.close()
    hr8 = W.load_pack(os.path.join(golden_dir, "hr_int8_mixed_qat.hdrw"))
    comp = {"base." + k: v for k, v in hr8.items()}
    comp.update({"hg." + k: v for k, v in W.seeded_hg_w8a8_state(1234).items()})
    for pd in ("auto", "off"):
        p = HDRTVNetMI355X(comp, precision="int8-mixed", predequantize=pd, use_hg=True, warmup_passes=0)
        try:
            assert p._use_hg and p._hg_int8 and p._is_w8_model is (pd == "off")
            out, _ = p.infer(p.preprocess(d["frame"]))
            assert np.isfinite(out.cpu().numpy()).all()
        finally:
            p.close()
    with pytest.raises(FileNotFoundError):        # INT8 no-HG checkpoint + HG requested + nothing to find (1928-1937)
        HDRTVNetMI355X(os.path.join(golden_dir, "hr_int8_mixed_qat.hdrw"), precision="int8-mixed", use_hg=True, warmup_passes=0)


def _worker(golden_dir, tmp_path, **kw):
    from hdrtv_mi355x.worker import HeadlessPipelineWorker
    wdir = tmp_path / "weights" / "original"
    (wdir / "pytorch_int8" / "hr").mkdir(parents=True)
    os.symlink(os.path.join(golden_dir, "hr_weights.hdrw"), wdir / "HR.hdrw")
    os.symlink(os.path.join(golden_dir, "hr_int8_mixed_qat.hdrw"), wdir / "pytorch_int8" / "hr" / "HR_original_int8_mixed_qat.hdrw")
    return HeadlessPipelineWorker(str(tmp_path / "weights"), use_hg=False, proc_w=96, proc_h=64, **kw)


def test_hot_swap_keeps_the_sink_fed(golden_dir, tmp_path):
    """request_precision with a sink attached: the feeder is restarted on the new processor and frames keep arriving."""
    from hdrtv_mi355x import playback as P
    w = _worker(golden_dir, tmp_path, buffer_frames=2)
    assert w._load_model("FP16")
    got = []
    lock = threading.Lock()

    def sink(payload):
        with lock:
            got.append(payload.numpy().copy())
        payload.release()

    w._start_hdr_feeder(sink)
    src = P.SyntheticSource(96, 64, fps=240.0, n_frames=24, pool=2, kind="gradient")
    pb = P.RealtimePlayback(w, src, sink=True, realtime=False)

    orig = w._process_frame          # flips the preset after 8 frames, as the GUI's precision combo does mid-playback

    def counting(**kw):
        r = orig(**kw)
        if kw["frame_idx"] == 8:
            pb.request_precision("INT8 Mixed (QAT)")
        return r
    w._process_frame = counting
    res = pb.run()
    deadline = time.perf_counter() + 10.0
    while len(got) < res["frames_processed"] and time.perf_counter() < deadline:
        time.sleep(0.01)
    assert w._precision_key == "INT8 Mixed (QAT)" and w._hdr_queue is not None
    assert res["frames_processed"] == 24 and len(got) >= 22, (len(got), w.status_messages[-3:])
    w.close()


def test_ring_exhaustion_falls_back_and_feeder_errors_surface(golden_dir, tmp_path):
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.worker import HostFrame, PinnedFrame
    from oracle import hdrtvnet_oracle as O
    w = _worker(golden_dir, tmp_path)
    assert w._load_model("FP16")
    f = W.synthetic_frame(64, 96, seed=77, kind="noise")
    _, _, prepared, _, _ = w._process_frame(frame=f, frame_idx=0, mpv_w=None)
    want = O.post_rgb48(prepared.float().cpu().numpy()[0])
    held = [w._tensor_to_rgb48_bytes(prepared) for _ in range(3)]          # the whole ring, never released
    assert all(isinstance(h, PinnedFrame) for h in held)
    t0 = time.perf_counter()
    fb = w._tensor_to_rgb48_bytes(prepared)                                 # 250 ms acquire timeout, then the blocking buffer
    assert isinstance(fb, HostFrame) and time.perf_counter() - t0 >= 0.2 and w.ring_fallbacks == 1
    assert np.array_equal(fb.numpy(), want)
    for h in held:
        assert np.array_equal(h.numpy(), want)
        h.release()
    assert isinstance(w._tensor_to_rgb48_bytes(prepared), PinnedFrame)
    # a sink that raises ends the feeder thread; the next _process_frame reports it instead of dropping frames silently
    w._ring_shape = None

    def bad_sink(payload):
        payload.release()
        raise BrokenPipeError("sink went away")

    w._start_hdr_feeder(bad_sink)
    w._process_frame(frame=f, frame_idx=1, mpv_w=True)
    deadline = time.perf_counter() + 5.0
    while w._hdr_error is None and time.perf_counter() < deadline:
        time.sleep(0.01)
    with pytest.raises(RuntimeError, match="HDR feeder failed"):
        w._process_frame(frame=f, frame_idx=2, mpv_w=True)
    w.close()


@pytest.mark.parametrize("worker_opts,slots", [({}, 2), ({"lanes": 2, "frames_in_flight": 3}, 4)])
def test_dispatcher_on_the_device(golden_dir, worker_opts, slots):
    """hdrtv_mi355x/dispatch.py with the product's worker body: two worker processes (both on this box's one GPU), frames
    round-robin, two frames in flight per worker (upload / compute / download on three streams, DMA straight from and into
    the page-locked shared slots), RGB48 frames back in order and bit-identical to an in-process processor's -- at 1920x1080,
    where the persistent kernels walk many tiles and the copies are long enough to overlap compute.  Second case: the worker
    with two compute lanes and three frames in flight (frame n on lane n mod 2, buffers n mod 3)."""
    import ctypes as C
    import torch
    from hdrtv_mi355x import lib as L, weights as W
    from hdrtv_mi355x.dispatch import FrameDispatcher
    h, w, n = 1080, 1920, 9
    frames = [W.synthetic_frame(h, w, seed=200 + i, kind="gradient" if i % 2 else "noise") for i in range(n)]
    p = _hr(golden_dir, use_hg=True, hg_weights="seeded:1234")
    want = []
    u16 = torch.empty((h, w, 3), dtype=torch.uint16, device="cuda")
    for f in frames:
        out, _ = p.infer(p.preprocess(f))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert p._lib.hdrtv_post_rgb48(p._ctx, st, out.data_ptr(), L.F32, h, w, u16.data_ptr()) == 0
        want.append(u16.cpu().numpy().copy())
    p.close()
    got = {}
    args = dict({"model_path": os.path.join(golden_dir, "hr_weights.hdrw"), "use_hg": True, "hg_weights": "seeded:1234"}, **worker_opts)
    with FrameDispatcher(2, h, w, lambda i, v: got.__setitem__(i, v.copy()), init_args=args, devices=[0, 0], slots=slots) as d:
        for f in frames:
            d.submit(f)
        d.flush(timeout=120)
    assert d.exit_codes == [0, 0]
    assert sorted(got) == list(range(n))
    for i in range(n):
        assert np.array_equal(got[i], want[i]), i


def test_pq_table_survives_letterbox_growth(golden_dir):
    """One context: post_pq_rgb48 (builds the PQ boundary table), two letterbox calls of growing geometry (the second
    re-allocates the letterbox tables), post_pq_rgb48 again -- bit for bit the oracle both times; then close().  (Round 2's
    letterbox_setup freed the PQ table in its grow branch and left the pointer set: ADVICE r02.)"""
    import ctypes as C
    import torch
    from hdrtv_mi355x import lib as L
    from oracle import hdrtvnet_oracle as O
    from oracle import letterbox_oracle as LB
    p = _hr(golden_dir)
    try:
        h, w = 72, 128
        x = np.random.default_rng(9).uniform(-0.05, 1.05, (3, h, w)).astype(np.float32)
        xin = torch.from_numpy(x).cuda()
        u16 = torch.empty((h, w, 3), dtype=torch.uint16, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        want = O.post_pq_rgb48(x, 1000.0)

        def pq():
            u16.zero_()
            p._chk(p._lib.hdrtv_post_pq_rgb48(p._ctx, st, xin.data_ptr(), L.F32, h, w, C.c_float(1000.0), u16.data_ptr()),
                   "hdrtv_post_pq_rgb48")
            torch.cuda.synchronize()
            return u16.cpu().numpy()

        assert np.array_equal(pq(), want)
        for (sh, sw), (dh, dw) in (((57, 76), (96, 96)), ((270, 480), (1080, 1920))):
            f = np.random.default_rng(sh).integers(0, 256, (sh, sw, 3), dtype=np.uint8)
            src = torch.from_numpy(f).cuda()
            dst = torch.empty((dh, dw, 3), dtype=torch.uint8, device="cuda")
            p._chk(p._lib.hdrtv_letterbox_u8(p._ctx, st, src.data_ptr(), sh, sw, dst.data_ptr(), dh, dw), "hdrtv_letterbox_u8")
            torch.cuda.synchronize()
            assert np.array_equal(dst.cpu().numpy(), LB.letterbox_bgr(f, dw, dh))
            assert np.array_equal(pq(), want)
    finally:
        p.close()


def test_hip_graph_follows_mask_r_and_profiling(golden_dir):
    """set_hg_mask_r() after a capture must take effect on the next infer (the captured graph baked the old threshold in),
    and profile_enable() must time real launches, not a replay that records nothing."""
    import torch
    from hdrtv_mi355x import weights as W
    f = W.synthetic_frame(96, 128, seed=3, kind="gradient")
    res = {}
    for graphs in (False, True):
        p = _hr(golden_dir, use_hg=True, hg_weights="seeded:1234", use_cuda_graphs=graphs)
        try:
            t, c = p.preprocess(f)
            a = p.infer((t, c))[0].clone()
            a2 = p.infer((t, c))[0].clone()                   # replay
            p.set_hg_mask_r(0.3)
            b = p.infer((t, c))[0].clone()
            p.profile_enable(True)
            p.infer((t, c))
            prof = p.profile_read()
            p.profile_enable(False)
            b2 = p.infer((t, c))[0].clone()
            assert torch.equal(a, a2) and torch.equal(b, b2)
            assert len(prof) == p.infer_stats()[0] and all(ms > 0 for _, _, ms, _, _ in prof)
            res[graphs] = (a, b)
        finally:
            p.close()
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    assert not torch.equal(res[True][0], res[True][1])        # the dense mask blends the HG residual into far more pixels


@pytest.mark.gpu
def test_bench_prints_exactly_one_line_on_stdout():
    """The driver parses bench.py's stdout as ONE JSON line: nothing else (processor banners of the dispatcher's worker processes,
    warnings, progress) may land there.  Small frame, dispatcher leg included, everything else at its default."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--height", "270", "--width", "480", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-int8-extra"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["dispatcher_host_fed"]["value"] > 0


def test_ring_slot_state_machine(golden_dir):
    """hdrtv_ring_*: a slot cycles free -> acquired -> committed -> free; every call in the wrong state is HDRTV_ESTATE instead of
    stale pixels (waiting on an acquired-but-uncommitted slot used to return at once on a never-recorded event)."""
    import ctypes as C
    import torch
    from hdrtv_mi355x import lib as L
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    lib, ctx = p._lib, p._ctx
    try:
        assert lib.hdrtv_ring_wait(ctx, 0) == L.EINVAL                      # no ring yet
        assert lib.hdrtv_ring_create(ctx, 2, 64, 96) == 0
        hp, dp = C.c_void_p(), C.c_void_p()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.hdrtv_ring_commit(ctx, 0, st) == L.ESTATE                # free slot: nothing to commit
        assert lib.hdrtv_ring_release(ctx, 0) == L.ESTATE                   # double release
        s0 = lib.hdrtv_ring_acquire(ctx, 10, C.byref(hp), C.byref(dp))
        assert s0 == 0 and hp.value and dp.value
        assert lib.hdrtv_ring_wait(ctx, s0) == L.ESTATE                     # acquired, not committed: no copy in flight
        assert b"not committed" in lib.hdrtv_last_error(ctx)
        out = torch.rand((3, 64, 96), dtype=torch.float16, device="cuda")
        assert lib.hdrtv_post_rgb48(ctx, st, out.data_ptr(), L.F16, 64, 96, dp) == 0
        assert lib.hdrtv_ring_commit(ctx, s0, st) == 0
        assert lib.hdrtv_ring_commit(ctx, s0, st) == L.ESTATE               # committed twice
        assert lib.hdrtv_ring_wait(ctx, s0) == 0
        host = np.ctypeslib.as_array(C.cast(hp.value, C.POINTER(C.c_uint16)), shape=(64, 96, 3)).copy()
        from oracle import hdrtvnet_oracle as O
        assert np.array_equal(host, O.post_rgb48(out.float().cpu().numpy()))
        s1 = lib.hdrtv_ring_acquire(ctx, 10, C.byref(hp), C.byref(dp))
        assert s1 == 1
        assert lib.hdrtv_ring_acquire(ctx, 10, C.byref(hp), C.byref(dp)) == L.ESTATE      # exhausted
        assert lib.hdrtv_ring_release(ctx, s0) == 0 and lib.hdrtv_ring_release(ctx, s1) == 0     # an acquired slot may be given back
        assert lib.hdrtv_ring_wait(ctx, 5) == L.EINVAL
        assert lib.hdrtv_ring_destroy(ctx) == 0
        assert lib.hdrtv_ring_wait(ctx, 0) == L.EINVAL and lib.hdrtv_ring_commit(ctx, 0, st) == L.EINVAL
    finally:
        p.close()
