"""Frames in flight (hdrtv_set_lanes / hdrtv_infer_lane / processor.enqueue_frame).

A lane is one more activation workspace, set of boundary tensors and HIP stream for the SAME network: what a frame's
RGB48 bytes are must not depend on the lane it ran on, nor on what the other lanes were doing meanwhile.  The yardstick
is the one-lane, one-stream order (the reference's frame-at-a-time loop), itself held to the oracle and the goldens by
tests/test_gpu_parity.py and tests/test_gpu_headline_parity.py."""
import ctypes as C
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    return torch


def _make(golden_dir, int8, lanes):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    if int8:
        return HDRTVNetMI355X(os.path.join(golden_dir, "hr_int8_full_qat.hdrw"), precision="int8-full", predequantize="off", use_hg=True,
                              hg_weights="seeded-w8a8:1234", warmup_passes=0, lanes=lanes)
    return HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0, lanes=lanes)


@pytest.mark.parametrize("int8,size", [(False, (2160, 3840)), (False, (540, 962)), (True, (2160, 3840))])
def test_overlapping_lanes_write_the_bytes_of_one_frame_at_a_time(torch_cuda, golden_dir, int8, size):
    from hdrtv_mi355x import weights as W
    torch = torch_cuda
    h, w = size
    p = _make(golden_dir, int8, lanes=2)
    try:
        assert p.lanes == 2 and p._lib.hdrtv_get_lanes(p._ctx) == 2
        dev = p.device
        frames = [torch.from_numpy(W.synthetic_frame(h, w, seed=70 + i, kind=("noise", "gradient", "noise")[i])).to(dev) for i in range(3)]
        # yardstick: lane 0 only, one frame at a time
        want = []
        for f in frames:
            o = torch.empty((h, w, 3), dtype=torch.uint16, device=dev)
            p.enqueue_frame(0, f.data_ptr(), h, w, o.data_ptr())
            torch.cuda.synchronize(dev)
            want.append(o)
        assert not torch.equal(want[0], want[1]) and not torch.equal(want[0], want[2])
        # 12 frames back to back, frame i on lane i mod 2, frame content rotating against the lanes: every lane sees every frame
        # while the other lane is busy with a different one
        outs = [torch.zeros((h, w, 3), dtype=torch.uint16, device=dev) for _ in range(12)]
        which = [(i + i // 4) % 3 for i in range(12)]
        for i in range(12):
            p.enqueue_frame(i % 2, frames[which[i]].data_ptr(), h, w, outs[i].data_ptr())
        torch.cuda.synchronize(dev)
        assert {(i % 2, which[i]) for i in range(12)} == {(l, f) for l in range(2) for f in range(3)}
        for i in range(12):
            assert torch.equal(outs[i], want[which[i]]), (i, i % 2, which[i], int((outs[i] != want[which[i]]).sum()))
        # the reference-shaped calls still run on lane 0 and agree with it
        out, _ = p.infer(p.preprocess(frames[1].cpu().numpy()))
        o = torch.empty((h, w, 3), dtype=torch.uint16, device=dev)
        from hdrtv_mi355x import lib as L
        p._chk(p._lib.hdrtv_post_rgb48(p._ctx, p._stream(), out.data_ptr(), L.F32, h, w, o.data_ptr()), "post")
        torch.cuda.synchronize(dev)
        assert torch.equal(o, want[1])
    finally:
        p.close()


def test_lane_count_is_part_of_the_reservation(torch_cuda, golden_dir):
    from hdrtv_mi355x import lib as L
    from hdrtv_mi355x import weights as W
    torch = torch_cuda
    h, w = 270, 480
    p = _make(golden_dir, False, lanes=1)
    try:
        lib, ctx = p._lib, p._ctx
        p._ensure_buffers(h, w)
        dev = p.device
        f = torch.from_numpy(W.synthetic_frame(h, w, seed=5, kind="gradient")).to(dev)
        o0 = torch.empty((h, w, 3), dtype=torch.uint16, device=dev)
        p.enqueue_frame(0, f.data_ptr(), h, w, o0.data_ptr())
        torch.cuda.synchronize(dev)
        tin, tcond, tout, tagcm = p._lane_bufs[0]
        args = (p._stream(), tin.data_ptr(), tcond.data_ptr(), h, w, tout.data_ptr(), L.F32, tagcm.data_ptr())
        # one lane reserved: lane 1 does not exist
        assert lib.hdrtv_infer_lane(ctx, 1, *args) == L.EINVAL
        assert b"lane 1 of 1" in lib.hdrtv_last_error(ctx)
        assert lib.hdrtv_infer_lane(ctx, -1, *args) == L.EINVAL
        for bad in (0, 5, -2):
            assert lib.hdrtv_set_lanes(ctx, bad) == L.EINVAL
        assert lib.hdrtv_set_lanes(ctx, 1) == L.OK                 # unchanged: the reservation stays
        assert lib.hdrtv_infer_lane(ctx, 0, *args) == L.OK
        torch.cuda.synchronize(dev)
        # a new count drops the reservation: infer refuses until hdrtv_reserve has run again
        assert lib.hdrtv_set_lanes(ctx, 2) == L.OK and lib.hdrtv_get_lanes(ctx) == 2
        assert lib.hdrtv_infer_lane(ctx, 0, *args) == L.ESTATE
        assert lib.hdrtv_reserve(ctx, h, w) == L.OK
        assert lib.hdrtv_preprocess(ctx, args[0], f.data_ptr(), h, w, tin.data_ptr(), tcond.data_ptr()) == L.OK
        assert lib.hdrtv_infer_lane(ctx, 1, *args) == L.OK
        o1 = torch.empty((h, w, 3), dtype=torch.uint16, device=dev)
        assert lib.hdrtv_post_rgb48(ctx, args[0], tout.data_ptr(), L.F32, h, w, o1.data_ptr()) == L.OK
        torch.cuda.synchronize(dev)
        assert torch.equal(o0, o1)
        assert lib.hdrtv_infer_lane(ctx, 2, *args) == L.EINVAL
        with pytest.raises(ValueError):
            p.enqueue_frame(1, f.data_ptr(), h, w, o1.data_ptr())      # the Python mirror still has one lane
    finally:
        p.close()
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    with pytest.raises(ValueError):
        HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0, lanes=5)


def test_lane_limits(torch_cuda, golden_dir, monkeypatch):
    """One or two lanes; the fp32 preset one.  Three kernels running at once is where wrong hg.conv2 tiles were seen with W8A8 layers
    (about one frame in 500; none with two lanes or two hardware queues: tools/dbg/lane_stress2.py), and the fp32 vector kernels
    keep the packed-f32 arithmetic that failed beside another stream's MFMA waves."""
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    monkeypatch.delenv("HDRTV_LANES_ANY", raising=False)
    with pytest.raises(ValueError):
        _make(golden_dir, False, lanes=3)
    with pytest.raises(RuntimeError, match="one frame at a time"):
        HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), precision="fp32", use_hg=False, warmup_passes=0, lanes=2)
    p = _make(golden_dir, True, lanes=2)
    try:
        assert p.lanes == 2 and p._lib.hdrtv_set_lanes(p._ctx, 3) < 0 and b"1 or 2" in p._lib.hdrtv_last_error(p._ctx)
    finally:
        p.close()


@pytest.mark.parametrize("int8", [False, True])
def test_a_lane_is_not_disturbed_by_the_frames_in_flight_beside_it(torch_cuda, golden_dir, int8):
    """Rounds of lane 0 = one fixed frame with OTHER frames enqueued on lane 1 around it: lane 0's RGB48 bytes and its
    condition map are those of the quiet run every time.  (Round 5: packed-f32 arithmetic in pre_fused -- v_pk_mul / add / fma_f32
    from the SLP vectoriser -- read a stale operand when its waves shared a SIMD with conv1x1_i8's MFMA waves of another lane's
    frame: a handful of wrong condition-map values per disturbed call, 4 .. 14 of 40 such rounds; the library is built without
    packed f32 since, tests/test_isa_contracts.py.  72 lane-0 frames here; the stress tools ran 4000 int8 and 7500 + 9000 fp16 lane-frames clean.)"""
    from hdrtv_mi355x import weights as W
    torch = torch_cuda
    h, w = 2160, 3840
    p = _make(golden_dir, int8, lanes=2)
    try:
        dev = p.device
        frames = [torch.from_numpy(W.synthetic_frame(h, w, seed=70 + i, kind=("noise", "gradient", "noise", "gradient")[i])).to(dev) for i in range(4)]
        outs = [torch.empty((h, w, 3), dtype=torch.uint16, device=dev) for _ in range(2)]
        p.enqueue_frame(0, frames[2].data_ptr(), h, w, outs[0].data_ptr())
        torch.cuda.synchronize(dev)
        ref_out, ref_cond = outs[0].clone(), p._lane_bufs[0][1].clone()
        bad = []
        for r in range(36):
            order = (0, 1) if r % 2 == 0 else (1, 0)
            for rep in range(2):
                for l in order:
                    p.enqueue_frame(l, frames[2 if l == 0 else (l + r + rep) % 4].data_ptr(), h, w, outs[l].data_ptr())
            torch.cuda.synchronize(dev)
            nc, no = int((p._lane_bufs[0][1] != ref_cond).sum()), int((outs[0] != ref_out).sum())
            if nc or no:
                bad.append((r, nc, no))
        print(f"  disturbed rounds (round, condition-map values, RGB48 values): {bad}")
        # (the packed-f32 failures were 4 .. 14 disturbed rounds of 40; with THREE lanes one round of one run of this test once showed
        # 360 differing RGB48 values, and int8 frames about one in 500 a few wrong hg.conv2 tiles: hdrtv_set_lanes stops at two)
        assert not bad, bad
    finally:
        p.close()
