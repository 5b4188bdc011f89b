"""Letterbox: the oracle's restatement of cv2.resize INTER_AREA / INTER_CUBIC + black padding
(parity with cv2 itself is UNPINNED: OpenCV is not in the reference tree) against known answers, and the device
entry point hdrtv_letterbox_u8 against that oracle, bit for bit.  Reference: src/gui_scaling.py:228-244."""
import numpy as np
import pytest

from oracle import letterbox_oracle as L


def test_geometry_matches_the_reference_formula():
    # (w, h, out_w, out_h) -> new_w, new_h, x0, y0, interp ; values worked by hand from gui_scaling.py:234-243
    assert L.geometry(3840, 2160, 1920, 1080) == (1920, 1080, 0, 0, L.AREA)
    assert L.geometry(1280, 720, 1920, 1080) == (1920, 1080, 0, 0, L.CUBIC)
    assert L.geometry(720, 576, 1920, 1080) == (1350, 1080, 285, 0, L.CUBIC)          # 4:3 pillar-boxed
    assert L.geometry(1920, 800, 1920, 1080) == (1920, 800, 0, 140, L.CUBIC)          # scope: scale 1.0 -> bars only
    assert L.geometry(4096, 1716, 1920, 1080) == (1920, 804, 0, 138, L.AREA)          # round(1716 * 0.46875) = 804
    assert L.geometry(5, 3, 2, 2) == (2, 1, 0, 0, L.AREA)                             # round(1.2) = 1, never below 1


def test_area_known_answers():
    f = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3) * 3
    half = L.resize_area(f, 3, 2)                                  # 2x2 boxes: (a + b + c + d + 2) >> 2
    blk = f.reshape(2, 2, 3, 2, 3).astype(int).sum(axis=(1, 3))
    assert np.array_equal(half, (blk + 2) >> 2)
    g = np.zeros((3, 3, 3), np.uint8); g[0, 0] = 255; g[1, 1] = 1
    assert np.array_equal(L.resize_area(g, 1, 1)[0, 0], [28, 28, 28])           # 256/9 = 28.44 -> 28
    t = np.zeros((2, 3, 3), np.uint8); t[:, 0] = 90; t[:, 1] = 30; t[:, 2] = 60
    # 3 -> 2 columns: weights (1, .5)/1.5 and (.5, 1)/1.5: (90 + 15)/1.5 = 70, (15 + 60)/1.5 = 50
    assert np.array_equal(L.resize_area(t, 2, 2)[:, :, 0], [[70, 50], [70, 50]])
    k = np.full((9, 14, 3), 201, np.uint8)
    assert np.unique(L.resize_area(k, 5, 4)).tolist() == [201]                    # weights sum to one
    tab = L.area_tab(7, 3)
    for d in range(3):
        assert abs(sum(float(a) for dd, _, a in tab if dd == d) - 1.0) < 1e-6


def test_cubic_known_answers():
    k = np.full((5, 7, 3), 99, np.uint8)
    assert np.unique(L.resize_cubic(k, 20, 11)).tolist() == [99]                   # coefficients sum to 2048
    xo, xc = L.cubic_tab(4, 8)
    assert xc.sum(axis=1).tolist() == [2048] * 8
    # destination 0 of a 2x enlargement samples at fx = -0.25 -> taps start at -2, fraction 0.75; by hand with A = -0.75:
    # c = (-0.03515625, 0.26171875, 0.87890625, -0.10546875) * 2048
    assert xo[0] == -2 and xc[0].tolist() == [-72, 536, 1800, -216]
    ramp = np.repeat(np.arange(0, 160, 10, dtype=np.uint8)[None, :, None], 4, 0).repeat(3, 2)
    up = L.resize_cubic(ramp, 32, 8)[0, :, 0].astype(int)
    assert np.all(np.diff(up[2:-2]) >= 0) and abs(int(up[16]) - 78) <= 1          # a ramp stays a ramp away from the edges


def test_letterbox_canvas():
    rng = np.random.default_rng(3)
    f = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    assert L.letterbox_bgr(f, 40, 30) is f                                          # same size: the frame itself
    out = L.letterbox_bgr(f, 64, 64)                                                # scale 1.6 -> 64x48, bars of 8 rows
    assert out.shape == (64, 64, 3) and not out[:8].any() and not out[56:].any() and out[8:56].any()
    out2 = L.letterbox_bgr(f, 20, 30)                                               # scale 0.5 -> 20x15 at y0 = 7
    assert np.array_equal(out2[7:22], L.resize_area(f, 20, 15)) and not out2[:7].any() and not out2[22:].any()
    out3 = L.letterbox_bgr(f, 50, 30)                                               # scale 1.0 horizontally padded only
    assert np.array_equal(out3[:, 5:45], f) and not out3[:, :5].any() and not out3[:, 45:].any()


@pytest.mark.gpu
@pytest.mark.parametrize("src_hw,dst_hw", [((108, 192), (54, 96)),      # 2x2 boxes
                                           ((216, 384), (72, 128)),     # 3x3 boxes
                                           ((100, 180), (64, 96)),      # fractional area, letterboxed
                                           ((72, 128), (180, 320)),     # 2.5x cubic
                                           ((57, 76), (96, 96)),        # 4:3 into a square: cubic + pillar/letter bars
                                           ((64, 96), (64, 120)),       # bars only
                                           ((270, 480), (1080, 1920))]) # 4x cubic to full HD
def test_device_letterbox_equals_oracle(src_hw, dst_hw):
    import ctypes as C
    import torch
    from hdrtv_mi355x import lib as Lb
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    import os
    golden = os.path.join(os.path.dirname(__file__), "golden")
    p = HDRTVNetMI355X(os.path.join(golden, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    rng = np.random.default_rng(src_hw[0] * 1000 + dst_hw[1])
    f = rng.integers(0, 256, (*src_hw, 3), dtype=np.uint8)
    want = L.letterbox_bgr(f, dst_hw[1], dst_hw[0])
    src = torch.from_numpy(f).cuda()
    dst = torch.full((*dst_hw, 3), 7, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p._chk(p._lib.hdrtv_letterbox_u8(p._ctx, st, src.data_ptr(), src_hw[0], src_hw[1], dst.data_ptr(), dst_hw[0], dst_hw[1]),
           "hdrtv_letterbox_u8")
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), want)
    # and through the processor: preprocess_letterboxed == preprocess(letterboxed frame)
    t1, c1 = p.preprocess_letterboxed(f, dst_hw[1], dst_hw[0])
    t1, c1 = t1.clone(), c1.clone()
    t2, c2 = p.preprocess(np.ascontiguousarray(want))
    assert torch.equal(t1, t2) and torch.equal(c1, c2)
    p.close()
