"""Objective metrics: the oracle's restatement of the reference's PSNR / SSIM / dE-ITP formulas
(gui_objective_metrics.py:438-528; parity unpinned: the module needs cv2) against known answers, and the device
entry point hdrtv_metrics against that oracle."""
import numpy as np
import pytest

from oracle import metrics_oracle as M


def test_known_answers():
    rng = np.random.default_rng(0)
    a = rng.random((3, 24, 40), dtype=np.float32)
    m = M.metrics(a, a)
    assert m["psnr_db"] == 99.0 and abs(m["sssim"] - 1.0) < 1e-6 and abs(m["delta_e_itp"] - 720e-6) < 1e-9
    b = np.clip(a + np.float32(0.1), 0, None)
    assert abs(M.psnr(a, b) - 20.0) < 1e-4                                   # mse = 0.01 -> 20 dB
    # L = 100 / 1000 / 10000 cd/m2 -> PQ 0.508078 / 0.751829 / 1.0
    assert np.allclose(M.pq_oetf_absolute(np.array([0.0, 100.0, 1000.0, 10000.0])), [7.31e-7, 0.508078, 0.751829, 1.0], atol=2e-6)
    k = M.gaussian_kernel()
    assert abs(float(k.sum()) - 1.0) < 1e-6 and abs(float(k[5]) - 0.266012) < 1e-5 and np.allclose(k, k[::-1])
    # achromatic pixels have T = P = 0 (BT.2124: the rows of the ICtCp matrix sum to zero) and I = PQ(luminance)
    i, t, p_ = M.itp(np.full((3, 2, 2), 1000.0, np.float32))
    # (float32: the +-6610..17933 products cancel to ~1e-4, not to zero)
    assert abs(float(i[0, 0]) - 0.751829) < 2e-6 and abs(float(t[0, 0])) < 5e-4 and abs(float(p_[0, 0])) < 5e-4
    # a flat image against itself shifted: SSIM of two constants = (2ab + c1) / (a^2 + b^2 + c1)
    fa, fb = np.full((3, 20, 20), 0.5, np.float32), np.full((3, 20, 20), 0.6, np.float32)
    # (float32 cancellation in blur(x*x) - mu*mu is ~1e-8 against c2 = 9e-4)
    assert abs(M.ssim(fa, fb) - (2 * 0.3 + 1e-4) / (0.25 + 0.36 + 1e-4)) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("hw,dtype", [((64, 96), "float32"), ((57, 83), "float32"), ((270, 480), "float16")])
def test_device_metrics_match_oracle(hw, dtype):
    import os
    import torch
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    golden = os.path.join(os.path.dirname(__file__), "golden")
    p = HDRTVNetMI355X(os.path.join(golden, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    rng = np.random.default_rng(hw[0])
    a = rng.random((3, *hw), dtype=np.float32)
    b = np.clip(a + rng.normal(0, 0.03, a.shape).astype(np.float32), -0.05, 1.1).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda().to(getattr(torch, dtype)), torch.from_numpy(b).cuda().to(getattr(torch, dtype))
    want = M.metrics(ta.float().cpu().numpy(), tb.float().cpu().numpy())
    got = p.objective_metrics(ta[None], tb[None])
    got2 = p.objective_metrics(ta[None], tb[None])
    assert got == got2                                                       # fixed reduction order
    print("  ", hw, dtype, got, want)
    assert abs(got["psnr_db"] - want["psnr_db"]) < 1e-3
    assert abs(got["sssim"] - want["sssim"]) < 2e-5
    assert abs(got["delta_e_itp"] - want["delta_e_itp"]) < 2e-3 * max(1.0, want["delta_e_itp"])
    same = p.objective_metrics(ta[None], ta[None])
    assert same["psnr_db"] == 99.0 and abs(same["sssim"] - 1.0) < 1e-6 and abs(same["delta_e_itp"] - 720e-6) < 1e-8
    p.close()
