#!/usr/bin/env python3
"""Golden vectors of the reference's two condition-map shortcuts, produced by RUNNING THE REFERENCE on CPU:
``HDRTVNetTorch(fast_condition_resize=True)`` (bilinear 0.25x, hdrtvnet_torch.py:2269-2276) and ``HDRTVNET_ZERO_COND=1``
(zero condition map, 2265-2267).  Data only is committed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_cond_modes.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

from hdrtv_mi355x import weights as W  # noqa: E402


def main():
    f = W.synthetic_frame(61, 103, seed=7, kind="gradient")
    res = {"frame": f}
    p = G.HDRTVNetTorch(G.HR_PT, device="cpu", precision="fp32", compile_model=False, use_hg=False, warmup_passes=0,
                        fast_condition_resize=True)
    r = G.run_case(p, f)
    res["cond_bilinear"], res["out_bilinear"], res["agcm_bilinear"] = r["cond"], r["out"], r["agcm_out"]
    os.environ["HDRTVNET_ZERO_COND"] = "1"
    try:
        p = G.HDRTVNetTorch(G.HR_PT, device="cpu", precision="fp32", compile_model=False, use_hg=False, warmup_passes=0)
        r = G.run_case(p, f)
    finally:
        del os.environ["HDRTVNET_ZERO_COND"]
    assert not r["cond"].any()
    res["out_zero"], res["agcm_zero"] = r["out"], r["agcm_out"]
    np.savez_compressed(os.path.join(HERE, "cond_modes_61x103_gradient_s7.npz"), **res)
    print({k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
