#!/usr/bin/env python3
"""Golden vectors of the reference's TRUE fake-quant INT8 execution (predequantize off: W8Conv2d / W8A8Conv2d /
W8Linear / W8A8Linear forward, src/models/hdrtvnet_torch.py:233-410), produced by RUNNING THE REFERENCE on CPU.
This is the arithmetic BASELINE.json configs[4] asks an int8-MFMA path to reproduce (tolerance: u8 MAE <= 5, the
reference's own bound); on AMD the reference itself pre-dequantises instead (tests/golden/gen_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_w8a8.py

Data only is committed: inputs, outputs and a few activation taps."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up the reference import path)

from hdrtv_mi355x import weights as W  # noqa: E402


def main():
    f = W.synthetic_frame(64, 96, seed=6, kind="gradient")
    for tag, prec in (("full_qat", "int8-full"), ("mixed_qat", "int8-mixed")):
        ipath = os.path.join(G.REF, f"src/models/weights/original/pytorch_int8/hr/HR_original_int8_{tag}.pt")
        pi = G.HDRTVNetTorch(ipath, device="cpu", precision=prec, compile_model=False, use_hg=False, warmup_passes=0,
                             predequantize="off")
        assert pi._is_w8_model, "expected the quantised layers to stay in place"
        r = G.run_case(pi, f, ["AGCM", "LE.cond_first", "LE.HR_conv1", "LE.recon_trunk3", "LE.HR_conv2"])
        # keep the fixture small: wide full-resolution taps keep every Nth channel (0, N, 2N, ...)
        for k, step in (("tap:LE.cond_first", 8), ("tap:LE.HR_conv1", 4), ("tap:LE.HR_conv2", 4)):
            r[k] = r[k][::step].copy()
        kinds = {}
        for n, m in pi.model.named_modules():
            t = type(m).__name__
            if t.startswith("W8"):
                kinds[n] = t + (":asym" if getattr(m, "is_asymmetric", False) else "")
        r["layer_kinds"] = np.array(sorted(f"{k}={v}" for k, v in kinds.items()))
        np.savez_compressed(os.path.join(HERE, f"int8_{tag}_w8a8_64x96_gradient_s6.npz"), **r)
        print(tag, {k: sum(1 for v in kinds.values() if v == k) for k in sorted(set(kinds.values()))})


if __name__ == "__main__":
    main()
