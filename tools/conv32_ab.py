#!/usr/bin/env python3
"""Diagnostic: LE taps with conv32s.hip (default) against conv32p.hip (HDRTV_VARIANTS=conv32_old=1,le_rows=0), first differing tap and where.
usage: python tools/conv32_ab.py [H W] [fp16|int8-full|int8-mixed]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
from hdrtv_mi355x import weights as W  # noqa: E402
from hdrtv_mi355x.processor import HDRTVNetMI355X  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (270, 486)
variant = sys.argv[3] if len(sys.argv) > 3 else "fp16"
g = os.path.join(REPO, "tests", "golden")
if variant == "fp16":
    p = HDRTVNetMI355X(os.path.join(g, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
else:
    p = HDRTVNetMI355X(os.path.join(g, f"hr_int8_{variant.split('-')[1]}_qat.hdrw"), precision=variant, predequantize="off",
                       use_hg=False, warmup_passes=0)
taps = ("le.conv_first", "le.fea0", "le.fea1a", "le.fea1", "le.fea2", "le.fea3", "le.up1", "le.up2", "le.up3", "le.out")
f = W.synthetic_frame(h, w, seed=41, kind="gradient")
res = []
for old in ("1", None):
    if old:
        os.environ["HDRTV_VARIANTS"] = "le_rows=0,conv32_old=" + old
    else:
        os.environ["HDRTV_VARIANTS"] = "le_rows=0"
    out, _ = p.infer(p.preprocess(f))
    r = {"out": out.float().cpu().numpy()}
    for t in taps:
        try:
            r[t] = p.tap(t).float().cpu().numpy()
        except Exception as e:  # noqa: BLE001
            r[t] = None
    res.append(r)
for k in ("le.conv_first", "le.fea0", "le.fea1a", "le.fea1", "le.fea2", "le.fea3", "le.up1", "le.up2", "le.up3", "le.out", "out"):
    a, b = res[0][k], res[1][k]
    if a is None:
        print(k, "n/a")
        continue
    d = a != b
    n = int(d.sum())
    msg = f"{k:14s} shape {a.shape} differing {n}"
    if n:
        idx = np.argwhere(d)
        msg += f"  first {idx[0].tolist()} last {idx[-1].tolist()} max|d| {np.abs(a - b).max():.4g} nan_new {int(np.isnan(b).sum())}"
        ax = idx[:, -2] if a.ndim == 3 and a.shape[-1] <= 64 else idx[:, -1]
        ys = idx[:, 0] if a.ndim == 3 and a.shape[-1] <= 64 else idx[:, -2]
        msg += f"  rows%16 {sorted(set((ys % 16).tolist()))[:16]} cols%16 {sorted(set((ax % 16).tolist()))[:16]}"
    print(msg)
p.close()
