#!/usr/bin/env python3
"""A/B: hdrtv_infer launched eagerly (65 launches per frame) against a hipGraph replay of the same launches.
usage: python tools/graph_ab.py [frames]"""
import contextlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
import torch

from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X

H, Wd = 2160, 3840
frames = [W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient") for i in range(2)]
for graphs in (False, True, False, True):
    with contextlib.redirect_stdout(sys.stderr):
        p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), device="cuda:0", precision="auto", use_hg=True,
                           hg_weights="seeded:1234", warmup_passes=0, use_cuda_graphs=graphs)
    pre = [tuple(t.clone() for t in p.preprocess(f)) for f in frames]
    for i in range(5):
        p.infer(pre[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        p.infer(pre[i % 2])
    torch.cuda.synchronize()
    print(f"graphs={graphs}: infer {1e3 * (time.perf_counter() - t0) / N:.3f} ms/frame")
    p.close()
