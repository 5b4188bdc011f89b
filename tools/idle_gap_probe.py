#!/usr/bin/env python3
"""Probe: is the frame rate power-limited?  An idle gap (a one-thread spin kernel, torch.cuda._sleep) is put in front of every
frame; if the chip is under its power cap with the clock lowered, the convolutions after an idle gap run at a higher clock and the
frame period grows by less than the gap.  usage: python tools/idle_gap_probe.py [frames]"""
import ctypes as C
import contextlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
import torch

from hdrtv_mi355x import lib as L
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X

H, Wd = 2160, 3840
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), device="cuda:0", precision="auto", use_hg=True,
                       hg_weights="seeded:1234", warmup_passes=0)
p._ensure_buffers(H, Wd)
frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient")).to(dev) for i in range(4)]
out = torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev)
lib, ctx = p._lib, p._ctx


def s():
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def step(i):
    fr = frames[i % 4]
    p._chk(lib.hdrtv_preprocess(ctx, s(), fr.data_ptr(), H, Wd, p._gpu_input.data_ptr(), p._gpu_cond.data_ptr()), "pre")
    p._chk(lib.hdrtv_infer(ctx, s(), p._gpu_input.data_ptr(), p._gpu_cond.data_ptr(), H, Wd, p._gpu_out.data_ptr(), L.F32, p._gpu_agcm.data_ptr()), "infer")
    p._chk(lib.hdrtv_post_rgb48(ctx, s(), p._gpu_out.data_ptr(), L.F32, H, Wd, out.data_ptr()), "post")


# calibrate the spin kernel: cycles per microsecond
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (a.elapsed_time(b) * 1000.0)
print(f"spin kernel: {cyc_per_us:.1f} cycles/us")
for gap_us in (0, 500, 1000, 2000, 4000, 0):
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        if gap_us:
            torch.cuda._sleep(int(gap_us * cyc_per_us))
        step(i)
        if i % 8 == 7:
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / N * 1e3
    print(f"gap {gap_us:5d} us: period {per:7.3f} ms  = compute {per - gap_us / 1e3:7.3f} ms + gap")
