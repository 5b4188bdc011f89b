#!/usr/bin/env python3
"""Developer aid (GPU box): HIP-event timing of hdrtv_preprocess / hdrtv_post_rgb48 alone at 3840x2160 (un-profiled),
with the HBM bytes each moves.  HDRTV_VARIANTS=pre_split=1 selects the two-kernel preprocess."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "hdr-realtime-video-pipeline_amd")]
from hdrtv_mi355x import lib as L, weights as W  # noqa: E402
from hdrtv_mi355x.processor import HDRTVNetMI355X  # noqa: E402

H, Wd = 2160, 3840
p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
p._ensure_buffers(H, Wd)
fr = torch.from_numpy(W.synthetic_frame(H, Wd, seed=1, kind="noise")).cuda()
out32 = torch.rand((3, H, Wd), dtype=torch.float32, device="cuda")
u16 = torch.empty((H, Wd, 3), dtype=torch.uint16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


t = timeit(lambda: p._lib.hdrtv_preprocess(p._ctx, st, fr.data_ptr(), H, Wd, p._gpu_input.data_ptr(), p._gpu_cond.data_ptr()))
mb = (H * Wd * 3 + H * Wd * 6 + (H // 4) * (Wd // 4) * 6) / 1e6
print(f"preprocess ({'split' if 'pre_split=1' in os.environ.get('HDRTV_VARIANTS', '') else 'fused'}): {t:.1f} us, {mb:.1f} MB algorithmic -> {mb / t * 1e3:.0f} GB/s")
t = timeit(lambda: p._lib.hdrtv_post_rgb48(p._ctx, st, out32.data_ptr(), L.F32, H, Wd, u16.data_ptr()))
mb = H * Wd * 18 / 1e6
print(f"post_rgb48 (f32 in): {t:.1f} us, {mb:.1f} MB -> {mb / t * 1e3:.0f} GB/s")
t = timeit(lambda: p._lib.hdrtv_post_pq_rgb48(p._ctx, st, out32.data_ptr(), L.F32, H, Wd, C.c_float(1000.0), u16.data_ptr()))
print(f"post_pq_rgb48 (f32 in, exact table): {t:.1f} us -> {mb / t * 1e3:.0f} GB/s")
p.close()
