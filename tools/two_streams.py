#!/usr/bin/env python3
"""Throughput with two frames in flight (two contexts, two streams) against one: how much of the per-kernel tail and
small-grid idle time a second stream can fill.  usage: python tools/two_streams.py [steps]"""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import torch
from hdrtv_mi355x import lib as L, weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H, Wd = 2160, 3840
dev = torch.device("cuda", 0)
procs = [HDRTVNetMI355X(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0) for _ in range(2)]
for p in procs: p._ensure_buffers(H, Wd)
frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient")).to(dev) for i in range(4)]
rgb = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def step(i, k):
    p = procs[k]; st = C.c_void_p(streams[k].cuda_stream); fr = frames[i % 4]
    p._chk(p._lib.hdrtv_preprocess(p._ctx, st, fr.data_ptr(), H, Wd, p._gpu_input.data_ptr(), p._gpu_cond.data_ptr()), "pre")
    p._chk(p._lib.hdrtv_infer(p._ctx, st, p._gpu_input.data_ptr(), p._gpu_cond.data_ptr(), H, Wd, p._gpu_out.data_ptr(), L.F32, p._gpu_agcm.data_ptr()), "infer")
    p._chk(p._lib.hdrtv_post_rgb48(p._ctx, st, p._gpu_out.data_ptr(), L.F32, H, Wd, rgb[k].data_ptr()), "post")
for mode in (1, 2, 1, 2):
    for i in range(6): step(i, i % mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i, i % mode)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"frames in flight {mode}: {steps / dt:7.2f} frames/s  ({dt / steps * 1e3:.3f} ms per frame)")
