#!/usr/bin/env python3
"""Which of the two host transfers costs what beside the network: frames/s of the 4K path with (a) nothing,
(b) pinned uploads one frame ahead on a side stream, (c) RGB48 ring commits (device -> pinned host) on a copy stream,
(d) both.  usage: python tools/pcie_overlap.py [--int8]"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "hdr-realtime-video-pipeline_amd")]
import torch  # noqa: E402
from hdrtv_mi355x import lib as L, weights as W  # noqa: E402
from hdrtv_mi355x.processor import HDRTVNetMI355X  # noqa: E402

H, Wd, N = 2160, 3840, 40
int8 = "--int8" in sys.argv
dev = torch.device("cuda", 0)
proc = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), use_hg=True,
                      hg_weights="seeded-w8a8:1234" if int8 else "seeded:1234", warmup_passes=0)
proc._ensure_buffers(H, Wd)
lib, ctx = proc._lib, proc._ctx
frames = [W.synthetic_frame(H, Wd, seed=7 + i, kind="gradient") for i in range(2)]
pin = [torch.from_numpy(f).pin_memory() for f in frames]
dfr = [torch.from_numpy(f).to(dev) for f in frames]
rgb = torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev)
proc._chk(lib.hdrtv_ring_create(ctx, 3, H, Wd), "ring")
up, dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(upload, download):
    main = torch.cuda.current_stream(dev)
    st = C.c_void_p(main.cuda_stream)
    pending, t0 = [], None
    for i in range(N + 5):
        if i == 5:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        if upload:
            with torch.cuda.stream(up):
                dfr[i % 2].copy_(pin[i % 2], non_blocking=True)
                ev = torch.cuda.Event(); ev.record(up)
            main.wait_event(ev)
        dst = rgb.data_ptr()
        if download:
            hp, dp = C.c_void_p(), C.c_void_p()
            slot = proc._chk(lib.hdrtv_ring_acquire(ctx, 250, C.byref(hp), C.byref(dp)), "acq")
            dst = dp.value
        proc._chk(lib.hdrtv_preprocess(ctx, st, dfr[i % 2].data_ptr(), H, Wd, proc._gpu_input.data_ptr(), proc._gpu_cond.data_ptr()), "pre")
        proc._chk(lib.hdrtv_infer(ctx, st, proc._gpu_input.data_ptr(), proc._gpu_cond.data_ptr(), H, Wd, proc._gpu_out.data_ptr(), L.F32,
                                  proc._gpu_agcm.data_ptr()), "infer")
        proc._chk(lib.hdrtv_post_rgb48(ctx, st, proc._gpu_out.data_ptr(), L.F32, H, Wd, dst), "post")
        if download:
            e2 = torch.cuda.Event(); e2.record(main)
            dn.wait_event(e2)
            proc._chk(lib.hdrtv_ring_commit(ctx, slot, C.c_void_p(dn.cuda_stream)), "commit")
            pending.append(slot)
            if len(pending) == 2:
                s0 = pending.pop(0)
                lib.hdrtv_ring_wait(ctx, s0); lib.hdrtv_ring_release(ctx, s0)
    for s0 in pending:
        lib.hdrtv_ring_wait(ctx, s0); lib.hdrtv_ring_release(ctx, s0)
    torch.cuda.synchronize(dev)
    return N / (time.perf_counter() - t0)


for name, u, d in (("compute only", 0, 0), ("+ uploads", 1, 0), ("+ ring downloads", 0, 1), ("+ both", 1, 1), ("compute only", 0, 0)):
    print(f"{name:18s} {run(u, d):7.1f} frames/s")
proc.close()
