#!/usr/bin/env python3
"""Probe: do two frame pipelines on disjoint CU subsets (HDRTV_VARIANTS=force_ncu=N) out-run one pipeline on the whole chip?
LE is HBM / latency bound, the HG convs are MFMA / power bound, so a second frame's LE might hide under the first frame's HG.
usage: python tools/overlap_probe.py K NCU [frames]   (K contexts, each launched with NCU persistent workgroups)"""
import ctypes as C
import os
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
K, NCU = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 150
os.environ["HDRTV_VARIANTS"] = "force_ncu=" + str(NCU)
import contextlib

import torch

from hdrtv_mi355x import lib as L
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X

H, Wd = 2160, 3840
dev = torch.device("cuda:0")
procs = []
with contextlib.redirect_stdout(sys.stderr):
    for k in range(K):
        p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), device="cuda:0", precision="auto", use_hg=True,
                           hg_weights="seeded:1234", warmup_passes=0)
        p._ensure_buffers(H, Wd)
        procs.append(p)
frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient")).to(dev) for i in range(4)]
outs = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(K)]
streams = [torch.cuda.Stream(dev) for _ in range(K)]
bar = threading.Barrier(K + 1)


def loop(k, n):
    p, s = procs[k], C.c_void_p(streams[k].cuda_stream)
    lib, ctx = p._lib, p._ctx
    for i in range(n):
        fr = frames[(i + k) % 4]
        p._chk(lib.hdrtv_preprocess(ctx, s, fr.data_ptr(), H, Wd, p._gpu_input.data_ptr(), p._gpu_cond.data_ptr()), "pre")
        p._chk(lib.hdrtv_infer(ctx, s, p._gpu_input.data_ptr(), p._gpu_cond.data_ptr(), H, Wd, p._gpu_out.data_ptr(), L.F32,
                               p._gpu_agcm.data_ptr()), "infer")
        p._chk(lib.hdrtv_post_rgb48(ctx, s, p._gpu_out.data_ptr(), L.F32, H, Wd, outs[k].data_ptr()), "post")
        if i % 8 == 7:
            streams[k].synchronize()          # bound the queue depth like the ring does


def run(k):
    loop(k, 10)
    streams[k].synchronize()
    bar.wait()
    loop(k, N)
    streams[k].synchronize()
    bar.wait()


ths = [threading.Thread(target=run, args=(k,)) for k in range(K)]
for t in ths:
    t.start()
bar.wait()
t0 = time.perf_counter()
bar.wait()
dt = time.perf_counter() - t0
for t in ths:
    t.join()
print(f"contexts={K} ncu={NCU} frames={K * N} fps={K * N / dt:.2f}")
