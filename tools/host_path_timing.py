#!/usr/bin/env python3
"""Where the host-side milliseconds of one 4K frame go (worker path): run on the GPU box.
usage: python tools/host_path_timing.py [W H]"""
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import numpy as np, torch
from hdrtv_mi355x import weights as Wt
from hdrtv_mi355x.worker import HeadlessPipelineWorker
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, "original"))
os.symlink(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), os.path.join(d, "original", "HR.hdrw"))
w = HeadlessPipelineWorker(d, use_hg=True, proc_w=W, proc_h=H, hg_weights="seeded:1234", buffer_frames=1)
assert w._load_model("FP16")
p = w._processor
frames = [Wt.synthetic_frame(H, W, i, "noise") for i in range(3)]
def T(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def add(k, dt): acc.setdefault(k, []).append(dt * 1e3)
_im = torch.inference_mode(); _im.__enter__()
for it in range(12):
    f = frames[it % 3]
    t0 = T(); p._pin_input.copy_(torch.from_numpy(np.ascontiguousarray(f))); t1 = time.perf_counter(); add("pin memcpy", t1 - t0)
    p._gpu_raw.copy_(p._pin_input, non_blocking=True); t2 = T(); add("H2D", t2 - t1)
    t, c = p.preprocess(f); t3 = T(); add("preprocess() total", t3 - t2)
    out = p.infer((t, c)); t4 = T(); add("infer", t4 - t3)
    st = w._stage_hdr_display_tensor(out[0]); t5 = T(); add("stage D2D", t5 - t4)
    pl = w._tensor_to_rgb48_bytes(st); pl.wait_ready(); t6 = time.perf_counter(); add("rgb48 -> pinned ring", t6 - t5)
    v = pl.buffer_view(); n = open("/dev/null", "wb").write(v); t7 = time.perf_counter(); add("sink write", t7 - t6)
    pl.release()
    t8 = time.perf_counter(); r = w._process_frame(frame=f, frame_idx=it, mpv_w=None); t9 = time.perf_counter(); add("_process_frame(cpu out)", t9 - t8)
for k, v in acc.items():
    print(f"{k:28s} {np.median(v):8.2f} ms")
w.close()
