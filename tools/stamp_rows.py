#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of one le_rows.hip launch (needs a `make -C csrc STAMP=1` build: tools/stamp_run.sh
with TOOL=stamp_rows.py).  usage: python tools/stamp_rows.py LE.recon_trunk1.0 [H W]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
layer = sys.argv[1]
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
import torch  # noqa: E402
from hdrtv_mi355x import weights as Wt  # noqa: E402
from hdrtv_mi355x.processor import HDRTVNetMI355X, _hip_memcpy_d2d  # noqa: E402

p = HDRTVNetMI355X(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), use_hg=False, warmup_passes=0)
f = Wt.synthetic_frame(H, W, 1, "noise")
t, c = p.preprocess(f)
if "HDRTV_STAMP_LAUNCH" not in os.environ:
    p.profile_enable(True)
    p.infer((t, c))
    torch.cuda.synchronize()
    prof = p.profile_read()
    idx = [r[0] for r in prof].index(layer)
    print("launch index", idx, "kernel", prof[idx][1], "ms", prof[idx][2])
    p.close()
    sys.exit(subprocess.call([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, HDRTV_STAMP_LAUNCH=str(idx))))
for _ in range(3):
    p.infer((t, c))
torch.cuda.synchronize()
ptr, cc, hh, ww, lay = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
p._lib.hdrtv_get_tap(p._ctx, b"dbg.stamps", C.byref(ptr), C.byref(cc), C.byref(hh), C.byref(ww), C.byref(lay))
buf = torch.empty(cc.value, dtype=torch.float32, device="cuda")
_hip_memcpy_d2d(buf.data_ptr(), ptr.value, cc.value * 4)
torch.cuda.synchronize()
st = buf.cpu().numpy().view(np.uint64).reshape(-1, 8, 8).astype(np.float64)       # [workgroup][wave][phase]
st = st[st.sum((1, 2)) > 0]
names = {"B (waves 0-3)": ["0 DMA issue", "1 conv", "2 epilogue a / sft + write", "3 epilogue b", "4 closing wait", "5 barrier", "6 -", "7 loop"],
         "C (waves 4-7)": ["0 DMA issue", "1 conv", "2 sft + write", "3 epilogue + stores", "4 closing wait", "5 barrier", "6 -", "7 loop"]}
for (role, nm), rows in zip(names.items(), (st[:, :4].reshape(-1, 8), st[:, 4:].reshape(-1, 8))):
    tot = rows.sum(1).mean()
    print(f"role {role}: {len(rows)} waves, mean cycles per wave {tot:.0f}")
    for i, n in enumerate(nm):
        print(f"  {n:28s} {rows[:, i].mean():12.0f}  {100 * rows[:, i].mean() / tot:5.1f} %   (min {rows[:, i].min():.0f} max {rows[:, i].max():.0f})")
p.close()
