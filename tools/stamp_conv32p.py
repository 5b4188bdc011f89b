#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of one conv32p launch (needs a `make -C csrc STAMP=1 -B` build).
usage: python tools/stamp_conv32p.py LE.HR_conv1 [H W]"""
import os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
layer = sys.argv[1]
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
import torch
from hdrtv_mi355x import weights as Wt
from hdrtv_mi355x.processor import HDRTVNetMI355X
if "HDRTV_STAMP_LAUNCH" not in os.environ:
    p = HDRTVNetMI355X(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    f = Wt.synthetic_frame(H, W, 1, "noise")
    t, c = p.preprocess(f); p.profile_enable(True); p.infer((t, c)); torch.cuda.synchronize()
    names = [r[0] for r in p.profile_read()]
    idx = names.index(layer)
    print("launch index", idx, "ms", [r for r in p.profile_read() if r[0] == layer][0][2])
    p.close()
    env = dict(os.environ, HDRTV_STAMP_LAUNCH=str(idx))
    sys.exit(subprocess.call([sys.executable, __file__] + sys.argv[1:], env=env))
p = HDRTVNetMI355X(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), use_hg=False, warmup_passes=0)
f = Wt.synthetic_frame(H, W, 1, "noise")
t, c = p.preprocess(f)
for _ in range(3): p.infer((t, c))
torch.cuda.synchronize()
raw = p.tap("dbg.stamps").numpy().ravel().astype(np.float32).view(np.uint32)
# tap() converts f32 storage -> float; re-read raw bytes instead
import ctypes as C
ptr, cc, hh, ww, lay = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
p._lib.hdrtv_get_tap(p._ctx, b"dbg.stamps", C.byref(ptr), C.byref(cc), C.byref(hh), C.byref(ww), C.byref(lay))
buf = torch.empty(cc.value, dtype=torch.float32, device="cuda")
from hdrtv_mi355x.processor import _hip_memcpy_d2d
_hip_memcpy_d2d(buf.data_ptr(), ptr.value, cc.value * 4); torch.cuda.synchronize()
st = buf.cpu().numpy().view(np.uint64).reshape(-1, 8).astype(np.float64)
if "conv32_old=1" in os.environ.get("HDRTV_VARIANTS", ""):
    names = ["0 offsets+resid", "1 conv MFMA", "2 staging write", "3 wait vmcnt", "4 barrier1", "5 DMA issue+stores", "6 SFT(t+1)", "7 barrier2+loop"]
else:       # conv32s.hip
    names = ["0 addr+resid+DMA issue", "1 conv MFMA", "2 SFT(t+1)", "3 epilogue+vmcnt(0)", "4 closing wait", "5 barrier", "6 -", "7 loop"]
    # (role split: waves 0-3 have no phase 2, waves 4-7 no phases 1 and 3)
for sel, rows in (("all waves", st), ("waves 0-3 (conv waves / MFMA first)", st.reshape(-1, 8, 8)[:, :4].reshape(-1, 8) if len(st) % 8 == 0 else st),
                  ("waves 4-7 (prep waves / SFT first)", st.reshape(-1, 8, 8)[:, 4:].reshape(-1, 8) if len(st) % 8 == 0 else st)):
    tot = rows.sum(1).mean()
    print(f"{sel}: sampled {len(rows)}, mean cycles per wave {tot:.0f}")
    for i, n in enumerate(names):
        print(f"  {n:24s} {rows[:, i].mean():12.0f}  {100 * rows[:, i].mean() / tot:5.1f} %   (min {rows[:, i].min():.0f} max {rows[:, i].max():.0f})")
p.close()
