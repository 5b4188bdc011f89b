"""Are the fused-vs-per-layer differences systematic or do the runs themselves vary?  Three runs of each form, all pairs compared."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
mode = sys.argv[1] if len(sys.argv) > 1 else "int8"
h, w = 2160, 3840
g = os.path.join(REPO, "tests", "golden")
if mode == "int8":
    p = HDRTVNetMI355X(os.path.join(g, "hr_int8_full_qat.hdrw"), precision="int8-full", predequantize="off", use_hg=False, warmup_passes=0)
    var, taps = "le_rows", ["le.fea0", "le.fea1a", "le.fea1", "le.out"]
else:
    p = HDRTVNetMI355X(os.path.join(g, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    var, taps = "cond2_fused", ["le.cond2", "le.out"]
f = W.synthetic_frame(h, w, seed=41, kind="gradient")
runs = []
for v in (1, 0, 1, 0, 1, 0):
    p.set_variant(var, v)
    out, _ = p.infer(p.preprocess(f))
    runs.append((v, [p.tap(t).clone() for t in taps]))
for ti, t in enumerate(taps):
    line = []
    for a in range(6):
        for b in range(a + 1, 6):
            line.append(f"{runs[a][0]}{runs[b][0]}:{int((runs[a][1][ti] != runs[b][1][ti]).sum())}")
    print(f"{t:10s} pairs (variant values of the two runs : differing values)  " + " ".join(line))
p.close()
