"""Which tensors depend on the workgroup count?  usage: python tools/dbg/ncu_invariance.py [int8|fp16] [H W] [ncu]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "hdr-realtime-video-pipeline_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
mode = sys.argv[1] if len(sys.argv) > 1 else "int8"
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2160, 3840)
ncu = int(sys.argv[4]) if len(sys.argv) > 4 else 128
g = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
if mode == "int8":
    p = HDRTVNetMI355X(os.path.join(g, "hr_int8_full_qat.hdrw"), precision="int8-full", predequantize="off", use_hg=True, hg_weights="seeded-w8a8-minmax:1234", warmup_passes=0)
    taps = ["agcm.out", "le.cond", "le.cond1", "le.cond2", "le.cond3", "le.cond4", "le.fea0", "le.fea1a", "le.fea1", "le.fea2", "le.fea3", "le.t3y", "le.up1", "le.t4", "le.up2", "le.t5", "le.out", "hg8.p1", "hg8.conv2", "hg8.conv3_2", "hg8.conv4_2", "hg8.conv5_2", "hg8.conv_code2", "hg8.conv6", "hg8.conv7", "hg8.conv8", "hg8.conv9", "hg.part", "hg.part2"]
else:
    p = HDRTVNetMI355X(os.path.join(g, "hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    taps = ["agcm.out", "le.cond", "le.cond1", "le.cond2", "le.cond3", "le.cond4", "le.fea0", "le.fea1a", "le.fea1", "le.t5", "le.out", "hg.conv2", "hg.conv9", "hg.part", "hg.part2"]
f = W.synthetic_frame(h, w, seed=12, kind="gradient")
res = []
for n in (0, ncu, 0):
    p.set_variant("force_ncu", n)
    out, agcm = p.infer(p.preprocess(f))
    res.append([out.clone()] + [p.tap(t).clone() for t in taps])
for name, a, b, c in zip(["out"] + taps, *res):
    print(f"{name:16s} default vs {ncu}: {int((a != b).sum()):9d} differ of {a.numel()} | default vs default again: {int((a != c).sum())}")
p.close()
