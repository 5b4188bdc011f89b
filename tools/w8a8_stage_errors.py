#!/usr/bin/env python3
"""Developer aid (GPU box): error of every LE stage of a W8A8 run against the oracle's fake-quant graph, end to end and
(second column) with the oracle re-started from the device's own previous tensor.  usage: tools/w8a8_stage_errors.py full|mixed [HxW]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "hdr-realtime-video-pipeline_amd")]
from hdrtv_mi355x import weights as W  # noqa: E402
from hdrtv_mi355x.processor import HDRTVNetMI355X  # noqa: E402
from oracle import hdrtvnet_oracle as O  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "full"
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64x96").split("x"))
path = os.path.join(REPO, "tests", "golden", f"hr_int8_{tag}_qat.hdrw")
p = HDRTVNetMI355X(path, precision=f"int8-{tag}", predequantize="off", use_hg=False, warmup_passes=0)
sd = O.w8a8_state(W.load_pack(path))
f = W.synthetic_frame(h, w, seed=6, kind="gradient")
t, c = p.preprocess(f)
out, agcm = p.infer((t, c))
a = agcm.float().cpu().numpy()[0]
taps = {}
ref = O.le(sd, a, taps)
T = lambda n: p.tap(n).numpy()  # noqa: E731
relu = O.relu
rows = [("le.cond", taps["LE.cond_first"]), ("le.cond1", taps["LE.CondNet1"]), ("le.cond2", taps["LE.CondNet2"]),
        ("le.cond3", taps["LE.CondNet3"]), ("le.cond4", taps["LE.CondNet4"]), ("le.f0a", relu(taps["LE.conv_first"])),
        ("le.fea0", relu(taps["LE.HR_conv1"])), ("le.fea1a", relu(taps["LE.down_conv1"])), ("le.fea1", taps["LE.recon_trunk1"]),
        ("le.fea2a", relu(taps["LE.down_conv2"])), ("le.fea2", taps["LE.recon_trunk2"]), ("le.fea3", relu(taps["LE.down_conv3"])),
        ("le.t3y", taps["LE.recon_trunk3"] + relu(taps["LE.down_conv3"])), ("le.t4", taps["LE.recon_trunk4"]),
        ("le.t5", taps["LE.recon_trunk5"]), ("le.f0b", relu(taps["LE.HR_conv2"]))]
for name, want in rows:
    got = T(name)
    if got.shape != want.shape:
        print(f"{name:10s} shape {got.shape} vs {want.shape}")
        continue
    d = np.abs(got - want)
    print(f"{name:10s} max={d.max():.3e} mean={d.mean():.3e} rel_mean={d.mean() / (np.abs(want).mean() + 1e-12):.3e}")
got = out.float().cpu().numpy()[0]
d = np.abs(got - ref)
print(f"{'out':10s} max={d.max():.3e} mean={d.mean():.3e}")
# last two layers given the device's own input
y = relu(O.conv2d(O.sft(sd, "LE.SFT_layer2", T("le.up3"), T("le.cond1")), sd["LE.HR_conv2.weight"], sd["LE.HR_conv2.bias"], 1, 1))
d = np.abs(T("le.f0b") - y)
print(f"HR_conv2 given device up3/cond1: max={d.max():.3e} mean={d.mean():.3e}")
y = a + O.conv2d(T("le.f0b"), sd["LE.conv_last.weight"], sd["LE.conv_last.bias"], 1, 1)
d = np.abs(got - y)
print(f"conv_last given device f0b: max={d.max():.3e} mean={d.mean():.3e}")
wl = sd["LE.conv_last.weight"]
print("conv_last x_scale", wl.x_scale, "x_zero", wl.x_zero, "| HR_conv2 x_scale", sd["LE.HR_conv2.weight"].x_scale)
