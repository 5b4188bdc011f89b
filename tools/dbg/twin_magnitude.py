"""How far apart are the fused and the per-layer forms where they are not bit-identical?  count, max |a - b| per tensor."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
g = os.path.join(REPO, "tests", "golden")
def run(p, var_sets, taps, frames):
    for (h, w), seed in frames:
        f = W.synthetic_frame(h, w, seed=seed, kind="gradient" if seed % 2 else "noise")
        res = []
        for vs in var_sets:
            for k, v in vs.items():
                p.set_variant(k, v)
            out, _ = p.infer(p.preprocess(f))
            res.append([out.float().clone()] + [p.tap(t).clone() for t in taps])
        print(f"  {h}x{w}: " + " | ".join(f"{n} {int((a != b).sum())}/{a.numel()} max {float((a - b).abs().max()):.3e}" for n, a, b in zip(("out",) + tuple(taps), res[0], res[1])), flush=True)
SIZES = (((2160, 3840), 41), ((1080, 1920), 42), ((1081, 1923), 45), ((720, 1280), 46), ((540, 960), 43), ((270, 486), 47))
print("int8 rows vs per layer (no HG)")
p = HDRTVNetMI355X(os.path.join(g, "hr_int8_full_qat.hdrw"), precision="int8-full", predequantize="off", use_hg=False, warmup_passes=0)
run(p, ({"le_rows": 0}, {"le_rows": 1}), ("le.fea0", "le.fea1", "le.t5"), SIZES); p.close()
print("fp16 cond2/cond3 fused vs separate (no HG)")
p = HDRTVNetMI355X(os.path.join(g, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
run(p, ({"cond2_fused": 0, "cond3_fused": 0}, {"cond2_fused": 1, "cond3_fused": 1}), ("le.cond2", "le.cond3", "le.cond4", "le.fea1"), SIZES + (((60, 100), 48), ((52, 76), 49)))
run(p, ({"cond2_fused": 0, "cond3_fused": 1}, {"cond2_fused": 1, "cond3_fused": 1}), ("le.cond2", "le.cond3", "le.cond4"), SIZES[:2])
p.close()
print("int8 4K with HG: default vs one tile per workgroup")
p = HDRTVNetMI355X(os.path.join(g, "hr_int8_full_qat.hdrw"), precision="int8-full", predequantize="off", use_hg=True, hg_weights="seeded-w8a8:1234", warmup_passes=0)
run(p, ({"force_ncu": 0}, {"force_ncu": 4000000}), ("le.out",), SIZES[:1]); p.close()
