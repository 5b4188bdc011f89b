#!/usr/bin/env python3
"""Per-layer times of the precision="fp32" graph at 1920x1080 (or --size HxW), for each value of a variant.
    python tools/fp32_layers.py [--variant f32_rows=1,2] [--top 20]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1080x1920")
    ap.add_argument("--variant", default="f32_narrow_below=1,3")
    ap.add_argument("--top", type=int, default=18)
    a = ap.parse_args()
    import torch
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    h, w = (int(v) for v in a.size.split("x"))
    p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), precision="fp32", use_hg=True,
                       hg_weights="seeded:1234", warmup_passes=0)
    frame = W.synthetic_frame(h, w, seed=3, kind="gradient")
    name, vals = a.variant.split("=")
    for v in (int(x) for x in vals.split(",")):
        p.set_variant(name, v)
        t, c = p.preprocess(frame)
        p.infer((t, c))
        p.profile_enable(True)
        p.infer((t, c))
        torch.cuda.synchronize()
        rows = p.profile_read()
        p.profile_enable(False)
        rows = [dict(layer=r[0], kernel=r[1], ms=r[2], macs=r[3]) for r in rows]
        tot = sum(r["ms"] for r in rows)
        macs = sum(r["macs"] for r in rows)
        print(f"{name}={v}: {tot:.2f} ms in {len(rows)} launches, {2 * macs / tot / 1e9:.1f} TFLOP/s")
        for r in sorted(rows, key=lambda r: -r["ms"])[:a.top]:
            print(f"   {r['layer']:<34s} {r['kernel']:<14s} {r['ms']:7.3f} ms  {2 * r['macs'] / max(r['ms'], 1e-9) / 1e9:7.1f} TFLOP/s")
    p.close()


if __name__ == "__main__":
    main()
