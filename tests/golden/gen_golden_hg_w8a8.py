#!/usr/bin/env python3
"""W8A8 HG head (BASELINE.json configs[4]; the reference's int8 HG checkpoints are not shipped):

1. calibrate: activation ranges of the seeded fp32 HG head on synthetic gradient frames (the oracle evaluates the
   network; ranges go to hdrtv_mi355x/data/hg_w8a8_calib_seed1234.json, the recipe the product quantises with);
2. golden: RUN THE REFERENCE -- its own HG_Composite with the 18 layers of weights.HG_W8A8_GROUPS replaced by its own
   W8A8Conv2d (hdrtvnet_torch.py:296-364, asymmetric, fp32 compute on CPU) carrying those quantiser values -- on one
   frame, and keep inputs, outputs and a few taps.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_hg_w8a8.py

Data only is committed."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up the reference import path)
import torch  # noqa: E402

sys.path.insert(0, G.REPO)
from hdrtv_mi355x import weights as W  # noqa: E402
from oracle import hdrtvnet_oracle as O  # noqa: E402

SEED = 1234


def calibrate():
    hr = {k: np.asarray(v, np.float32) for k, v in W.load_pack(os.path.join(HERE, "hr_weights.hdrw")).items()}
    hg = W.seeded_hg_state(SEED)
    lo, hi = {}, {}

    def see(group, *tensors):
        for t in tensors:
            lo[group] = min(lo.get(group, 0.0), float(t.min()))
            hi[group] = max(hi.get(group, 0.0), float(t.max()))

    for seed in (11, 12, 3):
        f = W.synthetic_frame(288, 480, seed=seed, kind="gradient")
        base, _ = O.hr_forward(hr, *O.preprocess(f))
        mask = O.hg_mask(base)
        del mask
        c1 = O._hg_block(hg, "conv1", base)
        p1 = O.maxpool2(c1)
        c2 = O._hg_block(hg, "conv2", p1)
        p3 = O.maxpool2(O._hg_block(hg, "conv3_1", c2))
        c3 = O._hg_block(hg, "conv3_2", p3)
        p4 = O.maxpool2(O._hg_block(hg, "conv4_1", c3))
        c4 = O._hg_block(hg, "conv4_2", p4)
        p5 = O.maxpool2(O._hg_block(hg, "conv5_1", c4))
        c5 = O._hg_block(hg, "conv5_2", p5)
        pc = O.maxpool2(O._hg_block(hg, "conv_code1", c5))
        code = O._hg_block(hg, "conv_code2", pc)

        def fuse(name, a, b):
            return O.conv2d(np.concatenate((a, b), axis=0), hg[name + ".weight"], hg[name + ".bias"])

        u1 = O._hg_up(hg, "Up_conv1", code)
        c6 = fuse("conv6", u1, c5)
        u2 = O._hg_up(hg, "Up_conv2", c6)
        c7 = fuse("conv7", u2, c4)
        u3 = O._hg_up(hg, "Up_conv3", c7)
        c8 = fuse("conv8", u3, c3)
        u4 = O._hg_up(hg, "Up_conv4", c8)
        c9 = fuse("conv9", u4, c2)
        see("p1", p1); see("conv2+up4", c2, u4); see("p3", p3); see("conv3_2+up3", c3, u3); see("p4", p4); see("conv4_2+up2", c4, u2)
        see("p5", p5); see("conv5_2+up1", c5, u1); see("pc", pc); see("conv_code2", code)
        see("conv6", c6); see("conv7", c7); see("conv8", c8); see("conv9", c9)
    ranges = {g: (round(lo[g], 6), round(hi[g], 6)) for g in W.HG_W8A8_GROUPS}
    path = os.path.join(G.REPO, "hdr-realtime-video-pipeline_amd", "hdrtv_mi355x", "data", f"hg_w8a8_calib_seed{SEED}.json")
    with open(path, "w") as fjs:
        json.dump({"comment": "min/max of the seeded fp32 HG head's activations on synthetic gradient frames "
                              "(288x480, seeds 11, 12, 3); written by tests/golden/gen_golden_hg_w8a8.py",
                   "ranges": ranges}, fjs, indent=1)
    return ranges


def main(integer_zero=True):
    """integer_zero: x_zero rounded to a whole number of steps (the round-1 table) or the calibration's float minimum (what
    the reference's calibrate_w8a8 produces: x_zero = running min)."""
    ranges = calibrate()
    print(ranges)
    qstate = W.hg_w8a8_state(W.seeded_hg_state(SEED), ranges, integer_zero=integer_zero)

    from models.hdrtvnet_torch import W8A8Conv2d   # the reference's layer
    hg_state = W.seeded_hg_state(SEED)
    with tempfile.TemporaryDirectory() as td:
        hg_path = os.path.join(td, "HG_seeded.pt")
        torch.save({k: torch.from_numpy(np.array(v)) for k, v in hg_state.items()}, hg_path)
        proc = G.make_proc(use_hg=True, hg_path=hg_path)
    hg = proc.model.hg
    mods = dict(hg.named_modules())
    for layers in W.HG_W8A8_GROUPS.values():
        for name in layers:
            conv = mods[name]
            q = W8A8Conv2d(conv, compute_dtype=torch.float32, asymmetric=True)
            # the layer quantises its weights itself; the product's recipe must give the same integers
            assert np.array_equal(q.weight_int8.numpy(), qstate[name + ".weight_int8"]), name
            assert np.array_equal(q.w_scale.numpy(), qstate[name + ".w_scale"]), name
            q.x_scale.data = torch.tensor(float(qstate[name + ".x_scale"]), dtype=torch.float32)
            q.x_zero.data = torch.tensor(float(qstate[name + ".x_zero"]), dtype=torch.float32)
            parent = mods[name.rsplit(".", 1)[0]] if "." in name else hg
            setattr(parent, name.rsplit(".", 1)[-1], q)
    taps = ["base", "hg.conv2", "hg.conv3_2", "hg.conv5_2", "hg.conv_code2", "hg.conv6", "hg.conv8", "hg.conv9"]
    f = W.synthetic_frame(96, 128, seed=3, kind="gradient")
    r = G.run_case(proc, f, taps)
    base = r["tap:base"]
    m = base.max(0, keepdims=True)
    r["mask"] = ((((m - 0.75) / 0.25).clip(0, 1)) > 0.1).astype(np.float32)
    for k, step in (("tap:hg.conv2", 16), ("tap:hg.conv3_2", 16), ("tap:hg.conv5_2", 4), ("tap:hg.conv_code2", 4),
                    ("tap:hg.conv6", 4), ("tap:hg.conv8", 16), ("tap:hg.conv9", 16)):
        r[k] = r[k][::step].copy()
    for k in ("tensor", "cond", "rgb48"):
        r.pop(k)
    name = "hg_w8a8_96x128_gradient_s3.npz" if integer_zero else "hg_w8a8_floatzero_96x128_gradient_s3.npz"
    np.savez_compressed(os.path.join(HERE, name), **r)
    print("saved", name, {k: getattr(v, "shape", None) for k, v in r.items()})


if __name__ == "__main__":
    main(integer_zero="--float-zero" not in sys.argv)
