#!/usr/bin/env python3
"""Golden vectors at the BASELINE sizes, by RUNNING THE REFERENCE (build container only: needs /root/reference, CPU fp32).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_fullsize.py

  * full_1080x1920_hg_s11.npz   HDRTVNetTorch (HR.pt + the seeded HG head) at 1920x1080 -- configs[1], the size at which the
                                reference switches its modules to the aligned fast graph (hdrtvnet_torch.py:204-226,
                                HDRUNet3T1._forward_assume_aligned, HDRUNet3T1_arch.py:106-150): asserted below
  * full_2160x3840_hg_s12.npz   the same model at 3840x2160 -- configs[2], the headline configuration (the safe-aligned graph:
                                3840x2160 is not in the reference's aligned set)
A fixture holds the input frame in a compact form (seed + kind: tests rebuild it with weights.synthetic_frame), strided samples
of the outputs (every 8th row / 16th column at 1080p, 12th / 24th at 4K: out, agcm_out, base, mask, rgb48, u8), two dense patches,
a summary of the whole tensors (per-channel mean / mean |x| / min / max) and the per-channel sums of all RGB48 integers, < 1 MB each.
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("HDRTV_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))

from models.hdrtvnet_torch import HDRTVNetTorch  # noqa: E402  (the reference)
from hdrtv_mi355x import weights as W  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
HR_PT = os.path.join(REF, "src/models/weights/original/HR.pt")
torch.set_num_threads(8)


def rgb48_reference(out_tensor: torch.Tensor) -> np.ndarray:
    """feeders.py:196 + 223-227, same op order, on CPU tensors."""
    rgb = out_tensor.squeeze(0).permute(1, 2, 0)
    rgb_f32 = torch.empty(tuple(rgb.shape), dtype=torch.float32)
    rgb_f32.copy_(rgb)
    rgb_f32.clamp_(0.0, 1.0).mul_(65535.0).add_(0.5)
    rgb_u16 = torch.empty(tuple(rgb.shape), dtype=torch.uint16)
    rgb_u16.copy_(rgb_f32)
    return rgb_u16.numpy().copy()


def summary(x):
    x = np.asarray(x, np.float64)
    return np.stack([x.mean((1, 2)), np.abs(x).mean((1, 2)), x.min((1, 2)), x.max((1, 2))]).astype(np.float64)


def run(proc, h, w, seed, kind, want_aligned, with_hg, RS, CS):
    frame = W.synthetic_frame(h, w, seed=seed, kind=kind)
    got = {}
    hooks = []
    if with_hg:
        mods = dict(proc.model.named_modules())

        def hook(_m, _i, o):
            got["base"] = (o[0] if isinstance(o, (tuple, list)) else o).detach().clone()
        hooks.append(mods["base"].register_forward_hook(hook))
    t0 = time.time()
    with torch.inference_mode():
        tensor, cond = proc.preprocess(frame)
        flags = [bool(m.assume_aligned_shapes) for m in proc.model.modules() if hasattr(m, "assume_aligned_shapes")]
        assert flags and all(f == want_aligned for f in flags), (flags, want_aligned)
        out = proc.infer((tensor.clone(), cond.clone()))
        out0, agcm = out[0].clone(), out[1].clone()
        rgb48 = rgb48_reference(out0.clone())
        u8 = proc.postprocess((out0.clone(), agcm)).copy()
    for hk in hooks:
        hk.remove()
    print(f"  {h}x{w}: reference forward {time.time() - t0:.1f} s, aligned fast graph = {flags[0]}")
    o, a = out0.numpy()[0], agcm.numpy()[0]
    res = dict(shape=np.array([h, w]), seed=np.array(seed), kind=np.array(kind), aligned=np.array(flags[0]),
               stride=np.array([RS, CS]),
               out=o[:, ::RS, ::CS].copy(), agcm_out=a[:, ::RS, ::CS].copy(), rgb48=rgb48[::RS, ::CS].copy(), u8_bgr=u8[::RS, ::CS].copy(),
               out_summary=summary(o), agcm_summary=summary(a),
               # a dense corner and a dense centre patch: the strided samples never see neighbouring pixels
               out_corner=o[:, :32, :48].copy(), out_centre=o[:, h // 2 - 16:h // 2 + 16, w // 2 - 24:w // 2 + 24].copy(),
               rgb48_sum=np.array([int(rgb48[..., c].astype(np.int64).sum()) for c in range(3)]))
    if with_hg:
        base = got["base"].numpy()[0]
        m = base.max(0, keepdims=True)
        mask = ((((m - 0.75) / 0.25).clip(0, 1)) > 0.1)
        res.update(base=base[:, ::RS, ::CS].copy(), base_summary=summary(base), mask=mask[:, ::RS, ::CS].copy(),
                   mask_count=np.array(int(mask.sum())))
    return res


def main():
    print("1920x1080 HR + HG (seeded) through the aligned fast graph")
    hg_state = W.seeded_hg_state(1234)
    with tempfile.TemporaryDirectory() as td:
        hg_path = os.path.join(td, "HG_seeded.pt")
        torch.save({k: torch.from_numpy(np.array(v)) for k, v in hg_state.items()}, hg_path)
        proc = HDRTVNetTorch(HR_PT, device="cpu", precision="fp32", compile_model=False, use_hg=True, hg_weights=hg_path, warmup_passes=0)
    assert type(proc.model).__name__ == "HG_Composite"
    r = run(proc, 1080, 1920, 11, "gradient", True, True, 8, 16)
    np.savez_compressed(os.path.join(OUT, "full_1080x1920_hg_s11.npz"), **r)
    print("3840x2160 HR + HG (seeded)")
    r = run(proc, 2160, 3840, 12, "gradient", False, True, 12, 24)
    np.savez_compressed(os.path.join(OUT, "full_2160x3840_hg_s12.npz"), **r)
    for fn in ("full_1080x1920_hg_s11.npz", "full_2160x3840_hg_s12.npz"):
        print(f"{os.path.getsize(os.path.join(OUT, fn)):>9d}  {fn}")


if __name__ == "__main__":
    main()
