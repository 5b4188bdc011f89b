"""All six HR INT8 checkpoints the reference ships (src/gui_config.py:40-93 presets; four of them new in round 3:
``HR_original_int8_{full,mixed}.pt`` -- PTQ -- and ``..._qat_film.pt``), in both of the reference's execution modes.
Goldens: tests/golden/gen_golden_int8_all.py ran the reference itself on CPU (``predequantize`` "auto" and "off").

CPU part (not gpu): the oracle reproduces both runs.  GPU part: the device path with ``predequantize="auto"`` (fp16 convs of
the dequantised weights: the fp16 bars of test_gpu_parity.py) and ``predequantize="off"`` (W8A8 layers on int8 MFMA: the
bars of test_gpu_int8_hr.py; the reference's own bound for a re-quantised graph is u8 MAE <= 5)."""
import os

import numpy as np
import pytest

TAGS = ["full", "mixed", "full_qat_film", "mixed_qat_film"]
PREC = {"full": "int8-full", "mixed": "int8-mixed", "full_qat_film": "int8-full", "mixed_qat_film": "int8-mixed"}


def _frame():
    from hdrtv_mi355x import weights as W
    return W.synthetic_frame(64, 96, seed=6, kind="gradient")


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_the_reference_runs(golden_dir, tag):
    from hdrtv_mi355x import weights as W
    from oracle import hdrtvnet_oracle as O
    st = W.load_pack(os.path.join(golden_dir, f"hr_int8_{tag}.hdrw"))
    d = np.load(os.path.join(golden_dir, f"int8_{tag}_both_64x96_gradient_s6.npz"))
    kinds = dict(x.split("=") for x in d["layer_kinds"])
    n_w8a8 = sum(1 for v in kinds.values() if v.startswith("W8A8"))
    assert n_w8a8 == (128 if tag.startswith("full") else 29) and all(v.endswith(":asym") for v in kinds.values() if v.startswith("W8A8"))
    t, c = O.preprocess(_frame())
    # predequantize "auto": int8 * scale in fp32, activation fake-quant dropped (hdrtvnet_torch.py:444-462)
    out, a = O.hr_forward(W.dequantize_int8_state(st, "fp32"), t, c)
    assert np.abs(out - d["auto:out"]).max() <= 2e-5 and np.abs(a - d["auto:agcm_out"]).max() <= 2e-5
    assert np.abs(O.postprocess_u8(out).astype(int) - d["auto:u8_bgr"].astype(int)).max() <= 1
    assert np.array_equal(O.post_rgb48(d["auto:out"]), d["auto:rgb48"])
    # predequantize "off": the fake-quant layers; ATen's conv under the oracle's graph is the reference's arithmetic
    q = O.w8a8_state(st)
    assert sum(1 for v in q.values() if getattr(v, "x_scale", None) is not None) == n_w8a8
    O.use_backend("aten")
    try:
        out, a = O.hr_forward(q, t, c)
    finally:
        O.use_backend("c")
    e = np.abs(out - d["off:out"])
    mae = np.abs(O.postprocess_u8(out).astype(int) - d["off:u8_bgr"].astype(int)).mean()
    print(f"  {tag} fake-quant, ATen operators: out max {e.max():.2e} mean {e.mean():.2e} u8 MAE {mae:.3f}")
    if tag.startswith("full"):
        assert e.max() <= 1e-5 and mae == 0.0
    else:                       # the mixed recipe's fp layers differ by an ulp, which flips isolated quantisation steps
        assert e.max() <= 3e-2 and e.mean() <= 1.5e-3 and mae <= 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_device_runs_every_shipped_checkpoint(golden_dir, tag):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    d = np.load(os.path.join(golden_dir, f"int8_{tag}_both_64x96_gradient_s6.npz"))
    f = _frame()
    path = os.path.join(golden_dir, f"hr_int8_{tag}.hdrw")
    for mode in ("auto", "off"):
        p = HDRTVNetMI355X(path, precision=PREC[tag], predequantize=mode, use_hg=False, warmup_passes=0)
        try:
            assert p._is_w8_model is (mode == "off")
            out, agcm = p.infer(p.preprocess(f))
            out_np = out.float().cpu().numpy()[0]
            u8 = p.postprocess(out).astype(int)
            launches = p.infer_stats()[0]
            p.profile_enable(True)
            p.infer(p.preprocess(f))
            kernels = [k for _, k, _, _, _ in p.profile_read()]
            p.profile_enable(False)
        finally:
            p.close()
        e = np.abs(out_np - d[f"{mode}:out"])
        du8 = np.abs(u8 - d[f"{mode}:u8_bgr"].astype(int))
        n_i8 = sum(("i8" in k) or ("q8" in k) or ("<q" in k) for k in kernels)
        print(f"  {tag} predequantize={mode}: out max {e.max():.3e} mean {e.mean():.3e}; u8 max {du8.max()} MAE {du8.mean():.4f}; "
              f"{n_i8} of {launches} launches on int8 MFMA")
        if mode == "auto":
            assert n_i8 == 0 and e.max() <= 6e-3 and e.mean() <= 5e-4 and du8.max() <= 3 and du8.mean() <= 0.6
        else:
            assert n_i8 >= 25 and du8.mean() <= 5.0
            if tag.startswith("full"):          # 128 quantisers amplify single code flips (test_gpu_int8_hr.py's bars)
                assert e.mean() <= 2e-2 and du8.mean() <= 3.0
            else:
                assert e.max() <= 6e-2 and e.mean() <= 2e-3 and du8.mean() <= 0.5
