"""Parity evidence for the BASELINE.json configurations as they are benchmarked ("next round" item 1):

  * configs[2]: the HG head at 3840x2160 (2176 padded rows, the 8160-tile schedules of conv_pglds) against the oracle on
    the device's own LE output, once with the reference's mask (mask_r = 0.75) and once dense (0.3);
  * end-to-end RGB48 integers against the ``rgb48`` vectors the reference itself produced for every HR / HG golden
    (gui_pipeline_worker_feeders.py:223-227 applied to the reference's fp32 output), with the LSB histogram SURVEY
    section 7 asks for.  Bit-exactness of the final integers through an fp16 network against an fp32 reference is not
    attainable; the asserted bounds are the float bars of test_gpu_parity.py in u16 LSB;
  * configs[0]'s size, 960x540 (the 135 -> 68 -> 136 ``_align_to`` crop), on the device;
  * configs[4] as benchmarked: native-int8 HR together with the int8 HG head against the oracle's fake-quant composite.

All tests need a real MI355X (``pytest -m gpu``) and call through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OUT_MAX, OUT_MEAN = 3.7e-3, 2.5e-4          # test_gpu_parity.py's float bars for the final output (1.5 x measured)
# RGB48 against the reference's own integers: 1.5 x the worst case measured on the r04 build over every golden below and the
# full-size fixtures of test_gpu_fullsize_reference.py (max 151 LSB, mean 9.5 LSB of 65535), not the float bars converted to LSB
LSB_MAX, LSB_MEAN = 230, 14.5


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    return torch


def _stats(name, got, want):
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    print(f"  {name}: max_abs={d.max():.3e} mean_abs={d.mean():.3e} ref_absmean={np.abs(want).mean():.3e}")
    return d.max(), d.mean()


def _rgb48(proc, out):
    """RGB48 of ``out`` through hdrtv_post_rgb48 (the feeder's quantiser) -> u16 [H,W,3]."""
    import torch
    from hdrtv_mi355x import lib as L
    out = out.contiguous()
    h, w = int(out.shape[-2]), int(out.shape[-1])
    u16 = torch.empty((h, w, 3), dtype=torch.uint16, device=out.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = proc._lib.hdrtv_post_rgb48(proc._ctx, st, out.data_ptr(), L.F32 if out.dtype == torch.float32 else L.F16, h, w,
                                    u16.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return u16.cpu().numpy()


def _lsb_histogram(name, got, want, keep=None):
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    if keep is not None:
        d = d[keep]
    edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 65536]
    hist = [(int(((d >= lo) & (d < hi)).sum())) for lo, hi in zip(edges[:-1], edges[1:])]
    cells = " ".join(f"[{lo},{hi}):{n}" for lo, hi, n in zip(edges[:-1], edges[1:], hist) if n)
    print(f"  RGB48 |LSB error| {name}: n={d.size} exact={hist[0] / d.size:.3%} max={int(d.max())} mean={d.mean():.2f} "
          f"p99={np.percentile(d, 99):.0f}  histogram {cells}")
    return int(d.max()), float(d.mean())


# ------------------------------------------------------------------------------------------ RGB48 vs the reference's vectors
@pytest.mark.parametrize("name", ["hr_64x96_noise_s0", "hr_60x100_noise_s2", "hr_52x76_gradient_s5", "hr_32x96_gradient_s1_taps"])
def test_rgb48_end_to_end_vs_reference_hr(torch_cuda, golden_dir, name):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    try:
        out, _ = p.infer(p.preprocess(d["frame"]))
        got = _rgb48(p, out)
    finally:
        p.close()
    assert got.shape == d["rgb48"].shape and got.dtype == np.uint16
    mx, mean = _lsb_histogram(name, got, d["rgb48"])
    assert mx <= LSB_MAX and mean <= LSB_MEAN


@pytest.mark.parametrize("name", ["hg_96x128_gradient_s3", "hg_80x112_gradient_s4"])
def test_rgb48_end_to_end_vs_reference_hg(torch_cuda, golden_dir, name):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    from oracle import hdrtvnet_oracle as O
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    try:
        out, _ = p.infer(p.preprocess(d["frame"]))
        got = _rgb48(p, out)
        base = p.tap("le.out").numpy()
    finally:
        p.close()
    # the highlight mask is a hard threshold on the LE output: a pixel whose bit flipped under fp16 gets (or loses) the whole
    # HG residual, which is not a rounding error; such pixels are counted, bounded, and excluded from the LSB bars
    same = (O.hg_mask(base) == d["mask"])[0]
    print(f"  mask flips vs the reference run: {int((~same).sum())} of {same.size}")
    assert (~same).mean() <= 0.002
    _lsb_histogram(name + " (all pixels)", got, d["rgb48"])
    mx, mean = _lsb_histogram(name + " (mask bit equal)", got, d["rgb48"], keep=same)
    assert mx <= LSB_MAX and mean <= LSB_MEAN


@pytest.mark.parametrize("tag,prec", [("full_qat", "int8-full"), ("mixed_qat", "int8-mixed")])
def test_rgb48_end_to_end_vs_reference_int8_storage(torch_cuda, golden_dir, tag, prec):
    """The INT8 checkpoints as the reference executes them on ROCm (pre-dequantised at load)."""
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    d = np.load(os.path.join(golden_dir, f"int8_{tag}_64x96_gradient_s6.npz"))
    p = HDRTVNetMI355X(os.path.join(golden_dir, f"hr_int8_{tag}.hdrw"), precision=prec, use_hg=False, warmup_passes=0)
    try:
        out, _ = p.infer(p.preprocess(d["frame"]))
        got = _rgb48(p, out)
    finally:
        p.close()
    mx, mean = _lsb_histogram(f"int8 {tag} (pre-dequantised)", got, d["rgb48"])
    assert mx <= LSB_MAX and mean <= LSB_MEAN


@pytest.mark.parametrize("name", ["hr_64x96_noise_s0", "hr_60x100_noise_s2", "hr_52x76_gradient_s5"])
def test_rgb48_vs_the_oracle_with_f16_layer_io(torch_cuda, golden_dir, hr_state, name):
    """A second yardstick that separates "f16 storage" from "our kernels": the oracle re-run with every convolution's input,
    weights and output rounded to f16 (what an ideal f16-storage / fp32-accumulate execution of the reference's graph does by
    construction; tests/test_oracle_golden.py::test_w8a8_fp16_storage_sensitivity uses the same device).  Measured: that ideal
    execution is as far from the fp32 reference (mean 6-9 LSB) as the device is (7-9.5 LSB), and the two f16 executions are as
    far from EACH OTHER (9-12 LSB: independent rounding noise adds in quadrature) -- so an f16-I/O oracle is not a tighter
    reference for the integers, but it bounds what the kernels may add: the device must be no further from the reference than
    1.25 x the ideal f16 execution is."""
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    from oracle import hdrtvnet_oracle as O
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    try:
        out, _ = p.infer(p.preprocess(d["frame"]))
        got = _rgb48(p, out)
    finally:
        p.close()
    r16 = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)      # noqa: E731
    orig = O.conv2d
    O.conv2d = lambda x, w, b=None, stride=1, pad=0: r16(orig(r16(x), r16(w), b, stride, pad))
    try:
        rt, rc = O.preprocess(d["frame"])
        out16, _ = O.hr_forward(hr_state, r16(rt), r16(rc))
    finally:
        O.conv2d = orig
    want16 = O.post_rgb48(out16)
    mx, mean = _lsb_histogram(name + " vs the f16-I/O oracle", got, want16)
    mxr, meanr = _lsb_histogram(name + ": f16-I/O oracle vs the reference", want16, d["rgb48"])
    mxd, meand = _lsb_histogram(name + " vs the reference (fp32), again", got, d["rgb48"])
    assert meand <= 1.25 * meanr and mxd <= 1.5 * mxr + 16 and mean <= 1.6 * meanr


# ------------------------------------------------------------------------------------------ configs[0]: 960x540 on the device
def test_540p_on_the_device(torch_cuda, golden_dir, hr_state):
    """BASELINE.json configs[0]'s frame (960x540: the condition map is 135x240, LE's 1/8 level 68 rows -> 136 after the
    up-convs, cropped back to 135 by ``_align_to``) through the device: AGCM against the reference run's vectors, the
    whole HR forward against the oracle."""
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    from oracle import hdrtvnet_oracle as O
    d = np.load(os.path.join(golden_dir, "agcm_540x960_s0.npz"))
    frame = np.random.default_rng(0).integers(0, 256, (540, 960, 3), dtype=np.uint8)
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    try:
        t, c = p.preprocess(frame)
        out, agcm = p.infer((t, c))
        cond = c.float().cpu().numpy()[0]
        agcm_np, out_np = agcm.float().cpu().numpy()[0], out.float().cpu().numpy()[0]
        fea6 = p.tap("agcm.bias").numpy().ravel()[160:166]
        u8 = p.postprocess(out).copy()
    finally:
        p.close()
    mx, _ = _stats("cond (subsampled) vs reference", cond[:, ::9, ::16], d["cond_sub"])
    assert mx <= 1.2e-3
    mx, _ = _stats("fea6 vs reference", fea6, d["fea6"])
    assert mx <= 2e-3
    mx, _ = _stats("agcm_out (subsampled) vs reference", agcm_np[:, ::9, ::16], d["agcm_sub"])
    assert mx <= 2e-3
    assert np.abs(agcm_np.mean((1, 2)) - d["agcm_mean"]).max() <= 5e-4
    rt, rc = O.preprocess(frame)
    rout, ragcm = O.hr_forward(hr_state, rt, rc)
    mx, mean = _stats("out 540x960 vs O.hr_forward", out_np, rout)
    assert mx <= OUT_MAX and mean <= OUT_MEAN
    du8 = np.abs(u8.astype(int) - O.postprocess_u8(rout).astype(int))
    print(f"  u8: max={du8.max()} mean={du8.mean():.3f}")
    assert du8.max() <= 3 and du8.mean() <= 0.6


# ------------------------------------------------------------------------------------------ configs[2]: HG at 3840x2160
def test_uhd_hg_vs_oracle(torch_cuda, golden_dir, hg_state):
    """Full AGCM + LE + HG at 3840x2160.  The oracle (PyTorch's CPU kernels under the oracle's graph: ~40 s on the GPU
    box's cores) evaluates the HG generator once on the DEVICE's LE output padded to 2176 rows; ``hg.tail`` (conv_last's
    output before the blend) lets both masks be applied to that one run with the reference's arithmetic
    ``mask * out + img``.  Checked: conv2, conv5_2 (1/16 resolution: the 540-tile layers), conv9 (input of Up_conv5),
    the ps_dot3 partial sums over all padded pixels, and the final output at mask_r = 0.75 and 0.3."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    from oracle import hdrtvnet_oracle as O
    h, w = 2160, 3840
    f = W.synthetic_frame(h, w, seed=61, kind="gradient")
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    try:
        out, _ = p.infer(p.preprocess(f))
        out75 = out.cpu().numpy()[0]
        base = p.tap("le.out").numpy()
        dev = {n: p.tap(n).numpy() for n in ("hg.conv2", "hg.conv5_2", "hg.conv9")}
        part = p.tap("hg.part").numpy().reshape(-1).reshape(2176, w, 4)[:, :, :3].transpose(2, 0, 1)
        mask75_dev = p.tap("hg.mask").numpy()[:, :h, :w]
        p.set_hg_mask_r(0.3)
        out, _ = p.infer(p.preprocess(f))
        out30 = out.cpu().numpy()[0]
        mask30_dev = p.tap("hg.mask").numpy()[:, :h, :w]
        assert np.array_equal(p.tap("le.out").numpy(), base)
    finally:
        p.close()
    ph = (32 - h % 32) % 32
    assert ph == 16
    bp = np.pad(base, ((0, 0), (0, ph), (0, 0)), mode="reflect")
    taps = {}
    O.use_backend("aten")
    try:
        O.hg_generator(hg_state, bp, np.zeros((1, h + ph, w), np.float32), taps)
    finally:
        O.use_backend("c")
    # 1.5 x measured (3.3e-3 / 1.3e-4, 8.8e-3 / 3.8e-4, 2.1e-3 / 1.9e-4): the bars of test_gpu_parity.py
    for name, (tol, tol_mean) in (("hg.conv2", (4.9e-3, 2.2e-4)), ("hg.conv5_2", (1.32e-2, 7.0e-4)), ("hg.conv9", (3.1e-3, 3.2e-4))):
        mx, mean = _stats(name + " 2176x3840", dev[name], taps[name])
        assert mx <= tol and mean <= tol_mean, name
    w10 = np.asarray(hg_state["conv10.weight"], np.float32).reshape(3, 128)[:, :64]
    want_part = np.einsum("ok,khw->ohw", w10, taps["hg.up5"]).astype(np.float32)
    mx, mean = _stats("hg.part (ps_dot3 epilogue) 2176x3840", part, want_part)
    assert mx <= 1.5e-3 and mean <= 1.5e-4                   # measured 9.8e-4 / 9.8e-5
    tail = taps["hg.tail"][:, :h, :w]
    for r, got, dev_mask in ((0.75, out75, mask75_dev), (0.3, out30, mask30_dev)):
        mask = O.hg_mask(base, r=r)
        assert np.array_equal(dev_mask, mask)
        ref = (mask * tail + base).astype(np.float32)                       # Hallucination_arch.py:136
        mx, mean = _stats(f"hg out 2160x3840, mask_r={r} (mask fraction {mask.mean():.4f})", got, ref)
        assert mx <= 1.4e-3 and mean <= 1e-4                 # measured 9.4e-4 / 6.7e-5 (dense mask)
        if r == 0.3:
            assert mask.mean() >= 0.2
            inside = mask[0] > 0
            dd = np.abs(got - ref)[:, inside]
            print(f"  masked-in pixels only ({int(inside.sum())}): max_abs={dd.max():.3e} mean_abs={dd.mean():.3e}")
            assert dd.max() <= 1.4e-3


# ------------------------------------------------------------------------------------------ configs[4]: int8 HR + int8 HG
@pytest.mark.parametrize("size", [(272, 480), (1080, 1920)])
@pytest.mark.parametrize("tag", ["full", "mixed"])
def test_native_int8_hr_with_int8_hg_vs_fake_quant_oracle(torch_cuda, golden_dir, tag, size):
    """The configuration ``bench.py --int8`` / ``config4_int8`` times: the shipped QAT checkpoint with its W8A8 layers on
    int8 MFMA (``predequantize="off"``) feeding the W8A8 HG head (reference-style min/max calibration), at 272x480 and at
    1920x1080 (thousands of tiles per int8 kernel, all 16 border classes of the zero-point constants; ~25 s of ATen), against
    the oracle's fake-quant composite (fp32, ATen convolutions: the arithmetic of the reference's CPU run).  Bars: the
    reference's own for a re-quantised graph (float MAE <= 0.02, u8 MAE <= 5, scripts/validate_tensorrt_sources.py:598-609)
    end to end; the HG head alone, on the device's own LE output, the bars of test_gpu_int8_hg.py."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    from oracle import hdrtvnet_oracle as O
    h, w = size
    f = W.synthetic_frame(h, w, seed=11, kind="gradient")
    qstate = W.seeded_hg_w8a8_state(1234, integer_zero=False)
    p = HDRTVNetMI355X(os.path.join(golden_dir, f"hr_int8_{tag}_qat.hdrw"), precision=f"int8-{tag}", predequantize="off",
                       use_hg=True, hg_weights="seeded-w8a8-minmax:1234", warmup_passes=0)
    try:
        assert p._is_w8_model and p._hg_int8
        out, agcm = p.infer(p.preprocess(f))
        out_np = out.cpu().numpy()[0]
        base = p.tap("le.out").numpy()
        u8 = p.postprocess(out).copy()
    finally:
        p.close()
    sd = O.w8a8_state(W.load_pack(os.path.join(golden_dir, f"hr_int8_{tag}_qat.hdrw")))
    hq = O.w8a8_state(qstate)
    O.use_backend("aten")
    try:
        rt, rc = O.preprocess(f)
        taps = {}
        ref, _ = O.hg_composite(sd, hq, rt, rc, taps)
        ph = (32 - h % 32) % 32
        mask = O.hg_mask(base)
        ref_on_base = O.hg_generator(hq, np.pad(base, ((0, 0), (0, ph), (0, 0)), mode="reflect"),
                                     np.pad(mask, ((0, 0), (0, ph), (0, 0)), mode="reflect"))[:, :h, :w]
    finally:
        O.use_backend("c")
    mx, mean = _stats(f"int8-{tag} LE out vs fake-quant oracle", base, taps["base"])
    mx, mean = _stats(f"int8-{tag} HR + int8 HG, final out vs fake-quant composite", out_np, ref)
    du8 = np.abs(u8.astype(int) - O.postprocess_u8(ref).astype(int))
    flips = float((O.hg_mask(base) != taps["mask"]).mean())
    print(f"  u8: max={du8.max()} MAE={du8.mean():.4f} (reference bar: MAE <= 5); mask flips {flips:.4%}")
    assert mean <= 0.02 and du8.mean() <= 5.0
    assert mean <= (1.5e-2 if tag == "full" else 3e-3)              # ~2-3x the measured level (printed above)
    e = np.abs(out_np - ref_on_base)
    print(f"  HG head on the device's own LE output: max {e.max():.3e} mean {e.mean():.3e}")
    assert e.max() <= 2e-2 and e.mean() <= 5e-4


def test_native_int8_default_path_at_3840x2160(torch_cuda, golden_dir, monkeypatch):
    """configs[4] AT ITS OWN SIZE, on the kernels ``bench.py``'s ``config4_int8`` times: the shipped full-QAT checkpoint with
    ``predequantize="off"`` and every variant at its default, feeding the W8A8 HG head, on one 3840x2160 frame -- 64 strips x
    4 segments of the fused LE row kernels on int8 MFMA (le_rows_i8.hip: the steady state of their code rings), 8160-tile int8
    HG layers.
      * against the oracle's fake-quant composite (fp32, ATen convolutions = the reference's CPU arithmetic; ~2 min of
        CPU): the reference's own bars for a re-quantised graph, float MAE <= 0.02 and u8 MAE <= 5
        (scripts/validate_tensorrt_sources.py:598-609), and ~2x the level this build measures;
      * schedule invariance, bit for bit: on 128 instead of 256 workgroups (variant force_ncu = 128: 64 strips x 2 segments,
        every other persistent kernel walks twice the tiles per workgroup), and against one tile per workgroup (force_ncu =
        4000000: no map is then large enough for a row kernel, so every layer runs on the per-layer int8 kernels -- the schedule
        the small goldens validate), as test_persistent_schedules_do_not_change_results does for fp16.  Since round 5 the fused
        kernels compute the per-layer kernels' bits, so which of the two a layer runs on no longer shows in the output;
      * every W8A8 layer outside the AGCM classifier ran on int8 MFMA (`execution_summary()` of the launch profile)."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X, summarize_profile
    from oracle import hdrtvnet_oracle as O
    torch = torch_cuda
    h, w = 2160, 3840
    f = W.synthetic_frame(h, w, seed=12, kind="gradient")
    path = os.path.join(golden_dir, "hr_int8_full_qat.hdrw")
    qstate = W.seeded_hg_w8a8_state(1234, integer_zero=False)
    monkeypatch.delenv("HDRTV_VARIANTS", raising=False)
    p = HDRTVNetMI355X(path, precision="int8-full", predequantize="off", use_hg=True, hg_weights="seeded-w8a8-minmax:1234", warmup_passes=0)
    runs = {}
    try:
        assert p._is_w8_model and p._hg_int8
        assert p.get_variant("le_rows") == 1 and p.get_variant("le_rows_i8") == 1 and p.get_variant("force_ncu") == 0     # the defaults
        for name, ncu in (("default", 0), ("128 workgroups", 128), ("one tile per workgroup", 4000000)):
            p.set_variant("force_ncu", ncu)
            p.profile_enable(True)
            out, agcm = p.infer(p.preprocess(f))
            prof = p.profile_read()
            p.profile_enable(False)
            runs[name] = (out.clone(), agcm.clone(), p.tap("le.out").clone(), p.tap("le.fea0").clone(), p.postprocess(out).copy(), prof)
    finally:
        p.close()
    kern = [k for _, k, *_ in runs["default"][5]]
    ran = summarize_profile(runs["default"][5])
    print(f"  default run: {ran['text']}")
    assert {"le_head_rows<i8>", "le_rb_rows<i8>", "le_tail_rows<i8>"} <= set(kern), sorted(set(kern))
    assert not any("rows" in k for _, k, *_ in runs["one tile per workgroup"][5])
    assert ran["fq-f16"]["launches"] == 0                                        # no W8A8 layer as fake-quant on fp16 MFMA
    assert ran["int8"]["gmac"] >= 0.99 * sum(v["gmac"] for k, v in ran.items() if k != "text")
    for other in ("128 workgroups", "one tile per workgroup"):
        for name, a, b in zip(("out", "agcm", "le.out", "le.fea0"), runs["default"][:4], runs[other][:4]):
            assert torch.isfinite(a).all(), name
            assert torch.equal(a, b), (other, name, int((a != b).sum()))
    sd = O.w8a8_state(W.load_pack(path))
    hq = O.w8a8_state(qstate)
    O.set_threads(min(16, os.cpu_count() or 1))
    O.use_backend("aten")
    try:
        taps = {}
        ref, _ = O.hg_composite(sd, hq, *O.preprocess(f), taps)
    finally:
        O.use_backend("c")
    out_np = runs["default"][0].float().cpu().numpy()[0]
    base = runs["default"][2].numpy()
    _stats("int8-full LE out 2160x3840 vs fake-quant oracle", base, taps["base"])
    mx, mean = _stats("int8-full HR + int8 HG 2160x3840, final out vs fake-quant composite", out_np, ref)
    du8 = np.abs(runs["default"][4].astype(int) - O.postprocess_u8(ref).astype(int))
    flips = float((O.hg_mask(base) != taps["mask"]).mean())
    print(f"  u8: max={du8.max()} MAE={du8.mean():.4f} (reference bar: MAE <= 5); mask flips {flips:.4%}")
    assert mean <= 0.02 and du8.mean() <= 5.0
    assert mean <= 1.5e-2
