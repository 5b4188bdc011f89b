"""``precision="fp32"`` (HDRTVNetTorch's maximum-precision preset, hdrtvnet_torch.py:1694-1712) on the device: the fp32 graph
(csrc/fp32_ops.hip + fp32_graph.hip) against what the reference's OWN fp32 run produced (tests/golden/*.npz: CPU fp32).

Both sides compute every product and sum in fp32; they differ only in summation order (ATen / oneDNN blocking vs one FMA
chain per output here).  Measured: every LE tensor within 3.7e-6, every HG tensor within 2.2e-5 (conv_code2, sums over 9216
products), final outputs within 1.3e-6; RGB48 integers (the contract of gui_pipeline_worker_feeders.py:223-227) identical to
the reference's on 99.7-99.8 % of the values and 1 LSB off on the rest (floats that land on a rounding boundary).  Bars: 1.5x.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_HR, TOL_HG = 6e-6, 3.5e-5
EQ48 = 0.995            # fraction of RGB48 integers that must equal the reference's; the rest within 1 LSB


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    return torch


@pytest.fixture(scope="module")
def p32_hr(torch_cuda, golden_dir):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), precision="fp32", use_hg=False, warmup_passes=0)
    yield p
    p.close()


@pytest.fixture(scope="module")
def p32_hg(torch_cuda, golden_dir):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), precision="fp32", use_hg=True, hg_weights="seeded:1234",
                       warmup_passes=0)
    yield p
    p.close()


def _mx(name, got, want):
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    print(f"  {name}: max_abs={d.max():.3e} mean_abs={d.mean():.3e}")
    return d.max()


def _rgb48(p, out, torch):
    from hdrtv_mi355x import lib as L
    import ctypes as C
    h, w = out.shape[-2:]
    dst = torch.empty((h, w, 3), dtype=torch.int16, device=out.device)
    p._chk(p._lib.hdrtv_post_rgb48(p._ctx, p._stream(), out.data_ptr(), L.F32, h, w, dst.data_ptr()), "hdrtv_post_rgb48")
    torch.cuda.synchronize()
    return dst.cpu().numpy().view(np.uint16)


OUR_TAP = {"LE.cond_first": "le32.cond", "LE.CondNet1": "le32.cond1", "LE.CondNet2": "le32.cond2", "LE.CondNet3": "le32.cond3",
           "LE.CondNet4": "le32.cond4", "LE.SFT_layer1": "le32.s1", "LE.down_conv1": None, "LE.recon_trunk1": "le32.fea1",
           "LE.recon_trunk2": "le32.fea2", "LE.recon_trunk3": "le32.t3b", "LE.recon_trunk4": "le32.o2", "LE.recon_trunk5": "le32.o1",
           "LE.SFT_layer2": "le32.s2s", "LE.conv_last": "le32.last"}


@pytest.mark.parametrize("name", ["hr_64x96_noise_s0", "hr_60x100_noise_s2", "hr_52x76_gradient_s5", "hr_32x96_gradient_s1_taps"])
def test_fp32_hr_vs_the_reference(p32_hr, golden_dir, torch_cuda, name):
    """AGCM + LE (aligned, ragged -> _align_to, gradient) against the reference's tensors, stage by stage."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    t, c = p32_hr.preprocess(d["frame"])
    assert t.dtype == torch_cuda.float32 and c.dtype == torch_cuda.float32
    assert np.array_equal(t.cpu().numpy()[0], d["tensor"])                       # u8 * fp32(1/255): exact
    assert _mx("cond", c.cpu().numpy()[0], d["cond"]) <= 1e-6
    out, agcm = p32_hr.infer((t, c))
    assert out.dtype == torch_cuda.float32 and agcm.dtype == torch_cuda.float32
    if "fea6" in d.files:
        assert _mx("fea6", p32_hr.tap("agcm32.fea6").numpy().ravel(), d["fea6"]) <= 1e-6
    assert _mx("agcm_out", agcm.cpu().numpy()[0], d["agcm_out"]) <= TOL_HR
    for k in d.files:
        if k.startswith("tap:") and OUR_TAP.get(k[4:]):
            got, want = p32_hr.tap(OUR_TAP[k[4:]]).numpy(), d[k]
            if got.shape != want.shape:
                got = got[::got.shape[0] // want.shape[0]]
            assert _mx(k, got, want) <= TOL_HR, k
    assert _mx("out", out.cpu().numpy()[0], d["out"]) <= TOL_HR
    got48 = _rgb48(p32_hr, out, torch_cuda)
    diff = np.abs(got48.astype(int) - d["rgb48"].astype(int))
    print(f"  rgb48: equal {100.0 * (diff == 0).mean():.3f} %  max {diff.max()} LSB")
    assert (diff == 0).mean() >= EQ48 and diff.max() <= 1
    u8 = p32_hr.postprocess((out, agcm)).astype(int)
    assert np.abs(u8 - d["u8_bgr"].astype(int)).max() <= 1 and (u8 == d["u8_bgr"]).mean() >= 0.999


@pytest.mark.parametrize("name", ["hg_96x128_gradient_s3", "hg_80x112_gradient_s4"])
def test_fp32_hg_vs_the_reference(p32_hg, golden_dir, torch_cuda, name):
    """The whole composite (80x112: reflect padding to 96x128) against the reference's run with the same seeded HG head."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    t, c = p32_hg.preprocess(d["frame"])
    out, agcm = p32_hg.infer((t, c))
    assert _mx("base", p32_hg.tap("le32.out").numpy(), d["tap:base"]) <= TOL_HR
    assert np.array_equal(p32_hg.tap("hg32.mask").numpy(), d["mask"])
    for k in d.files:
        if k.startswith("tap:hg."):
            got, want = p32_hg.tap("hg32." + k[7:]).numpy(), d[k]
            if got.shape != want.shape:
                got = got[::got.shape[0] // want.shape[0]]
            assert _mx(k, got, want) <= TOL_HG, k
    assert _mx("out", out.cpu().numpy()[0], d["out"]) <= TOL_HG
    diff = np.abs(_rgb48(p32_hg, out, torch_cuda).astype(int) - d["rgb48"].astype(int))
    print(f"  rgb48: equal {100.0 * (diff == 0).mean():.3f} %  max {diff.max()} LSB")
    assert (diff == 0).mean() >= EQ48 and diff.max() <= 1


def test_fp32_condition_map_shortcuts(torch_cuda, golden_dir):
    """fast_condition_resize (bilinear) and HDRTVNET_ZERO_COND with fp32 tensors (hdrtvnet_torch.py:2262-2276)."""
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    d = np.load(os.path.join(golden_dir, "cond_modes_61x103_gradient_s7.npz"))
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), precision="fp32", use_hg=False, warmup_passes=0,
                       fast_condition_resize=True)
    try:
        t, c = p.preprocess(d["frame"])
        assert _mx("cond (bilinear)", c.cpu().numpy()[0], d["cond_bilinear"]) <= 1e-6
        out, agcm = p.infer((t, c))
        assert _mx("agcm (bilinear)", agcm.cpu().numpy()[0], d["agcm_bilinear"]) <= TOL_HR
        assert _mx("out (bilinear)", out.cpu().numpy()[0], d["out_bilinear"]) <= TOL_HR
        p._chk(p._lib.hdrtv_set_cond_mode(p._ctx, 2), "hdrtv_set_cond_mode")
        t, c = p.preprocess(d["frame"])
        assert float(c.abs().max()) == 0.0
        out, agcm = p.infer((t, c))
        assert _mx("out (zero cond)", out.cpu().numpy()[0], d["out_zero"]) <= TOL_HR
    finally:
        p.close()


@pytest.mark.parametrize("fixture", ["full_1080x1920_hg_s11.npz", "full_2160x3840_hg_s12.npz"])
def test_fp32_at_the_baseline_sizes_vs_the_reference(p32_hg, golden_dir, torch_cuda, fixture):
    """BASELINE configs[1]'s and configs[2]'s sizes: the fixtures tests/golden/gen_golden_fullsize.py took from HDRTVNetTorch
    (fp32; at 1920x1080 the aligned fast graph) -- strided samples, dense patches and the sums of all RGB48 integers."""
    import time
    from hdrtv_mi355x import weights as W
    d = np.load(os.path.join(golden_dir, fixture))
    h, w = (int(v) for v in d["shape"])
    frame = W.synthetic_frame(h, w, seed=int(d["seed"]), kind=str(d["kind"]))
    t, c = p32_hg.preprocess(frame)
    out, agcm = p32_hg.infer((t, c))
    torch_cuda.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out, agcm = p32_hg.infer((t, c))
    torch_cuda.cuda.synchronize()
    print(f"  fp32 infer at {w}x{h}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per frame")
    rs, cs = (int(v) for v in d["stride"])
    o = out.cpu().numpy()[0]
    assert _mx("agcm_out", agcm.cpu().numpy()[0][:, ::rs, ::cs], d["agcm_out"]) <= TOL_HR
    base = p32_hg.tap("le32.out").numpy()
    assert _mx("base", base[:, ::rs, ::cs], d["base"]) <= TOL_HR
    mask = p32_hg.tap("hg32.mask").numpy()
    flips = int((mask[:, ::rs, ::cs] != d["mask"]).sum())
    print(f"  mask flips on the sample grid: {flips}; mask pixels {int(mask.sum())} vs {int(d['mask_count'])}")
    assert flips == 0 and abs(int(mask.sum()) - int(d["mask_count"])) <= 4
    assert _mx("out", o[:, ::rs, ::cs], d["out"]) <= TOL_HG
    assert _mx("out corner", o[:, :32, :48], d["out_corner"]) <= TOL_HG
    assert _mx("out centre", o[:, h // 2 - 16:h // 2 + 16, w // 2 - 24:w // 2 + 24], d["out_centre"]) <= TOL_HG
    got48 = _rgb48(p32_hg, out, torch_cuda)
    diff = np.abs(got48[::rs, ::cs].astype(int) - d["rgb48"].astype(int))
    print(f"  rgb48 samples: equal {100.0 * (diff == 0).mean():.3f} %  max {diff.max()} LSB")
    assert (diff == 0).mean() >= EQ48 and diff.max() <= 1
    sums = np.array([int(got48[..., ch].astype(np.int64).sum()) for ch in range(3)])
    rel = np.abs(sums - d["rgb48_sum"]) / d["rgb48_sum"]
    print(f"  rgb48 channel sums: relative difference {rel}")
    assert rel.max() <= 1e-6


def test_fp32_refuses_an_int8_checkpoint(torch_cuda, golden_dir):
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    with pytest.raises(ValueError):
        HDRTVNetMI355X(os.path.join(golden_dir, "hr_int8_full_qat.hdrw"), precision="fp32", use_hg=False, warmup_passes=0)


def test_fp32_through_hip_graph_replay_and_the_dispatcher(torch_cuda, golden_dir):
    """The fp32 graph (186 launches + one D2D copy) captured into a hipGraph replays bit for bit, and a dispatcher worker built
    with precision="fp32" hands back the RGB48 frames an in-process fp32 processor produces."""
    import ctypes as C
    from hdrtv_mi355x import lib as L, weights as W
    from hdrtv_mi355x.dispatch import FrameDispatcher
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    h, w = 272, 480
    frames = [W.synthetic_frame(h, w, seed=70 + i, kind="gradient" if i % 2 else "noise") for i in range(4)]
    path = os.path.join(golden_dir, "hr_weights.hdrw")
    outs, want = [], []
    for graphs in (False, True):
        p = HDRTVNetMI355X(path, precision="fp32", use_hg=True, hg_weights="seeded:1234", warmup_passes=0, use_cuda_graphs=graphs)
        try:
            t, c = p.preprocess(frames[1])
            for _ in range(3):
                out, agcm = p.infer((t, c))
            torch_cuda.cuda.synchronize()
            outs.append((out.clone(), agcm.clone()))
            if not graphs:
                for f in frames:
                    o, _ = p.infer(p.preprocess(f))
                    want.append(_rgb48(p, o, torch_cuda).copy())
        finally:
            p.close()
    assert torch_cuda.equal(outs[0][0], outs[1][0]) and torch_cuda.equal(outs[0][1], outs[1][1])
    got = {}
    args = {"model_path": path, "precision": "fp32", "use_hg": True, "hg_weights": "seeded:1234"}
    with FrameDispatcher(1, h, w, lambda i, v: got.__setitem__(i, v.copy()), init_args=args, devices=[0], slots=2) as d:
        for f in frames:
            d.submit(f)
        d.flush(timeout=120)
    assert d.exit_codes == [0] and sorted(got) == list(range(len(frames)))
    for i in range(len(frames)):
        assert np.array_equal(got[i], want[i]), i


def test_create_ex_argument_checks(torch_cuda, golden_dir):
    """hdrtv_create_ex: a precision it does not know is EINVAL, an INT8 pack with HDRTV_PREC_F32 is EWEIGHTS, and an fp32 context
    refuses f16 outputs -- each with a message behind hdrtv_last_error."""
    import ctypes as C
    from hdrtv_mi355x import lib as L
    lib = L.load()
    blob = open(os.path.join(golden_dir, "hr_weights.hdrw"), "rb").read()
    ctx = C.c_void_p()
    assert lib.hdrtv_create_ex(blob, len(blob), None, 0, 0, 7, C.byref(ctx)) == L.EINVAL
    assert b"precision" in lib.hdrtv_last_error(ctx)
    lib.hdrtv_destroy(ctx)
    q = open(os.path.join(golden_dir, "hr_int8_full_qat.hdrw"), "rb").read()
    ctx = C.c_void_p()
    assert lib.hdrtv_create_ex(q, len(q), None, 0, 0, L.PREC_F32, C.byref(ctx)) == L.EWEIGHTS
    assert b"INT8" in lib.hdrtv_last_error(ctx)
    lib.hdrtv_destroy(ctx)
    ctx = C.c_void_p()
    assert lib.hdrtv_create_ex(blob, len(blob), None, 0, 0, L.PREC_F32, C.byref(ctx)) == 0
    assert lib.hdrtv_reserve(ctx, 64, 96) == 0
    x = torch_cuda.zeros((3, 64, 96), dtype=torch_cuda.float32, device="cuda")
    cnd = torch_cuda.zeros((3, 16, 24), dtype=torch_cuda.float32, device="cuda")
    out = torch_cuda.zeros((3, 64, 96), dtype=torch_cuda.float32, device="cuda")
    st = C.c_void_p(torch_cuda.cuda.current_stream().cuda_stream)
    assert lib.hdrtv_infer(ctx, st, x.data_ptr(), cnd.data_ptr(), 64, 96, out.data_ptr(), L.F16, None) == L.EINVAL
    assert lib.hdrtv_infer(ctx, st, x.data_ptr(), cnd.data_ptr(), 64, 96, out.data_ptr(), L.F32, None) == 0
    torch_cuda.cuda.synchronize()
    assert bool(torch_cuda.isfinite(out).all())
    lib.hdrtv_destroy(ctx)


def test_fp32_matrix_pipe_convs_against_the_vector_kernel(torch_cuda, golden_dir):
    """conv_f32_mfma (3x3 / stride-1 layers on v_mfma_f32_32x32x2_f32) against conv_f32 (one fmaf chain per output element on the
    vector ALUs; variant f32_mfma = 0): the same products, the same order of accumulation (input channel, tap) from zero, the bias
    behind the sum -- an fp32 MFMA is itself an fmaf chain (MI355X_MICROARCH.md), so the two graphs must agree to the last bit
    or, should the matrix pipe associate its two products per instruction the other way round, to an ulp or two per layer.
    Printed: the fraction of identical output values; asserted: <= 2e-6 everywhere, the profile names the kernels, and the
    layers that must stay on the vector kernel (1x1, stride 2, 3-channel ends, maps too small for the chip) do."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    torch = torch_cuda
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), precision="fp32", use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    try:
        assert p.get_variant("f32_mfma") == 1
        for (h, w), seed in (((272, 480), 3), ((1080, 1920), 4), ((61, 103), 5)):
            f = W.synthetic_frame(h, w, seed=seed, kind="gradient")
            res, kern = [], []
            for v in (0, 1):
                p.set_variant("f32_mfma", v)
                p.profile_enable(True)
                out, agcm = p.infer(p.preprocess(f))
                prof = p.profile_read()
                p.profile_enable(False)
                kern.append(prof)
                res.append((out.clone(), agcm.clone()))
            assert not any(k == "conv_f32_mfma" for _, k, *_ in kern[0])
            on = [(l, m) for l, k, _, m, _ in kern[1] if k == "conv_f32_mfma"]
            off = [(l, m) for l, k, _, m, _ in kern[1] if k == "conv_f32"]
            share = sum(m for _, m in on) / max(1.0, sum(m for _, m in on + off))
            d = (res[0][0] - res[1][0]).abs()
            same = float((d == 0).float().mean())
            print(f"  {w}x{h}: {len(on)} layers ({share:.1%} of the conv MACs) on the matrix pipe, {len(off)} on the vector kernel; "
                  f"out identical on {same:.3%} of the values, max |delta| {float(d.max()):.2e}")
            assert float(d.max()) <= 2e-6
            if h >= 1080:
                assert share >= 0.8              # 3x3 layers with >= 64 input channels + the 64 -> 64 / concat 1x1 layers
            assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-6
    finally:
        p.close()
