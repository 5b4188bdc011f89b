"""W8A8 HG head on int8 MFMA (BASELINE.json configs[4]) against the oracle's fake-quant restatement, which
tests/test_oracle_golden.py::test_hg_w8a8_fake_quant_execution pins bit-exact to the reference's own W8A8Conv2d.

The device computes the quantised layers in exact integer arithmetic, the oracle (like the reference) as fp32
convolutions of dequantised values; both then round to the next layer's 8-bit codes.  Values that land within float
noise of a rounding boundary come out one code apart, and such a flip propagates as a (small) real difference, so the
bars are statistical: codes equal or adjacent almost everywhere, final output within the tolerances below.  The
reference's own bar for a re-quantised graph is float MAE <= 0.02 / u8 MAE <= 5
(scripts/validate_tensorrt_sources.py:598-609)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# two calibrations of the same seeded head: integer zero points (x_zero = -k x_scale: padding is an exact code, the whole
# layer is integer arithmetic) and the reference's own calibrate_w8a8 rule (x_zero = running minimum, a float: padded taps
# are code 128 and the border pixels get a per-class constant, conv3x3_pglds_i8.hip shift_of)
@pytest.fixture(scope="module", params=["integer-zero", "minmax"])
def zero_style(request):
    return request.param


@pytest.fixture(scope="module")
def proc_q(golden_dir, zero_style):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True,
                       hg_weights="seeded-w8a8:1234" if zero_style == "integer-zero" else "seeded-w8a8-minmax:1234", warmup_passes=0)
    yield p
    p.close()


@pytest.fixture(scope="module")
def qstate(zero_style):
    from hdrtv_mi355x import weights as W
    return W.seeded_hg_w8a8_state(1234, integer_zero=zero_style == "integer-zero")


def _oracle_on_base(qstate, base):
    from oracle import hdrtvnet_oracle as O
    q = O.w8a8_state(qstate)
    mask = O.hg_mask(base)
    h, w = base.shape[1:]
    ph, pw = (32 - h % 32) % 32, (32 - w % 32) % 32
    taps = {}
    ref = O.hg_generator(q, np.pad(base, ((0, 0), (0, ph), (0, pw)), mode="reflect"),
                         np.pad(mask, ((0, 0), (0, ph), (0, pw)), mode="reflect"), taps)[:, :h, :w]
    return ref, taps, q, mask


# device int8 tensor -> (oracle tap holding the same tensor before quantisation, a layer that reads it)
CODE_TAPS = (("hg8.p1", "hg.p1", "conv2.0"), ("hg8.conv2", "hg.conv2", "conv3_1.0"), ("hg8.conv3_2", "hg.conv3_2", "conv4_1.0"),
             ("hg8.conv4_2", "hg.conv4_2", "conv5_1.0"), ("hg8.conv5_2", "hg.conv5_2", "conv_code1.0"),
             ("hg8.conv_code2", "hg.conv_code2", "Up_conv1.0"), ("hg8.conv6", "hg.conv6", "Up_conv2.0"),
             ("hg8.conv7", "hg.conv7", "Up_conv3.0"), ("hg8.conv8", "hg.conv8", "Up_conv4.0"),
             ("hg8.conv9", "hg.conv9", "Up_conv5.0"))


@pytest.mark.parametrize("hw,seed", [((96, 128), 3), ((80, 112), 4), ((272, 480), 11)])
def test_w8a8_hg_vs_oracle(proc_q, qstate, hw, seed):
    from hdrtv_mi355x import weights as W
    f = W.synthetic_frame(*hw, seed=seed, kind="gradient")
    out, _ = proc_q.infer(proc_q.preprocess(f))
    out = out.cpu().numpy()[0]
    base = proc_q.tap("le.out").numpy()
    ref, taps, q, mask = _oracle_on_base(qstate, base)
    h, w = hw
    for dev_name, ora_name, reader in CODE_TAPS:
        s, z = float(qstate[reader + ".x_scale"]), float(qstate[reader + ".x_zero"])
        k = round(-z / s)
        codes = proc_q.tap(dev_name).numpy() + 128.0                       # the reference's u8 code
        want = np.clip(np.rint((taps[ora_name] - np.float32(z)) / np.float32(s)), 0, 255)
        d = np.abs(codes - want)
        print(f"  {dev_name}: codes differ at {np.mean(d > 0):.4%}, by more than one at {np.mean(d > 1):.4%}, max {d.max():.0f} "
              f"(k={k}, used range {want.min():.0f}..{want.max():.0f})")
        if dev_name == "hg8.p1":             # the fp16 -> int8 boundary: only conv1's fp16 rounding separates the two
            assert np.mean(d > 0) <= 0.03 and d.max() <= 1
    e = np.abs(out - ref)
    print(f"  {hw} out vs oracle (same base): max {e.max():.3e} mean {e.mean():.3e}; mask fraction {mask.mean():.4f}")
    assert e.max() <= 2e-2 and e.mean() <= 5e-4


# (layer, input code tensors, output tensor, store): each layer alone, fed the DEVICE's own input codes
LAYERS = (("conv2", ("hg8.p1",), "hg8.conv2", "block"), ("conv3_1", ("hg8.conv2",), "hg8.p3", "pool"), ("conv3_2", ("hg8.p3",), "hg8.conv3_2", "block"),
          ("conv4_1", ("hg8.conv3_2",), "hg8.p4", "pool"), ("conv4_2", ("hg8.p4",), "hg8.conv4_2", "block"),
          ("conv5_1", ("hg8.conv4_2",), "hg8.p5", "pool"), ("conv5_2", ("hg8.p5",), "hg8.conv5_2", "block"),
          ("conv_code1", ("hg8.conv5_2",), "hg8.pc", "pool"), ("conv_code2", ("hg8.pc",), "hg8.conv_code2", "block"),
          ("Up_conv1", ("hg8.conv_code2",), "hg8.up1", "up"), ("conv6", ("hg8.up1", "hg8.conv5_2"), "hg8.conv6", "fuse"),
          ("Up_conv2", ("hg8.conv6",), "hg8.up2", "up"), ("conv7", ("hg8.up2", "hg8.conv4_2"), "hg8.conv7", "fuse"),
          ("Up_conv3", ("hg8.conv7",), "hg8.up3", "up"), ("conv8", ("hg8.up3", "hg8.conv3_2"), "hg8.conv8", "fuse"),
          ("Up_conv4", ("hg8.conv8",), "hg8.up4", "up"), ("conv9", ("hg8.up4", "hg8.conv2"), "hg8.conv9", "fuse"), ("Up_conv5", ("hg8.conv9",), "hg.part", "dot3"))
READER = {"hg8.p3": "conv3_2.0", "hg8.conv3_2": "conv4_1.0", "hg8.p4": "conv4_2.0", "hg8.conv4_2": "conv5_1.0",
          "hg8.p5": "conv5_2.0", "hg8.conv5_2": "conv_code1.0", "hg8.pc": "conv_code2.0", "hg8.conv_code2": "Up_conv1.0",
          "hg8.up1": "conv6", "hg8.conv6": "Up_conv2.0", "hg8.up2": "conv7", "hg8.conv7": "Up_conv3.0", "hg8.up3": "conv8",
          "hg8.conv8": "Up_conv4.0", "hg8.conv2": "conv3_1.0", "hg8.up4": "conv9", "hg8.p1": "conv2.0", "hg8.conv9": "Up_conv5.0"}


@pytest.mark.parametrize("hw,seed", [((96, 128), 3), ((272, 480), 11), ((480, 854), 99)])
def test_w8a8_layers_exact_given_device_inputs(proc_q, qstate, hw, seed):
    """Each int8 layer in isolation: the oracle's layer (fp32 convolution of the dequantised DEVICE input codes, then
    BatchNorm / ReLU / pool / pixel shuffle and the output quantiser) against the device's output codes.  The integer
    kernel is exact, the fp32 convolution is not, so values within ~1e-6 relative of a rounding boundary may differ by
    one code; nothing else may."""
    from hdrtv_mi355x import weights as W
    from oracle import hdrtvnet_oracle as O
    q = O.w8a8_state(qstate)
    f = W.synthetic_frame(*hw, seed=seed, kind="gradient")
    proc_q.infer(proc_q.preprocess(f))

    def qp(reader):
        s, z = np.float32(qstate[reader + ".x_scale"]), np.float32(qstate[reader + ".x_zero"])
        return s, z

    for name, srcs, dst, kind in LAYERS:
        xs = []
        for t in srcs:
            s, z = qp(READER[t])
            xs.append(((proc_q.tap(t).numpy() + np.float32(128.0)) * s + z).astype(np.float32))
        x = np.concatenate(xs, axis=0)
        if kind == "fuse":
            y = O.conv2d(x, q[name + ".weight"], q[name + ".bias"])
        elif kind == "dot3":
            # Up_conv5 + pixel shuffle + ReLU, then the first 64 input channels of conv10 (the fused epilogue): the device
            # keeps only these three partial sums per pixel, f32 [H][W][4]
            u5 = O._hg_up(q, name, x).astype(np.float16).astype(np.float32)
            w10 = np.asarray(qstate["conv10.weight"], np.float32).reshape(3, 128)[:, :64]
            y = np.tensordot(w10, u5, axes=(1, 0))
            hp, wp = y.shape[1:]
            got = proc_q.tap(dst).numpy().reshape(hp, wp, 4)[:, :, :3].transpose(2, 0, 1)
            d = np.abs(got - y)
            print(f"  {name:10s} -> {dst} (f32 partial sums): max {d.max():.3e} mean {d.mean():.3e} |ref| {np.abs(y).mean():.3e}")
            assert d.max() <= 2e-3
            continue
        elif kind == "up":
            y = O._hg_up(q, name, x)
        else:
            y = O._hg_block(q, name, x)
            if kind == "pool":
                y = O.maxpool2(y)
        got = proc_q.tap(dst).numpy()
        if dst.startswith("hg8."):
            s, z = qp(READER[dst])
            want = np.clip(np.rint((y - z) / s), 0, 255)
            d = np.abs(got + 128.0 - want)
            print(f"  {name:10s} -> {dst}: {int((d > 0).sum())} of {d.size} codes differ (max {d.max():.0f})")
            assert d.max() <= 1 and np.mean(d > 0) <= 1e-3, name
        else:
            d = np.abs(got - y)
            print(f"  {name:10s} -> {dst} (f16): max {d.max():.3e} mean {d.mean():.3e}")
            assert d.max() <= 4e-3 * max(1.0, float(np.abs(y).max())), name


def test_w8a8_hg_vs_reference_golden(proc_q, golden_dir, zero_style):
    """Against the reference's own run (W8A8Conv2d swapped into its HG_Composite; tests/golden/gen_golden_hg_w8a8.py)."""
    d = np.load(os.path.join(golden_dir, "hg_w8a8_96x128_gradient_s3.npz" if zero_style == "integer-zero"
                             else "hg_w8a8_floatzero_96x128_gradient_s3.npz"))
    out, _ = proc_q.infer(proc_q.preprocess(d["frame"]))
    out = out.cpu().numpy()[0]
    e = np.abs(out - d["out"])
    u8 = proc_q.postprocess((proc_q.infer(proc_q.preprocess(d["frame"])))).astype(int)
    mae = np.abs(u8 - d["u8_bgr"].astype(int)).mean()
    print(f"  out vs reference golden: max {e.max():.3e} mean {e.mean():.3e}; u8 MAE {mae:.4f}")
    assert e.max() <= 3e-2 and e.mean() <= 5e-4 and mae <= 0.2


def test_w8a8_schedules_do_not_change_results(golden_dir):
    """The persistent int8 kernels at 3840x2160: the real tile schedule against one tile per workgroup, bit for bit."""
    import torch
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    f = W.synthetic_frame(2160, 3840, seed=5, kind="gradient")
    outs = []
    for force in (None, "4000000"):
        if force:
            os.environ["HDRTV_FORCE_NCU"] = force
        try:
            p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True, hg_weights="seeded-w8a8:1234",
                               warmup_passes=0)
        finally:
            os.environ.pop("HDRTV_FORCE_NCU", None)
        o, _ = p.infer(p.preprocess(f))
        outs.append((o.clone(), p.tap("hg8.conv8").clone(), p.tap("hg8.p3").clone()))
        p.close()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_calibrate_and_requantise_on_device(golden_dir):
    """calibrate_hg_w8a8 (the reference's calibrate_w8a8, method "max") on the fp16 model's own activations, then the
    resulting W8A8 checkpoint loaded back: ranges agree with the shipped table (same weights, similar frames) and the
    int8 model stays close to the fp16 one."""
    import json
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    hr = os.path.join(golden_dir, "hr_weights.hdrw")
    frames = [W.synthetic_frame(288, 480, seed=s, kind="gradient") for s in (11, 12, 3)]
    p = HDRTVNetMI355X(hr, use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    ranges = p.calibrate_hg_w8a8(frames)
    ckpt = p.hg_w8a8_checkpoint(ranges)
    test = W.synthetic_frame(272, 480, seed=21, kind="gradient")
    ref, _ = p.infer(p.preprocess(test))
    ref = ref.cpu().numpy()[0]
    with pytest.raises(RuntimeError):
        HDRTVNetMI355X.calibrate_hg_w8a8(type("X", (), {"_hg_state_fp": None})(), frames)
    p.close()
    table = json.load(open(os.path.join(os.path.dirname(W.__file__), "data", "hg_w8a8_calib_seed1234.json")))["ranges"]
    for g, (lo, hi) in ranges.items():
        tlo, thi = table[g]
        print(f"  {g:14s} device [{lo:8.4f}, {hi:8.4f}]  table [{tlo:8.4f}, {thi:8.4f}]")
        assert abs(hi - thi) <= 0.03 * thi + 0.02 and abs(lo - tlo) <= 0.03 * abs(tlo) + 0.02
    pq = HDRTVNetMI355X(hr, use_hg=True, hg_weights=ckpt, warmup_passes=0)
    assert pq._hg_int8
    out, _ = pq.infer(pq.preprocess(test))
    e = np.abs(out.cpu().numpy()[0] - ref)
    print(f"  W8A8 vs fp16 HG output: max {e.max():.3e} mean {e.mean():.3e}")
    assert e.max() <= 5e-2 and e.mean() <= 5e-4
    pq.close()


def test_w8a8_checkpoint_file_and_loader_errors(golden_dir, tmp_path, qstate, zero_style):
    """A W8A8 HG checkpoint from a torch file in the reference's wrapper layout ({"state_dict": ...}, hdrtvnet_torch.py:1491)
    loads through ``hg_weights=<path>``; layouts the int8 path cannot represent exactly are rejected with the layer named."""
    import torch
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    hr = os.path.join(golden_dir, "hr_weights.hdrw")
    path = str(tmp_path / "HG_w8a8.pt")
    torch.save({"state_dict": {k: torch.from_numpy(np.array(v)) for k, v in qstate.items()}, "quantization": "w8a8"}, path)
    p = HDRTVNetMI355X(hr, use_hg=True, hg_weights=path, warmup_passes=0)
    assert p._hg_int8
    f = W.synthetic_frame(96, 128, seed=3, kind="gradient")
    a, _ = p.infer(p.preprocess(f))
    a = a.clone()
    p.close()
    p2 = HDRTVNetMI355X(hr, use_hg=True, hg_weights="seeded-w8a8:1234" if zero_style == "integer-zero" else "seeded-w8a8-minmax:1234",
                        warmup_passes=0)
    b, _ = p2.infer(p2.preprocess(f))
    assert torch.equal(a, b)
    p2.close()
    # a non-integer zero point on one layer loads (round 1 rejected it): that layer switches to code-128 padding + border classes
    odd = dict(qstate)
    odd["Up_conv2.0.x_zero"] = np.array(float(qstate["Up_conv2.0.x_zero"]) + 0.4 * float(qstate["Up_conv2.0.x_scale"]), np.float32)
    p3 = HDRTVNetMI355X(hr, use_hg=True, hg_weights=odd, warmup_passes=0)
    c3, _ = p3.infer(p3.preprocess(f))
    assert np.isfinite(c3.cpu().numpy()).all() and float((c3 - a).abs().max()) <= 5e-2
    p3.close()
    # two readers of one tensor with different quantisers
    bad = dict(qstate)
    bad["conv8.x_scale"] = np.array(float(qstate["conv8.x_scale"]) * 2, np.float32)
    with pytest.raises(ValueError, match="share"):
        HDRTVNetMI355X(hr, use_hg=True, hg_weights=bad, warmup_passes=0)
    # a partially quantised head (one layer left in floating point)
    bad = {k: v for k, v in qstate.items() if not k.startswith("conv5_2.0.")}
    bad["conv5_2.0.weight"] = W.seeded_hg_state(1234)["conv5_2.0.weight"]
    bad["conv5_2.0.bias"] = qstate["conv5_2.0.bias"]
    with pytest.raises(ValueError, match="conv5_2.0"):
        HDRTVNetMI355X(hr, use_hg=True, hg_weights=bad, warmup_passes=0)


def test_private_weight_int8_schedule_is_bit_identical(golden_dir, monkeypatch):
    """conv_prw_i8 (csrc/conv3x3_prw_i8.hip: the private-weight schedule of conv3x3_prw.hip on int8 MFMA) against
    conv_pglds_i8 (variant prw = 0): exact integer sums and the same epilogue arithmetic, so every int8 tensor of the head, the
    dot-product partial sums and the final output agree bit for bit -- at 4K, 1080p and sizes with ragged tiles, with both
    calibrations (integer and float zero points: the border-class constants), 16-row and 8-row tiles (prw = 2 / 3)."""
    import torch
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    taps = ("hg8.p1", "hg8.conv2", "hg8.p3", "hg8.conv3_2", "hg8.p4", "hg8.conv4_2", "hg8.p5", "hg8.conv5_2", "hg8.pc", "hg8.conv_code2",
            "hg8.up1", "hg8.conv6", "hg8.up2", "hg8.conv7", "hg8.up3", "hg8.conv8", "hg8.up4", "hg8.conv9", "hg.part")
    for hgw in ("seeded-w8a8:1234", "seeded-w8a8-minmax:1234"):
        p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True, hg_weights=hgw, warmup_passes=0)
        try:
            for (h, w), seed in (((2160, 3840), 81), ((1080, 1920), 82), ((270, 486), 83), ((97, 131), 84)):
                if hgw.startswith("seeded-w8a8-minmax") and h == 1080:
                    continue
                f = W.synthetic_frame(h, w, seed=seed, kind="gradient")
                res = []
                p.set_variant("prw_i8", 2)                   # the default (1) keeps the 16-row shape off: slower in sustained runs
                for mode in ("0", "1", "2", "3"):
                    p.set_variant("prw", int(mode))
                    out, _ = p.infer(p.preprocess(f))
                    res.append([out.clone()] + [p._tap_device(t).clone() for t in taps])
                    if mode != "0":
                        p.profile_enable(True)
                        p.infer(p.preprocess(f))
                        kern = [k for _, k, _, _, _ in p.profile_read()]
                        p.profile_enable(False)
                        assert sum("conv_prw" in k and "_i8" in k for k in kern) >= (12 if mode != "1" else (7 if h >= 1080 else 0)), (mode, kern)
                for other in res[1:]:
                    for name, a, b in zip(("out",) + taps, res[0], other):
                        assert torch.equal(a, b), (hgw, h, w, name)
        finally:
            p.close()
