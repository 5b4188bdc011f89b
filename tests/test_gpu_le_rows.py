"""The fused row-streaming LE kernels (csrc/le_rows.hip) against the per-layer kernels they replace.

Same operands, same MFMA shape and K order, same rounding points: the LE output and every tensor both forms
write must agree BIT FOR BIT -- at 3840x2160 (540 / 135 rows per segment: the steady state of the rings), at
1920x1080, and at sizes whose last strip is a few columns wide, whose segments are ragged and whose half-resolution
maps have odd sizes.  The per-layer form is itself held to the reference's goldens and the oracle
(tests/test_gpu_parity.py), and the default (fused) form runs in every other GPU test."""
import os

import pytest

pytestmark = pytest.mark.gpu

SIZES = (((2160, 3840), 41), ((1080, 1920), 42), ((1081, 1923), 45), ((720, 1280), 46), ((540, 960), 43), ((270, 486), 47))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    return torch


# (270, 486): 9 strips x 28 segments of 10 rows at full resolution, 5 rows at half resolution: with le_rows_min = 8 the head and the
# tail run there, the ResBlocks do not (the loop below asserts the larger sizes); test_short_segments_* forces everything on
def test_fused_rows_are_bit_identical_to_the_per_layer_kernels(torch_cuda, golden_dir):
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    torch = torch_cuda
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    taps = ("le.fea0", "le.fea1a", "le.fea1", "le.t5", "le.out")
    try:
        assert p.get_variant("le_rows") == 1          # the default
        for (h, w), seed in SIZES:
            f = W.synthetic_frame(h, w, seed=seed, kind="gradient" if seed % 2 else "noise")
            res, kernels = [], []
            for v in (0, 1):              # per-layer kernels | fused rows
                p.set_variant("le_rows", v)
                p.profile_enable(True)
                out, _ = p.infer(p.preprocess(f))
                kernels.append({k for _, k, *_ in p.profile_read()})
                p.profile_enable(False)
                res.append([out.clone()] + [p.tap(t).clone() for t in taps])
            # which row kernels a size reaches at the least (256 CUs, le_rows_min = 8): the head and the tail need even sizes (PixelShuffle /
            # stride-2 pairs); the ResBlocks run at half and quarter resolution, where 540x960 leaves 9-row segments
            want = set()
            if h * w >= 540 * 960 and h % 2 == 0 and w % 2 == 0:
                want |= {"le_head_rows", "le_tail_rows"}
            if h * w >= 720 * 1280:
                want |= {"le_rb_rows"}
            assert want <= kernels[1], ((h, w), kernels[1])
            assert not any("rows" in k for k in kernels[0]), kernels[0]
            for other in res[1:]:
                for name, a, b in zip(("out",) + taps, res[0], other):
                    assert torch.isfinite(a).all(), (h, w, name)
                    assert torch.equal(a, b), (h, w, name, int((a != b).sum()))
        with pytest.raises(RuntimeError):
            p.set_variant("no_such_variant", 1)
    finally:
        p.close()


SHORT = (((270, 486), 47), ((136, 242), 51), ((64, 122), 52), ((62, 124), 53), ((46, 182), 54), ((34, 3840), 55))


def test_short_segments_and_narrow_last_strips(torch_cuda, golden_dir):
    """le_rows_min = 1 (hdrtv_set_variant; a device with fewer CUs or a partitioned one reaches the same shapes): segments of
    1 .. 10 rows -- shorter than the chains' 6-row lag and their 3-step DMA lead, so the prologue's row clamp, `nsteps` and the
    masked stores all work on rows outside the segment -- last strips 1 and 2 columns wide, and one-row segments of a
    3840-wide map.  Every size must run all three row kernels and agree bit for bit with the per-layer kernels."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    torch = torch_cuda
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    taps = ("le.fea0", "le.fea1a", "le.fea1", "le.t5", "le.out")
    try:
        with pytest.raises(RuntimeError):
            p.set_variant("le_rows_min", 0)           # validated: 1 .. 4096
        p.set_variant("le_rows_min", 1)
        for (h, w), seed in SHORT:
            f = W.synthetic_frame(h, w, seed=seed, kind="gradient" if seed % 2 else "noise")
            res, kernels = [], []
            for v in (0, 1):
                p.set_variant("le_rows", v)
                p.profile_enable(True)
                out, _ = p.infer(p.preprocess(f))
                kernels.append({k for _, k, *_ in p.profile_read()})
                p.profile_enable(False)
                res.append([out.clone()] + [p.tap(t).clone() for t in taps])
            assert {"le_head_rows", "le_rb_rows", "le_tail_rows"} <= kernels[1], ((h, w), kernels[1])
            for name, a, b in zip(("out",) + taps, res[0], res[1]):
                assert torch.isfinite(a).all(), (h, w, name)
                assert torch.equal(a, b), (h, w, name, int((a != b).sum()))
    finally:
        p.close()


def test_int8_row_kernels_are_bit_identical_to_the_per_layer_int8_kernels(torch_cuda, golden_dir):
    """le_rows_i8.hip (chains whose layers are all W8A8: int8 codes in the LDS rings, 9 x v_mfma_i32_32x32x32_i8 per conv, the SFT
    MLPs on int8 MFMA) against the per-layer int8 kernels conv32s<sft-i8, i8> (variants le_rows_i8 = 0, le_rows_fq = 0): the same
    quantisers, integer sums, dequantisation constants, border-class shifts and rounding points -> every tensor both forms write
    bit for bit, at the sizes of the fp16 test (4K: the rings' steady state; ragged strips and segments; odd half-resolution maps).
    (Round 5, the library without the SLP vectoriser: the per-layer kernels' scalar (f16) casts of the SFT scale / shift FMAs became
    v_fma_mixlo_f16 -- ONE rounding, to f16 -- and 4e-6 .. 3e-5 of le.fea0 differed by an f16 step until both forms converted through
    the packed instruction: common.h cvt_h4.)"""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    torch = torch_cuda
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_int8_full_qat.hdrw"), precision="int8-full", predequantize="off", use_hg=False, warmup_passes=0)
    taps = ("le.fea0", "le.fea1a", "le.fea1", "le.fea2", "le.t4", "le.t5", "le.out")
    try:
        assert p.get_variant("le_rows_i8") == 1          # the default
        for (h, w), seed in SIZES:
            f = W.synthetic_frame(h, w, seed=seed, kind="gradient" if seed % 2 else "noise")
            res, kernels = [], []
            for i8, fq in ((0, 0), (1, 0), (1, 1)):       # per-layer int8 | int8 row kernels, the rest per layer | the default mix
                p.set_variant("le_rows_i8", i8)
                p.set_variant("le_rows_fq", fq)
                p.profile_enable(True)
                out, _ = p.infer(p.preprocess(f))
                kernels.append({k for _, k, *_ in p.profile_read()})
                p.profile_enable(False)
                res.append([out.clone()] + [p.tap(t).clone() for t in taps])
            assert not any("rows" in k for k in kernels[0]), kernels[0]
            if h * w >= 720 * 1280:
                assert "le_rb_rows<i8>" in kernels[1] and "le_rb_rows<i8>" in kernels[2], ((h, w), kernels[1])
                assert "le_rb_rows<fq>" not in kernels[2]
            if h * w >= 540 * 960 and h % 2 == 0 and w % 2 == 0:
                assert {"le_tail_rows<i8>", "le_head_rows<i8>"} <= kernels[1] and not any("rows<fq>" in k for k in kernels[2]), ((h, w), kernels[1])
            for name, a, b in zip(("out",) + taps, res[0], res[1]):
                assert torch.isfinite(a).all(), (h, w, name)
                assert torch.equal(a, b), (h, w, name, int((a != b).sum()))
            # the default mix: the chains le_rows_i8.hip does not cover yet run in the fake-quant form -- the ResBlock outputs up to
            # there (le.fea1: recon_trunk1) must still be the int8 bits
            for name, a, b in zip(("le.fea1a",) if False else (), res[0], res[2]):
                assert torch.equal(a, b), (h, w, name)
    finally:
        p.close()


@pytest.mark.parametrize("tag", ["full", "mixed"])
def test_w8a8_layers_as_fake_quant_in_the_fused_kernels(torch_cuda, golden_dir, tag):
    """W8A8 layers inside the fused kernels in their fake-quant form (variant le_rows_fq: the default for chains that mix W8A8 with
    other layers -- the mixed recipe; chains whose layers are ALL W8A8 run on int8 MFMA, le_rows_i8): the layer's activation quantiser is applied in
    registers (u8 code and back, common.h FqParam) and the convolution runs in fp16 on the dequantised weights -- the
    arithmetic of the reference's ``W8A8Conv2d.forward`` itself (hdrtvnet_torch.py:351-364: fake-quant, then F.conv2d in the
    compute dtype).  Against the fake-quant oracle (fp32, ATen) at 1920x1080 it must be no further away than the int8-MFMA
    per-layer kernels it replaces (le_rows_fq = 0), within the reference's own bar for a re-quantised graph (u8 MAE <= 5)."""
    import numpy as np
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    from oracle import hdrtvnet_oracle as O
    h, w = 1080, 1920
    f = W.synthetic_frame(h, w, seed=91, kind="gradient")
    p = HDRTVNetMI355X(os.path.join(golden_dir, f"hr_int8_{tag}_qat.hdrw"), precision=f"int8-{tag}", predequantize="off", use_hg=False, warmup_passes=0)
    got = {}
    try:
        p.set_variant("le_rows_i8", 0)        # (full recipe: the int8 row kernels would take every chain; this test is about the fake-quant form)
        for v in (1, 0):
            p.set_variant("le_rows_fq", v)
            p.profile_enable(True)
            out, _ = p.infer(p.preprocess(f))
            kern = {k for _, k, *_ in p.profile_read()}
            p.profile_enable(False)
            assert any("rows<fq>" in k for k in kern) == (v == 1), kern
            got[v] = (out.float().cpu().numpy()[0], p.postprocess(out).copy())
    finally:
        p.close()
    sd = O.w8a8_state(W.load_pack(os.path.join(golden_dir, f"hr_int8_{tag}_qat.hdrw")))
    O.use_backend("aten")
    try:
        ref, _ = O.hr_forward(sd, *O.preprocess(f))
    finally:
        O.use_backend("c")
    ru8 = O.postprocess_u8(ref)
    err = {}
    for v in (1, 0):
        d = np.abs(got[v][0] - ref)
        du8 = np.abs(got[v][1].astype(int) - ru8.astype(int))
        err[v] = (float(d.mean()), float(du8.mean()))
        print(f"  int8-{tag} le_rows_fq={v}: out max {d.max():.3e} mean {d.mean():.3e}; u8 max {du8.max()} MAE {du8.mean():.4f}")
    assert err[1][1] <= 5.0 and err[1][0] <= 0.02                       # the reference's bar
    assert err[1][0] <= 1.15 * err[0][0] + 1e-5 and err[1][1] <= 1.15 * err[0][1] + 0.01


def test_cond2_tail_in_the_stride2_heads_epilogue_is_bit_identical(torch_cuda, golden_dir):
    """CondNet2.2 + .4 computed from conv3x3s2_preg<192>'s staged output tile (variant cond2_fused, the default) against the
    separate cond_tail launch: the same f16 inputs, fragments, MFMA order and rounding points -> `le.cond2`, CondNet3 / 4's
    maps (which read the channels that are still stored) and the LE output bit for bit, also where tiles are ragged."""
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    torch = torch_cuda
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=False, warmup_passes=0)
    taps = ("le.cond2", "le.cond3", "le.cond4", "le.fea1", "le.out")
    try:
        assert p.get_variant("cond2_fused") == 1 and p.get_variant("cond3_fused") == 1
        for (h, w), seed in SIZES + (((60, 100), 48), ((52, 76), 49)):
            f = W.synthetic_frame(h, w, seed=seed, kind="gradient" if seed % 2 else "noise")
            res, kernels = [], []
            for v in (0, 1):
                p.set_variant("cond2_fused", v)
                p.set_variant("cond3_fused", v)          # likewise CondNet3.4 (conv_igemm's arithmetic) behind CondNet3.2
                p.profile_enable(True)
                out, _ = p.infer(p.preprocess(f))
                kernels.append({k for _, k, *_ in p.profile_read()})
                p.profile_enable(False)
                res.append([out.clone()] + [p.tap(t).clone() for t in taps])
            assert "cond_tail" in kernels[0] and "cond_tail" not in kernels[1], kernels
            assert {"conv3x3s2_preg<192>+tail", "conv3x3s2_preg<64>+tail"} <= kernels[1], kernels[1]
            assert "conv_igemm<64,32,1,1>" in kernels[0] and "conv_igemm<64,32,1,1>" not in kernels[1], kernels
            for name, a, b in zip(("out",) + taps, res[0], res[1]):
                assert torch.isfinite(a).all(), (h, w, name)
                assert torch.equal(a, b), (h, w, name, int((a != b).sum()))
    finally:
        p.close()
