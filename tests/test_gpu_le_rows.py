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
            if h * w >= 720 * 1280:
                assert any(k.startswith("le_") and "rows" in k for k in kernels[1]), kernels[1]
            assert not any("rows" in k for k in kernels[0]), kernels[0]
            for other in res[1:]:
                for name, a, b in zip(("out",) + taps, res[0], other):
                    assert torch.isfinite(a).all(), (h, w, name)
                    assert torch.equal(a, b), (h, w, name, int((a != b).sum()))
        with pytest.raises(RuntimeError):
            p.set_variant("no_such_variant", 1)
    finally:
        p.close()
