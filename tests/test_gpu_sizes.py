"""Sizes the tile geometry does not divide (the playback presets of the reference, src/gui_config.py, plus odd ones):
every kernel has edge tiles, the HG head pads to a multiple of 32, LE crops / replicate-pads its skips.  Checks:
finite, bit-reproducible, and -- at the sizes the oracle finishes in seconds -- within the parity tolerances."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def proc(golden_dir):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    p = HDRTVNetMI355X(os.path.join(golden_dir, "hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    yield p
    p.close()


@pytest.mark.parametrize("hw", [(480, 854), (768, 1366), (720, 1280), (1200, 1600), (1440, 2560), (1080, 1918), (1082, 1920)])
def test_edge_tile_sizes(proc, hw):
    import torch
    from hdrtv_mi355x import weights as W
    h, w = hw
    f = W.synthetic_frame(h, w, seed=h + w, kind="gradient")
    out1 = proc.infer(proc.preprocess(f))[0].clone()
    u8 = proc.postprocess(out1).copy()
    out2 = proc.infer(proc.preprocess(f))[0].clone()
    assert tuple(out1.shape) == (1, 3, h, w) and torch.isfinite(out1).all() and torch.equal(out1, out2)
    assert u8.shape == (h, w, 3)
    base = proc.tap("le.out")
    assert torch.isfinite(base).all() and float(base.abs().max()) < 16.0


def test_480p_vs_oracle(proc, hr_state, hg_state):
    from hdrtv_mi355x import weights as W
    from oracle import hdrtvnet_oracle as O
    h, w = 480, 854
    f = W.synthetic_frame(h, w, seed=99, kind="gradient")
    out, agcm = proc.infer(proc.preprocess(f))
    base = proc.tap("le.out").numpy()
    rt, rc = O.preprocess(f)
    rbase, ragcm = O.hr_forward(hr_state, rt, rc)
    assert np.abs(agcm.float().cpu().numpy()[0] - ragcm).max() <= 2e-3
    assert np.abs(base - rbase).max() <= 6e-3 and np.abs(base - rbase).mean() <= 5e-4
    mask = O.hg_mask(base)
    ph, pw = (32 - h % 32) % 32, (32 - w % 32) % 32
    ref = O.hg_generator(hg_state, np.pad(base, ((0, 0), (0, ph), (0, pw)), mode="reflect"),
                         np.pad(mask, ((0, 0), (0, ph), (0, pw)), mode="reflect"))[:, :h, :w]
    d = np.abs(out.cpu().numpy()[0] - ref)
    print(f"  480x854: hg max {d.max():.3e} mean {d.mean():.3e}")
    assert d.max() <= 3e-3 and d.mean() <= 1e-4
