"""ctypes binding of libhdrtv_mi355x.so (C ABI: include/hdrtv_mi355x.h).

There is no CPU fallback: if the shared library is missing, or the device is not
gfx950, construction raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C hdr-realtime-video-pipeline_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_PKG), "lib", "libhdrtv_mi355x.so")
# the A/B build (make AB=1): the same sources plus superseded kernels, for the GPU bit-identity tests only
LIB_PATH_AB = os.path.join(os.path.dirname(_PKG), "lib", "libhdrtv_mi355x_ab.so")

OK, EINVAL, EWEIGHTS, EHIP, ENOMEM, ESTATE = 0, -1, -2, -3, -4, -5
F16, F32 = 0, 1
PREC_F16, PREC_F32 = 0, 1

# every symbol include/hdrtv_mi355x.h declares: (name, restype, argtypes)
_VP, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = [
    ("hdrtv_version", C.c_char_p, []),
    ("hdrtv_create", _I, [_VP, _SZ, _VP, _SZ, _I, C.POINTER(_VP)]),
    ("hdrtv_create_ex", _I, [_VP, _SZ, _VP, _SZ, _I, _I, C.POINTER(_VP)]),
    ("hdrtv_destroy", _I, [_VP]),
    ("hdrtv_has_hg", _I, [_VP]),
    ("hdrtv_set_hg_mask_r", _I, [_VP, C.c_float]),
    ("hdrtv_set_cond_mode", _I, [_VP, _I]),
    ("hdrtv_reserve", _I, [_VP, _I, _I]),
    ("hdrtv_preprocess", _I, [_VP, _VP, _VP, _I, _I, _VP, _VP]),
    ("hdrtv_infer", _I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _I, _VP]),
    ("hdrtv_set_lanes", _I, [_VP, _I]),
    ("hdrtv_get_lanes", _I, [_VP]),
    ("hdrtv_infer_lane", _I, [_VP, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _VP]),
    ("hdrtv_post_u8", _I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    ("hdrtv_post_rgb48", _I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    ("hdrtv_post_pq_rgb48", _I, [_VP, _VP, _VP, _I, _I, _I, C.c_float, _VP]),
    ("hdrtv_letterbox_u8", _I, [_VP, _VP, _VP, _I, _I, _VP, _I, _I]),
    ("hdrtv_metrics", _I, [_VP, _VP, _VP, _VP, _I, _I, _I, C.c_float, C.POINTER(C.c_double)]),
    ("hdrtv_ring_create", _I, [_VP, _I, _I, _I]),
    ("hdrtv_ring_acquire", _I, [_VP, _I, C.POINTER(_VP), C.POINTER(_VP)]),
    ("hdrtv_ring_commit", _I, [_VP, _I, _VP]),
    ("hdrtv_ring_wait", _I, [_VP, _I]),
    ("hdrtv_ring_release", _I, [_VP, _I]),
    ("hdrtv_ring_destroy", _I, [_VP]),
    ("hdrtv_get_tap", _I, [_VP, C.c_char_p, C.POINTER(_VP), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    ("hdrtv_infer_stats", _I, [_VP, C.POINTER(_I), C.POINTER(C.c_double)]),
    ("hdrtv_profile_enable", _I, [_VP, _I]),
    ("hdrtv_profile_get", _I, [_VP, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                               C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("hdrtv_set_variant", _I, [_VP, C.c_char_p, _I]),
    ("hdrtv_get_variant", _I, [_VP, C.c_char_p, C.POINTER(_I)]),
    ("hdrtv_last_error", C.c_char_p, [_VP]),
]

_libs = {}


def load(ab=False):
    """Load the shared library (once).  Raises RuntimeError when it has not been built.  ``ab=True``: the A/B library."""
    path = LIB_PATH_AB if ab else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the MI355X HIP extension is not built and there is no "
            "fallback path (run __graft_entry__.build())")
    # PyTorch ships its own HIP runtime (libamdhip64 with the system library's SONAME).  Whichever copy a process loads
    # first serves both users, and torch on the system copy finds no device: make torch's the first.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


def build_id(ab=False):
    """The library's build id: the hash over its sources that csrc/Makefile stamps into hdrtv_version() ("... build <id>")."""
    v = load(ab).hdrtv_version().decode()
    return v.rsplit(" build ", 1)[1] if " build " in v else "unstamped"


def source_hash(name):
    """First 12 hex digits of the SHA-1 of csrc/<name>: what tools/pmc_to_json.py records per source file."""
    import hashlib
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", name), "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()[:12]


class HdrtvError(RuntimeError):
    pass


def check(lib, ctx, rc, what):
    if rc < 0:
        msg = lib.hdrtv_last_error(ctx)
        raise HdrtvError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
