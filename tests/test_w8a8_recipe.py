"""The W8A8 HG quantisation recipe (weights.HG_W8A8_GROUPS / activation_qparams / hg_w8a8_state): host logic only."""
import json
import os

import numpy as np

from hdrtv_mi355x import weights as W


def test_activation_qparams_have_integer_zero_points():
    rng = np.random.default_rng(0)
    for _ in range(200):
        lo, hi = -abs(rng.normal()) * rng.integers(0, 2), abs(rng.normal()) * 5 + 1e-3
        s, z = W.activation_qparams(lo, hi)
        k = -np.float32(z) / np.float32(s)
        assert k == np.rint(k) and 0 <= k <= 255
        assert np.float32(np.float16(s)) == np.float32(s)                 # fp16-representable, as the checkpoint stores it
        # the range is covered: codes 0..255 dequantise to [z, z + 255 s]
        assert z <= min(lo, 0.0) + 0.5 * s + 1e-6 and z + 255 * s >= hi - 0.5 * s - 1e-6
    assert W.activation_qparams(0.0, 2.55)[1] == 0.0                       # post-ReLU tensors: k = 0


def test_hg_w8a8_state_layout_and_sharing():
    table = json.load(open(os.path.join(os.path.dirname(W.__file__), "data", "hg_w8a8_calib_seed1234.json")))["ranges"]
    assert list(table) == list(W.HG_W8A8_GROUPS)
    fp = W.seeded_hg_state(1234)
    q = W.seeded_hg_w8a8_state(1234)
    assert W.is_int8_state(q) and not W.is_int8_state(fp)
    quantised = [n for layers in W.HG_W8A8_GROUPS.values() for n in layers]
    assert len(quantised) == 18 and len(set(quantised)) == 18
    for layers in W.HG_W8A8_GROUPS.values():
        assert len({(float(q[n + ".x_scale"]), float(q[n + ".x_zero"])) for n in layers}) == 1
    for n in quantised:
        w = fp[n + ".weight"]
        ws = q[n + ".w_scale"]
        assert q[n + ".weight_int8"].dtype == np.int8 and q[n + ".weight_int8"].shape == w.shape and n + ".weight" not in q
        assert np.allclose(ws, np.abs(w.reshape(w.shape[0], -1)).max(1) / 127.0, rtol=1e-6)
        deq = q[n + ".weight_int8"].astype(np.float32) * ws.reshape(-1, 1, 1, 1)
        assert np.abs(deq - w).max() <= 0.5 * ws.max() * 1.0001               # round to nearest
        assert np.abs(q[n + ".weight_int8"].astype(int)).max() == 127
    # everything else passes through bit for bit (BatchNorm, fp16 layers)
    for k, v in fp.items():
        if not any(k == n + ".weight" for n in quantised):
            assert np.array_equal(q[k], v), k
    # the pack format carries it
    back = W.unpack_state(W.pack_state({k: v for k, v in q.items() if not k.endswith("num_batches_tracked")}))
    assert back["conv4_1.0.weight_int8"].dtype == np.int8 and float(back["conv8.x_zero"].reshape(-1)[0]) == float(q["conv8.x_zero"])
