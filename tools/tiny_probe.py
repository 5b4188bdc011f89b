import sys, os
sys.path.insert(0, 'hdr-realtime-video-pipeline_amd'); sys.path.insert(0, '.')
import numpy as np
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
from oracle import hdrtvnet_oracle as O
hr = W.load_pack('tests/golden/hr_weights.hdrw')
p = HDRTVNetMI355X('tests/golden/hr_weights.hdrw', use_hg=False, warmup_passes=0)
for (h, w) in [(68, 8), (8, 68), (68, 12), (65, 9), (9, 70), (40, 72), (33, 130)]:
    try:
        f = W.synthetic_frame(h, w, seed=3, kind='noise')
        got = p.process(f)
        want = O.process(hr, f)
        d = np.abs(got.astype(int) - want.astype(int))
        print(h, w, 'ok max', d.max(), 'mean', d.mean())
    except Exception as e:
        print(h, w, 'ERR', type(e).__name__, str(e)[:160])
p.close()
ph = HDRTVNetMI355X('tests/golden/hr_weights.hdrw', use_hg=True, hg_weights='seeded:1234', warmup_passes=0)
hg = W.seeded_hg_state(1234) if hasattr(W, 'seeded_hg_state') else None
for (h, w) in [(68, 8), (8, 68), (65, 33), (33, 130), (68, 40)]:
    try:
        f = W.synthetic_frame(h, w, seed=3, kind='gradient')
        got = ph.process(f)
        print('hg', h, w, 'ok', got.shape, 'finite', bool(np.isfinite(got.astype(float)).all()))
    except Exception as e:
        print('hg', h, w, 'ERR', type(e).__name__, str(e)[:160])
ph.close()
