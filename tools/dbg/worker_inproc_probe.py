"""The dispatcher worker's device side without the processes: _Mi355xWorker.begin / finish driven in this process over host
slots that are (a) hipHostRegister'ed ordinary memory, as the shared-memory slots are, or (b) torch pinned memory.
usage: python tools/dbg/worker_inproc_probe.py"""
import collections, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import numpy as np
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.dispatch import _Mi355xWorker

H, Wd = 2160, 3840
frames = [W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient") for i in range(2)]

def run(depth, lanes, slots, pinned, n=80):
    init = {"model_path": os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), "use_hg": True, "hg_weights": "seeded:1234",
            "frames_in_flight": depth, "lanes": lanes}
    w = _Mi355xWorker(0, 0, init)
    in_b, out_b = H * Wd * 3, H * Wd * 6
    if pinned:
        buf = torch.empty(slots * (in_b + out_b), dtype=torch.uint8, pin_memory=True).numpy()
    else:
        raw = np.empty(slots * (in_b + out_b) + 4096, dtype=np.uint8)
        off = (-raw.ctypes.data) % 4096
        buf = raw[off:off + slots * (in_b + out_b)]
        w.pin(memoryview(buf))
    ins = [np.ndarray((H, Wd, 3), np.uint8, buf, offset=s * in_b) for s in range(slots)]
    outs = [np.ndarray((H, Wd, 3), np.uint16, buf, offset=slots * in_b + s * out_b) for s in range(slots)]
    for s in range(slots):
        np.copyto(ins[s], frames[s % 2])
    inflight = collections.deque()
    def loop(n):
        t0 = time.perf_counter()
        for i in range(n):
            while len(inflight) >= depth:
                w.finish(inflight.popleft())
            inflight.append(w.begin(ins[i % slots], outs[i % slots]))
        while inflight:
            w.finish(inflight.popleft())
        return n / (time.perf_counter() - t0)
    loop(8)
    r = [loop(n) for _ in range(2)]
    w.close()
    return max(r)

for pinned in (False, True):
    for depth, lanes, slots in ((2, 1, 3), (2, 2, 3), (4, 2, 6), (4, 1, 6)):
        print(f"{'torch pinned' if pinned else 'hipHostRegister'}  depth {depth} lanes {lanes} slots {slots}: {run(depth, lanes, slots, pinned):7.2f} frames/s", flush=True)
