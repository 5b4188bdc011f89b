"""How many frames in flight pay?  One context with 4 lanes (hdrtv_set_lanes); K frames with frame i on lane i mod L for
L = 1 .. 4, device-resident in and out (no ring).  usage: python tools/dbg/two_stream_probe.py [--int8] [HxW]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
os.environ.setdefault("HDRTV_LANES_ANY", "1")      # more than two lanes, and lanes for any precision: experiments only
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X

int8 = "--int8" in sys.argv
size = [a for a in sys.argv[1:] if "x" in a]
H, Wd = (int(v) for v in size[0].split("x")) if size else (2160, 3840)
dev = torch.device("cuda", 0)
p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_int8_full_qat.hdrw" if int8 else "hr_weights.hdrw"), device="cuda:0",
                   precision="int8-full" if int8 else "auto", predequantize="off" if int8 else "auto", use_hg=True,
                   hg_weights="seeded-w8a8:1234" if int8 else "seeded:1234", warmup_passes=0, lanes=4)
p._ensure_buffers(H, Wd)
frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient")).to(dev) for i in range(4)]
outs = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(4)]

def run(n, lanes):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n):
        p.enqueue_frame(i % lanes, frames[i % 4].data_ptr(), H, Wd, outs[i % lanes].data_ptr())
    torch.cuda.synchronize(dev)
    return n / (time.perf_counter() - t0)

for lanes in (1, 2, 3, 4):
    run(8, lanes)
for rep in range(3):
    r = [run(60, lanes) for lanes in (1, 2, 3, 4)]
    print(f"{H}x{Wd} {'int8' if int8 else 'fp16'}: " + " | ".join(f"{l} lane{'s' if l > 1 else ' '} {v:7.2f} ({100 * (v / r[0] - 1):+.1f} %)" for l, v in zip((1, 2, 3, 4), r)), flush=True)
p.close()
