"""Two CONTEXTS (own weights, own workspace) on two streams vs two LANES of one context: same box, same frames."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
int8 = "--int8" in sys.argv
H, Wd = 2160, 3840
dev = torch.device("cuda", 0)
def make(lanes):
    p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_int8_full_qat.hdrw" if int8 else "hr_weights.hdrw"), device="cuda:0",
                       precision="int8-full" if int8 else "auto", predequantize="off" if int8 else "auto", use_hg=True,
                       hg_weights="seeded-w8a8:1234" if int8 else "seeded:1234", warmup_passes=0, lanes=lanes)
    p._ensure_buffers(H, Wd)
    return p
two = [make(1), make(1)]
one = make(2)
frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient")).to(dev) for i in range(4)]
outs = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(2)]
def run(n, mode):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 2
        if mode == "ctx":
            two[k].enqueue_frame(0, frames[i % 4].data_ptr(), H, Wd, outs[k].data_ptr())
        elif mode == "lanes":
            one.enqueue_frame(k, frames[i % 4].data_ptr(), H, Wd, outs[k].data_ptr())
        else:
            one.enqueue_frame(0, frames[i % 4].data_ptr(), H, Wd, outs[0].data_ptr())
    torch.cuda.synchronize(dev)
    return n / (time.perf_counter() - t0)
for m in ("one", "ctx", "lanes"):
    run(8, m)
for rep in range(4):
    print(" | ".join(f"{m} {run(60, m):7.2f}" for m in ("one", "ctx", "lanes", "ctx", "lanes")), flush=True)
