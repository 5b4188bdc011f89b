"""Every lane, rotating frames: quiet yardsticks for 4 frames (outputs; for lane 0 also every workspace tensor), then rounds of
frames on all lanes; any lane's RGB48 that differs from its frame's yardstick is reported, lane 0's with the first differing tensors.
usage: lane_stress2.py [--int8] [--rounds N] [--lanes L]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
os.environ.setdefault("HDRTV_LANES_ANY", "1")      # more than two lanes, and lanes for any precision: experiments only
import torch
from hdrtv_mi355x import weights as W
from hdrtv_mi355x.processor import HDRTVNetMI355X
int8 = "--int8" in sys.argv
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 20
lanes = int(sys.argv[sys.argv.index("--lanes") + 1]) if "--lanes" in sys.argv else 3
H, Wd = 2160, 3840
dev = torch.device("cuda", 0)
p = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_int8_full_qat.hdrw" if int8 else "hr_weights.hdrw"), device="cuda:0",
                   precision="int8-full" if int8 else "auto", predequantize="off" if int8 else "auto", use_hg=True,
                   hg_weights="seeded-w8a8:1234" if int8 else "seeded:1234", warmup_passes=0, lanes=lanes)
p._ensure_buffers(H, Wd)
names = [f"agcm.{k}{i}" for i in range(1, 6) for k in ("u", "mean", "rstd")] + ["agcm.part", "agcm.bias", "agcm.qconst" if int8 else "agcm.frags", "agcm.out", "le.cond", "le.cond1", "le.x192", "le.cond2", "le.cond3", "le.cond4",
         "le.f0a", "le.f0b", "le.fea0", "le.fea1a", "le.fea1", "le.l1b", "le.fea2a", "le.fea2", "le.l2b", "le.fea3", "le.l3b", "le.t3x", "le.t3y",
         "le.up1", "le.t4", "le.up2", "le.t5", "le.up3", "le.out", "hg.img", "hg.mask", "hg.part2"]
pre = "hg8." if int8 else "hg."
names += [pre + n for n in ("p1", "conv2", "p3", "conv3_2", "p4", "conv4_2", "p5", "conv5_2", "pc", "conv_code2", "up1", "conv6", "up2", "conv7", "up3", "conv8", "up4", "conv9")] + ["hg.part"]
have = []
for n in names:
    try:
        p._tap_device(n); have.append(n)
    except RuntimeError:
        pass
frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=70 + i, kind=("noise", "gradient", "noise", "gradient")[i])).to(dev) for i in range(4)]
outs = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(lanes)]
want, ref = [], []
for f in range(4):
    p.enqueue_frame(0, frames[f].data_ptr(), H, Wd, outs[0].data_ptr())
    torch.cuda.synchronize(dev)
    want.append(outs[0].clone())
    ref.append({n: p._tap_device(n) for n in have})
    ref[-1]["cond"] = p._lane_bufs[0][1].clone()
bad = 0
for r in range(rounds):
    sel = [(r + 3 * l + (r // 4)) % 4 for l in range(lanes)]
    for rep in range(2):
        for l in (range(lanes) if r % 2 == 0 else reversed(range(lanes))):
            p.enqueue_frame(l, frames[sel[l]].data_ptr(), H, Wd, outs[l].data_ptr())
    torch.cuda.synchronize(dev)
    for l in range(lanes):
        nd = int((outs[l] != want[sel[l]]).sum())
        if nd:
            bad += 1
            msg = f"round {r}: lane {l} frame {sel[l]} (others: {sel}): {nd} RGB48 values differ"
            if l == 0:
                d = [("cond", int((ref[sel[0]]["cond"] != p._lane_bufs[0][1]).sum()))] + [(n, int((ref[sel[0]][n] != p._tap_device(n)).sum())) for n in have]
                msg += "; lane-0 tensors that differ, in launch order: " + ", ".join(f"{n} {k}" for n, k in d if k)[:300]
                first = [n for n, k in d if k][0]
                if first != "cond":
                    a, b = ref[sel[0]][first], p._tap_device(first)
                    idx = (a != b).nonzero().flatten()
                    C_ = {"hg8.conv2": 128, "hg.conv2": 128}.get(first, 0)
                    if C_:
                        wq = Wd // 2 if (Wd // 2) % 32 == 0 else ((Wd // 2 + 31) // 32) * 32
                        px, ch = idx // C_, idx % C_
                        yy, xx = px // (a.numel() // C_ // ((a.numel() // C_) // wq) if False else wq), px % wq
                        msg += f"\n   {first}: rows {int(yy.min())}..{int(yy.max())} cols {int(xx.min())}..{int(xx.max())} channels {int(ch.min())}..{int(ch.max())}; distinct rows {len(torch.unique(yy))} cols {len(torch.unique(xx))} channels {len(torch.unique(ch))}; sample want/got " + " ".join(f"{int(a[i])}/{int(b[i])}" for i in idx[:12])
                        msg += f"\n   rows: {torch.unique(yy)[:40].tolist()} cols: {torch.unique(xx)[:40].tolist()}"
            print(msg, flush=True)
print(f"{'int8' if int8 else 'fp16'} lanes {lanes}: {bad} lane-frames of {rounds * lanes} differ", flush=True)
p.close()
