"""A/B of the HG 3x3 conv schedules on one box: HDRTV_VARIANTS=prw=0 (conv_pglds) against prw=1 (conv_prw), taps compared bit for bit,
per-layer HIP-event times printed side by side.  python tools/prw_ab.py [H W]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAPS = ("hg.conv2", "hg.conv3_2", "hg.conv4_2", "hg.conv5_2", "hg.conv_code2", "hg.conv6", "hg.conv7", "hg.conv8", "hg.conv9", "hg.part")


def child(h, w, out):
    sys.path[:0] = [REPO, os.path.join(REPO, "hdr-realtime-video-pipeline_amd")]
    import torch
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    p = HDRTVNetMI355X(os.path.join(REPO, "tests/golden/hr_weights.hdrw"), use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
    f = W.synthetic_frame(h, w, seed=21, kind="gradient")
    t, c = p.preprocess(f)
    o, _ = p.infer((t, c))
    res = {"out": o.clone().cpu()}
    for n in TAPS:
        res[n] = p._tap_device(n).cpu()
    for _ in range(5):
        p.infer((t, c))
    p.profile_enable(True)
    acc = {}
    for _ in range(10):
        p.infer((t, c))
        torch.cuda.synchronize()
        for layer, kern, ms, macs, nb in p.profile_read():
            a = acc.setdefault(layer, [kern, 0.0, macs])
            a[1] += ms / 10
    res["prof"] = acc
    torch.save(res, out)
    p.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "2", "3", "1"]
    import torch
    outs = []
    for v in modes:
        path = f"/tmp/prw_ab_{v}.pt"
        env = dict(os.environ, HDRTV_VARIANTS='prw=' + v)
        subprocess.run([sys.executable, __file__, "--child", str(h), str(w), path], env=env, check=True, stdout=subprocess.DEVNULL)
        outs.append(torch.load(path, weights_only=False))
    a = outs[0]
    for v, b in zip(modes[1:], outs[1:]):
        bad = [n for n in ("out",) + TAPS if not torch.equal(a[n], b[n])]
        for n in bad:
            print(f"   {n}: max|d| = {float((a[n].float() - b[n].float()).abs().max()):.3e}")
        print(f"prw={v} vs {modes[0]}: {'all taps bit-identical' if not bad else 'DIFFERENT: ' + ' '.join(bad)}")
    tot = [0.0] * len(modes)
    print(f"{'layer':14s} " + " | ".join(f"PRW={v:1s} kernel              ms     TF" for v in modes))
    for layer, (kern, ms, macs) in a["prof"].items():
        if "hg." in layer and ("pglds" in kern or "prw" in kern):
            cells = []
            for i, o in enumerate(outs):
                kb, msb, _ = o["prof"][layer]
                tot[i] += msb
                cells.append(f"{kb:20s} {msb:6.3f} {2 * macs / msb / 1e9:6.0f}")
            print(f"{layer:14s} " + " | ".join(cells))
    print("HG 3x3 total ms: " + "  ".join(f"PRW={v}: {t:.3f}" for v, t in zip(modes, tot)))
    print("frame ms:        " + "  ".join(f"PRW={v}: {sum(x[1] for x in o['prof'].values()):.3f}" for v, o in zip(modes, outs)))
