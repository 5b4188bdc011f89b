"""A/B of the HG 3x3 conv schedules on one box: HDRTV_PRW=0 (conv_pglds) against HDRTV_PRW=1 (conv_prw), taps compared bit for bit,
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
    import torch
    outs = []
    for v in ("0", "1"):
        path = f"/tmp/prw_ab_{v}.pt"
        env = dict(os.environ, HDRTV_PRW=v)
        subprocess.run([sys.executable, __file__, "--child", str(h), str(w), path], env=env, check=True, stdout=subprocess.DEVNULL)
        outs.append(torch.load(path, weights_only=False))
    a, b = outs
    for n in ("out",) + TAPS:
        same = torch.equal(a[n], b[n])
        extra = "" if same else f"  max|d|={float((a[n].float() - b[n].float()).abs().max()):.3e} n_diff={int((a[n] != b[n]).sum())}"
        print(f"{n:16s} {'bit-identical' if same else 'DIFFERENT'}{extra}")
    tot = [0.0, 0.0]
    for layer, (kern, ms, macs) in a["prof"].items():
        kb, msb, _ = b["prof"][layer]
        if "hg." in layer and ("pglds" in kern or "prw" in kern):
            tot[0] += ms; tot[1] += msb
            print(f"{layer:14s} {kern:22s} {ms:7.3f} ms {2 * macs / ms / 1e9:7.1f} TF | {kb:22s} {msb:7.3f} ms {2 * macs / msb / 1e9:7.1f} TF  x{ms / msb:.3f}")
    print(f"HG 3x3 total: {tot[0]:.3f} -> {tot[1]:.3f} ms;  frame: {sum(v[1] for v in a['prof'].values()):.3f} -> {sum(v[1] for v in b['prof'].values()):.3f} ms")
