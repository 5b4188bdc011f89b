#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, TCC has 4 slots) into
profiles/pmc_traffic.json: HBM bytes per launch and kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE counts 128-byte requests of a wide coalesced stream at 64 B -> doubled; both counters are in KiB).
usage: pmc_to_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <note>"""
import collections, csv, hashlib, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "hdr-realtime-video-pipeline_amd", "csrc")


def stamps():
    """What the figures were measured on: the library's build id (hdrtv_version()) and a hash per source file."""
    sys.path.insert(0, os.path.join(REPO, "hdr-realtime-video-pipeline_amd"))
    try:
        from hdrtv_mi355x import lib
        build = lib.build_id()
    except Exception as exc:  # noqa: BLE001
        build = f"unknown ({exc})"
    src = {f: hashlib.sha1(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:12] for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))}
    return build, src


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
build_id, sources = stamps()
out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "build_id": build_id, "sources": sources, "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes, mean per launch",
       "kernels": {}}
for k in sorted(set(f) & set(w)):
    out["kernels"][k] = {"fetch_kib": round(f[k][0], 1), "write_kib": round(w[k][0], 1), "launches_sampled": f[k][1],
                         "hbm_bytes_per_launch": round((2 * f[k][0] + w[k][0]) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"].get("conv_glds_kernel<3>", {})))
