#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, TCC has 4 slots) into
profiles/pmc_traffic.json: HBM bytes per launch and kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE counts 128-byte requests of a wide coalesced stream at 64 B -> doubled; both counters are in KiB).
usage: pmc_to_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <note>"""
import collections, csv, json, re, sys

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
out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes, mean per launch",
       "kernels": {}}
for k in sorted(set(f) & set(w)):
    out["kernels"][k] = {"fetch_kib": round(f[k][0], 1), "write_kib": round(w[k][0], 1), "launches_sampled": f[k][1],
                         "hbm_bytes_per_launch": round((2 * f[k][0] + w[k][0]) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"].get("conv_glds_kernel<3>", {})))
