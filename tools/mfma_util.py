#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from one rocprofv3 pass
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py ...
(MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over the chip's 1024 SIMDs;
GRBM_GUI_ACTIVE is summed over the 8 XCDs; effective clock = GRBM_GUI_ACTIVE / 8 / wall time).
usage: mfma_util.py <counter_collection.csv> <kernel_trace.csv> [out.json]"""
import collections, csv, json, sys

ctr = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    ctr[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for d, c in ctr.items():
    if d not in dur:
        continue
    name = str(c["name"]).replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    a = agg[name]
    a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += c.get("GRBM_GUI_ACTIVE", 0.0); a[2] += dur[d]; a[3] += 1
out = {}
for name, (busy, gui, t, n) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    if t <= 0 or gui <= 0:
        continue
    cyc = gui / 8.0                       # shader cycles the kernel was resident, per XCD
    out[name] = {"launches": n, "avg_ms": round(t / n * 1e3, 4), "effective_clock_ghz": round(cyc / t * 1e-9, 3),
                 "mfma_busy_fraction": round(busy / (cyc * 1024.0), 4)}
    print(f"{name[:56]:56s} n={n:4d} avg={t / n * 1e3:7.3f} ms  clock={cyc / t * 1e-9:5.2f} GHz  MFMA busy={busy / (cyc * 1024.0) * 100:5.1f} %")
if len(sys.argv) > 3:
    json.dump({"note": "per kernel over all its launches; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
