#!/usr/bin/env python3
"""Per-kernel wave-cycle breakdown from one rocprofv3 --pmc pass (counter_collection.csv):
fractions of SQ_WAVE_CYCLES parked (WAIT_ANY), issue-stalled (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY), MFMA pipe busy and the
effective clock.  usage: sq_breakdown.py <counter_collection.csv> <kernel_trace.csv> [filter]"""
import collections, csv, sys

ctr = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    ctr[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
dur = {r["Dispatch_Id"]: (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9 for r in csv.DictReader(open(sys.argv[2]))}
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d, c in ctr.items():
    if d not in dur:
        continue
    name = str(c["name"]).replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if flt and flt not in name:
        continue
    a = agg[name]
    for k, v in c.items():
        if k != "name":
            a[k] += v
    a["t"] += dur[d]; a["n"] += 1
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    parts = " ".join(f"{k[3:].lower()}={a[k] / wc:.3f}" for k in sorted(a) if k.startswith("SQ_") and k not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"))
    mf = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0
    print(f"{name[:40]:40s} n={int(a['n']):3d} avg={a['t'] / a['n'] * 1e3:7.3f} ms clk={cyc / a['t'] * 1e-9 if cyc else 0:5.2f} GHz mfma_busy={mf:.3f} | {parts}")
