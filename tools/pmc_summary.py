#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv, sys, collections
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    if flt and flt not in name: continue
    short = name.replace("void ", "").replace("(anonymous namespace)::", "")
    short = short.split("(")[0][:44]
    agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in d.values())
    print(f"{k:44s} n={n}")
    for c, v in sorted(d.items()):
        print(f"    {c:28s} mean={sum(v)/len(v):16.1f}")
