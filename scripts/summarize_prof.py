#!/usr/bin/env python3
"""Boils rocprofv3 CSV output down to the library's own kernels (names containing 'wamd').
usage: summarize_prof.py <raw dir> <out dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

raw, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
summary = {"kernel_stats": [], "trace": {}, "pmc": {}}

for f in glob.glob(os.path.join(raw, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "wamd" in r.get("Name", "")]
    summary["kernel_stats"] = keep
    with open(os.path.join(out, "kernel_stats.csv"), "w") as o:
        if rows:
            w = csv.DictWriter(o, fieldnames=rows[0].keys())
            w.writeheader()
            for r in rows[:15]:
                w.writerow(r)
            for r in keep:
                if r not in rows[:15]:
                    w.writerow(r)

for f in glob.glob(os.path.join(raw, "**", "*kernel_trace.csv"), recursive=True):
    per = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "wamd" not in name:
            continue
        per[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                            "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for name, d in per.items():
        summary["trace"][name] = dict(calls=len(d), avg_ns=sum(d) / len(d), min_ns=min(d), max_ns=max(d), **meta[name])

for f in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "wamd" not in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        for c, v in cs.items():
            summary["pmc"].setdefault(name, {})[c] = dict(dispatches=len(v), mean=sum(v) / len(v), max=max(v))

json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for name, t in summary["trace"].items():
    print("%-100s calls %4d avg %.3f ms" % (name[:100], t["calls"], t["avg_ns"] / 1e6))
for name, cs in summary["pmc"].items():
    print(name[:100])
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.6g over %d dispatches" % (c, v["mean"], v["dispatches"]))
