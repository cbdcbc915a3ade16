#!/usr/bin/env python3
"""Cuts rocprofv3 counter passes of `bench.py --profile-markers` into bench.py's timed regions.

usage: summarize_extras.py <raw dir with pmc_*/ and trace/ from scripts/profile_round2.sh> <bench line json> <out prefix>

The bench brackets every timed region with two float64 fill kernels of distinct sizes (bench.py: mark()).  Here the
FillFunctor<double> dispatches of a pass are ranked by grid size: rank 2i opens region i, rank 2i + 1 closes it, the
regions in the order the bench line's "profile_markers" lists them.  Per region and counter the library's kernels
(names containing "wamd") are summed and divided by the region's repetitions; per-unit figures follow:
  hbm_bytes_per_unit  = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / units   (FETCH_SIZE doubled: gfx950 tallies 128-byte
                        requests at 64 bytes - checked for scattered gathers with scripts/ubench/fetch_calib.hip)
  valu_insts_per_unit = SQ_INSTS_VALU * 64 / units                     (lane-ops: wave instructions x 64 lanes)
Writes <out prefix>_regions.json (everything) and <out prefix>_counters.json (what bench.py reads)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

raw, line_file, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
line = None
for ln in open(line_file):
    ln = ln.strip()
    if ln.startswith("{"):
        line = json.loads(ln)
pm = line["profile_markers"]
tags, reps, units = pm["tags"], pm["reps"], pm["units"]


def grid(r):   # counter files carry Grid_Size, trace files Grid_Size_X / _Y / _Z
    if r.get("Grid_Size"):
        return int(r["Grid_Size"])
    return int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)


def regions_of(rows, value_of):
    """rows in dispatch order -> {tag: {kernel: {key: sum}}}"""
    marks = sorted({grid(r) for r in rows if "FillFunctor<double>" in r["Kernel_Name"]})
    if len(marks) != 2 * len(tags):
        raise SystemExit("expected %d marker sizes, found %d" % (2 * len(tags), len(marks)))
    rank = {g: i for i, g in enumerate(marks)}
    cur = None
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for r in rows:
        name = r["Kernel_Name"]
        if "FillFunctor<double>" in name:
            i = rank[grid(r)]
            cur = tags[i // 2] if i % 2 == 0 else None
            continue
        if cur is None or "wamd" not in name:
            continue
        for k, v in value_of(r).items():
            acc[cur][name.split("(")[0]][k] += v
    return acc


out = {t: {"reps": reps[t], "units": units[t], "kernels": {}, "counters": {}} for t in tags}

for f in glob.glob(os.path.join(raw, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    acc = regions_of(rows, lambda r: {"ns": float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])), "calls": 1.0})
    for t, ks in acc.items():
        for k, v in ks.items():
            out[t]["kernels"][k] = {"calls_per_rep": v["calls"] / reps[t], "avg_us": v["ns"] / v["calls"] / 1e3,
                                    "us_per_rep": v["ns"] / reps[t] / 1e3}

for f in glob.glob(os.path.join(raw, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: (int(r["Dispatch_Id"]), r["Counter_Name"]))
    acc = regions_of(rows, lambda r: {r["Counter_Name"]: float(r["Counter_Value"])})
    for t, ks in acc.items():
        for k, v in ks.items():
            for c, x in v.items():
                out[t]["counters"].setdefault(k, {})[c] = x / reps[t]

small = {}
for t, e in out.items():
    tot = defaultdict(float)
    for k, cs in e["counters"].items():
        for c, x in cs.items():
            tot[c] += x
    e["total"] = dict(tot)
    u = e["units"] or 1
    s = {}
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        s["hbm_bytes_per_unit"] = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / u
    if "SQ_INSTS_VALU" in tot:
        s["valu_insts_per_unit"] = tot["SQ_INSTS_VALU"] * 64 / u
    if "SQ_WAVE_CYCLES" in tot and tot["SQ_WAVE_CYCLES"]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in tot:
                s[c.lower() + "_frac"] = tot[c] / tot["SQ_WAVE_CYCLES"]
    s["us_per_rep"] = sum(k["us_per_rep"] for k in e["kernels"].values())
    e["per_unit"] = s
    small[t] = s
json.dump(out, open(prefix + "_regions.json", "w"), indent=1)
json.dump(small, open(prefix + "_counters.json", "w"), indent=1)
for t in tags:
    s = small[t]
    print("%-26s %9.1f us  hbm %8.1f B/unit  valu %9.1f lane-ops/unit  wait_any %.2f  wait_inst %.2f" % (
        t, s.get("us_per_rep", 0), s.get("hbm_bytes_per_unit", float("nan")), s.get("valu_insts_per_unit", float("nan")),
        s.get("sq_wait_any_frac", float("nan")), s.get("sq_wait_inst_any_frac", float("nan"))))
