"""Sum rocprofv3 counter_collection.csv per kernel: python scripts/pmc_csv.py file.csv [substring]"""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    if pat in r["Kernel_Name"]:
        kn = r["Kernel_Name"][:60]
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[kn].add(r["Dispatch_Id"])
for k, v in agg.items():
    print(k, "dispatches", len(disp[k]))
    for c, x in sorted(v.items()):
        print("   %-24s %.5g per dispatch" % (c, x / len(disp[k])))
