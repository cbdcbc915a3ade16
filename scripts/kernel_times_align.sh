#!/bin/bash
# per-kernel durations of scripts/time_align.py (debug aid; run through gpurun)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/kta
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 scripts/time_align.py prof > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    if 'wamd' not in k: continue
    k = k.split('(')[0][-60:]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    a = agg.setdefault(k, [0, 0.0, 1e9]); a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for k, (c, t, mn) in agg.items():
    print("%-62s calls %3d  avg %9.4f ms  min %9.4f ms" % (k, c, t / c, mn))
PY
