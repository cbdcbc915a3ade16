#!/bin/bash
# per-dispatch durations of the distance kernel's passes (debug aid; run through gpurun)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/pt
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/line.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'myers_distance' in r['Kernel_Name']:
        print(r['Kernel_Name'][:40], r.get('Grid_Size_X', r.get('Grid_Size')), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms')
PY
