#!/bin/bash
# Debug aid: per-dispatch SQ counters of scripts/time_softclip.py (three kinds x three repetitions).
set -e
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sc
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out -o p --output-format csv -- python3 scripts/time_softclip.py
python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_sc")
f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "softclip" not in r["Kernel_Name"]:
        continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, c in rows.items():
    print(d, " ".join("%s=%.3g" % (k.replace("SQ_", ""), v / 262144) for k, v in sorted(c.items())))
PY
