#!/bin/bash
# Debug aid: PMC passes over scripts/time_align.py (run through gpurun); prints the align kernels' counters.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/pmca
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 scripts/time_align.py x > $OUT/t.txt 2>&1 || { tail -3 $OUT/t.txt; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 scripts/time_align.py x > $OUT/p_$tag.txt 2>&1 || echo "pass $tag failed"
done
python3 scripts/summarize_prof.py $OUT /tmp/pmca_sum | grep -A12 -E "align_traceback|calls" | grep -v "^--" | cut -c1-150
