#!/bin/bash
# Profiles the default bench command with rocprofv3 (run on the GPU box through gpurun).
# Kernel trace + stats first, then one PMC pass per counter group (the tool does not split
# counters over passes).  Summaries land in gpurun_out/prof_*; copy what matters to profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 $ARGS > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || echo "pass $tag failed"
done
python3 scripts/summarize_prof.py $OUT gpurun_out/prof_summary
for f in $OUT/*.json; do cp $f gpurun_out/prof_summary/; done
rm -rf $OUT
