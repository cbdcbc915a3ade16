#!/bin/bash
# Round-3 profile of the final tree (run on the GPU box through gpurun): the default bench command with the
# secondary shapes at the driver's own --steps 20 --warmup 5 (so the kernel-trace durations are those of the timed run), once under --kernel-trace --stats and once per counter group under --pmc (the tool does not
# split counters over passes; --pmc is never combined with another trace domain on this pool).  bench.py brackets
# its timed regions with marker kernels; scripts/summarize_extras.py cuts the passes into those regions.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof3
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary --profile-markers ${BENCH_ARGS:-}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  echo "pass $tag"
  timeout -s KILL 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 $ARGS > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || echo "pass $tag failed"
done
mkdir -p gpurun_out/prof3_summary
python3 scripts/summarize_extras.py $OUT $OUT/trace.json gpurun_out/prof3_summary/round3 || exit 1
python3 scripts/summarize_prof.py $OUT gpurun_out/prof3_summary > gpurun_out/prof3_summary/all_kernels.txt
cp $OUT/trace.json gpurun_out/prof3_summary/profiled_bench_line.json
rm -rf $OUT
