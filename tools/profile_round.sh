#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun).  Outputs land in gpurun_out/prof/ and are
# summarised into profiles/ by tools/summarise_profiles.py (run afterwards, on either side).
set -x
export TMPDIR=/tmp
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
CMD="python bench.py --steps 200 --warmup 64 --no-cpu-baseline"
$CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2> $OUT/pmc_sq.err
ls -R $OUT | head -40
