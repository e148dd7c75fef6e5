#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun).  Outputs land in gpurun_out/prof/ and are
# summarised into profiles/ by tools/summarise_profiles.py (run afterwards, on either side).
set -x
export TMPDIR=/tmp
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
CMD="timeout 400 python bench.py --no-cpu-baseline"     # default workload/steps; the CPU baseline leg is not profiled
$CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > /dev/null 2> $OUT/pmc_write.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2> $OUT/pmc_sq.err
ls -R $OUT | head -40
tail -3 $OUT/*.err
# summarise on the box (the raw traces are too large to travel back); keep only the small files
python tools/summarise_profiles.py $R > $OUT/summary.log 2>&1
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_rocprof_summary.md profiles/view_traffic.json gpurun_out/profiles_$R/ 2>/dev/null
cp $OUT/stats/s_kernel_stats.csv gpurun_out/profiles_$R/${R}_kernel_stats.csv 2>/dev/null
cp $OUT/bench_plain.json $OUT/bench_stats.json $OUT/summary.log gpurun_out/profiles_$R/ 2>/dev/null
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
