#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun).  Outputs land in gpurun_out/prof_<round>/ and are
# summarised into profiles/ by tools/summarise_profiles.py (run on the box: the raw traces are too large to travel back).
set -x
export TMPDIR=/tmp
R=${1:-r01}
OUT=gpurun_out/prof_$R
mkdir -p $OUT gpurun_out/profiles_$R
CMD="timeout 400 python bench.py --no-cpu-baseline"     # default workload/steps; the CPU baseline leg is not profiled
$CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > /dev/null 2> $OUT/pmc_write.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2> $OUT/pmc_sq.err
ls -R $OUT | head -40
tail -3 $OUT/*.err
python tools/summarise_profiles.py $R > $OUT/summary.log 2>&1
cp profiles/${R}_rocprof_summary.md profiles/view_traffic.json gpurun_out/profiles_$R/ 2>/dev/null
cp $OUT/stats/s_kernel_stats.csv gpurun_out/profiles_$R/${R}_kernel_stats.csv 2>/dev/null
cp $OUT/bench_plain.json $OUT/bench_stats.json $OUT/summary.log gpurun_out/profiles_$R/ 2>/dev/null
# a few steps of the kernel timeline (which chain is critical)
f=$(find $OUT/stats -name "s_kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 300 3 > gpurun_out/profiles_$R/${R}_step_timeline.txt 2>&1
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
# in-kernel phase clocks of the SHIPPED raster and dynamics kernels, raster ablations, SQ counters of the raster alone
python tools/view_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_$R/${R}_view_phases.txt
python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_$R/${R}_dyn_phases.txt
DRIVE=1 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_$R/${R}_dyn_phases_drive.txt
N=8 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_$R/${R}_dyn_phases_n8.txt
for n in 2 8; do N=$n python tools/collide_phases.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/profiles_$R/${R}_collide_phases.txt; done
STREAMS=1 python tools/ablate_view.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_$R/${R}_view_ablation.txt
bash tools/pmc_view.sh 2>&1 | grep -v amdgpu.ids | tail -24 > gpurun_out/profiles_$R/${R}_view_counters.txt
rm -rf gpurun_out/pmc_view
# the other configurations of the DESIGN table (one bench line each)
for cfg in "--streams 1" "--stagger 0" "--obs 0" "--agents 8" "--agents 1" "--agents 4" "--emulate-world 8" "--actions drive" "--envs 16384" "--envs 32768" "--rccl" "--terminal-obs 1"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 400 python bench.py --no-cpu-baseline $cfg > gpurun_out/profiles_$R/bench_$tag.json 2> /dev/null
done
# the default command as the driver runs it (with the CPU baseline legs: BASELINE.json configs[1] sample and configs[0])
timeout 600 python bench.py > gpurun_out/profiles_$R/bench_default.json 2> /dev/null
ls -la gpurun_out/profiles_$R
