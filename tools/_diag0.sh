set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
DRIVE=1 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05a/dyn_phases_drive.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o s -- python bench.py --no-cpu-baseline --actions drive --steps 300 > gpurun_out/r05a/bench_drive_traced.json 2> /tmp/kt.err
f=$(find /tmp/kt -name "s_kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 100 4 > gpurun_out/r05a/timeline_drive.txt 2>&1
python tools/trace_avg.py $f > gpurun_out/r05a/trace_avg_drive.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --actions drive > gpurun_out/r05a/bench_drive.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --agents 8 > gpurun_out/r05a/bench_n8.json 2>/dev/null
