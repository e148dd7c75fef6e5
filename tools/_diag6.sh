export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
for cfg in "--agents 8" "--actions drive"; do t=$(echo $cfg | tr -d ' -'); rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o s -- python bench.py --no-cpu-baseline --steps 300 $cfg > $O/bench_$t.json 2> /tmp/kt.err
f=$(find /tmp/kt -name "s_kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 100 3 > $O/timeline_$t.txt 2>&1
python tools/trace_avg.py $f > $O/trace_avg_$t.txt 2>&1; done
