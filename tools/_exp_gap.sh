cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/gap
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 600 "$@" > gpurun_out/gap/b_$tag.json 2>gpurun_out/gap/b_$tag.err; python - <<P
import json; d=json.load(open('gpurun_out/gap/b_$tag.json')); print('$tag', round(d['value']/1e6,3), round(d['ms_per_step'],4), round((d.get('roofline') or {}).get('avg_launch_ms',0)*1e3,1))
P
}
run soft_1
run soft_2
MCR_SOFT_SYNC=0 run ev_1
run n4 --agents 4
MCR_SOFT_SYNC=0 run n4_ev --agents 4
run n8 --agents 8
MCR_SOFT_SYNC=0 run n8_ev --agents 8
run n1 --agents 1
run obs0 --obs 0
MCR_SOFT_SYNC=0 run obs0_ev --obs 0
run emu8 --emulate-world 8
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap/tr -o t -- python bench.py --no-cpu-baseline --steps 400 > /dev/null 2>&1
f=$(find gpurun_out/gap/tr -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 100 2 > gpurun_out/gap/timeline_soft.txt 2>&1
rm -rf gpurun_out/gap/tr
cat gpurun_out/gap/timeline_soft.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
