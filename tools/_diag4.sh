export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
run() { tag=$1; shift; MCR_HIP_CFLAGS="$*" python -m multi_car_racing_amd.build --force > $O/build_$tag.log 2>&1 || { echo "build failed $tag"; return; }
  for cfg in "" "--actions drive" "--agents 8"; do t=$(echo $cfg | tr -d ' -'); timeout 300 python bench.py --no-cpu-baseline --steps 600 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$t', round(d['value']/1e6,3), d['ms_per_step'])" >> $O/ab.txt; done; }
run base
run maxilp -mllvm -amdgpu-sched-strategy=max-ilp
run bias0 -mllvm -amdgpu-schedule-metric-bias=0
run base2
