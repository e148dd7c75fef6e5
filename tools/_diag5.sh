export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
run() { tag=$1; shift; MCR_HIP_CFLAGS="$*" python -m multi_car_racing_amd.build --force > $O/build_$tag.log 2>&1 || { echo "build failed $tag"; return; }
  for cfg in "--actions drive" "--agents 8"; do t=$(echo $cfg | tr -d ' -'); timeout 300 python bench.py --no-cpu-baseline --steps 600 $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$t', round(d['value']/1e6,3), d['ms_per_step'])" >> $O/ab.txt; done
  echo "== $tag" >> $O/dyn.txt; DRIVE=1 python tools/dyn_phases.py 2>&1 | grep -A5 "side stream" >> $O/dyn.txt; }
run noslp -fno-slp-vectorize
run slp12 -mllvm -slp-threshold=12
run base
