export TMPDIR=/tmp
O=gpurun_out/${TAG:-r05x}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest.txt
for cfg in "" "--actions drive" "--agents 8"; do t=$(echo $cfg | tr -d ' -'); timeout 300 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value']/1e6,3), d['ms_per_step'], d['roofline']['frac'], d['config']['host_cores_busy_rank0'])" >> $O/ab.txt; done
