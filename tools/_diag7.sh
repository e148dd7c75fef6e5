export TMPDIR=/tmp
O=gpurun_out/${TAG:-r05j}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "contacts or pile_ups or n8 or side_stream or stream_ordering or touching or rccl" 2>&1 | tail -5 > $O/pytest.txt
for cfg in "" "--actions drive" "--agents 8" "--agents 4"; do t=$(echo $cfg | tr -d ' -'); timeout 300 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value']/1e6,3), d['ms_per_step'], d['roofline']['frac'])" >> $O/ab.txt; done
