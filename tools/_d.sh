export TMPDIR=/tmp
O=gpurun_out/${TAG:-r06n}; mkdir -p $O
for cfg in "--actions drive" "" "--agents 8"; do t=$(echo $cfg | tr -d ' -'); timeout 300 python bench.py --no-cpu-baseline $cfg 2> $O/err_$t.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value']/1e6,3), d['ms_per_step'], d['roofline']['frac'], d['config']['touch_verdict_mismatches_rank0'])" >> $O/ab.txt; done
timeout 2400 python -m pytest tests -m gpu -q -x -k "contacts or pile_ups or n8 or stream_ordering or freeze or two_handles or foreign" 2>&1 | tail -4 > $O/pytest.txt
