export TMPDIR=/tmp
O=gpurun_out/${TAG:-r05k}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_terminal_obs.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest_term.txt
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_terminal_obs.py 2>&1 | tail -8 > $O/pytest.txt
for cfg in "" "--terminal-obs 1" "--actions drive" "--agents 8" ""; do t=$(echo $cfg | tr -d ' -'); timeout 300 python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value']/1e6,3), d['ms_per_step'], d['roofline']['frac'])" >> $O/ab.txt; done
