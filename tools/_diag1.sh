set -x
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --actions drive > $O/bench_drive.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --agents 8 > $O/bench_n8.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null
DRIVE=1 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > $O/dyn_phases_drive.txt
N=8 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > $O/dyn_phases_n8.txt
