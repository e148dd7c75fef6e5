set -x
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r05f}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "contacts or pile_ups or n8 or side_stream or stream_ordering or touching" 2>&1 | tail -8 > $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --actions drive > $O/bench_drive.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --agents 8 > $O/bench_n8.json 2>/dev/null
DRIVE=1 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > $O/dyn_phases_drive.txt
N=8 python tools/dyn_phases.py 2>&1 | grep -v amdgpu.ids > $O/dyn_phases_n8.txt
MCR_EXTRA_CFLAGS=-DMCR_POSLOOP_PROFILE python -m multi_car_racing_amd.build --force > /dev/null 2>&1
for d in 0 1; do for v in 0 1; do
  DRIVE=$d N=$((d?2:8)) VEL=$v python tools/posloop_profile.py 2>&1 | grep -v amdgpu.ids >> $O/posloop.txt
done; done
