set -x
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
DRIVE=1 python tools/contact_stats.py 2>&1 | grep -v amdgpu.ids > $O/contact_stats_drive.txt
N=8 python tools/contact_stats.py 2>&1 | grep -v amdgpu.ids > $O/contact_stats_n8.txt
MCR_EXTRA_CFLAGS=-DMCR_POSLOOP_PROFILE python -m multi_car_racing_amd.build --force > /dev/null 2>&1
for d in 0 1; do for v in 0 1; do
  DRIVE=$d N=$((d?2:8)) VEL=$v python tools/posloop_profile.py 2>&1 | grep -v amdgpu.ids >> $O/posloop.txt
done; done
