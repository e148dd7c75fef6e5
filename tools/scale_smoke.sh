#!/bin/bash
# One command for the first multi-GPU lease: the scaling curve of BASELINE.json configs[1] -> configs[2]
# (4096 envs per GPU, weak scaling) at 1, 2, 4, 8 GPUs of this node, as far as devices exist, then the same 1-GPU run with
# the host share of an 8-rank job (--emulate-world 8).  Writes gpurun_out/scale_<N>.json (one bench line each).
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
STEPS=${1:-1000}
for N in 1 2 4 8; do
  if [ "$N" -le "$NDEV" ]; then
    timeout 900 python bench.py --gpus $N --steps $STEPS --no-cpu-baseline > gpurun_out/scale_$N.json 2> gpurun_out/scale_$N.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/scale_$N.json").read().strip().splitlines()[-1])
print("gpus %d: %.3f M env-steps/s, %.3f ms/step" % (d["n_gpus"], d["value"] / 1e6, d["ms_per_step"]))
PY
  else
    echo "gpus $N: skipped ($NDEV device(s) here)"
  fi
done
timeout 900 python bench.py --steps $STEPS --no-cpu-baseline --emulate-world 8 > gpurun_out/scale_emu8.json 2> gpurun_out/scale_emu8.err
python - <<PY
import json
d = json.loads(open("gpurun_out/scale_emu8.json").read().strip().splitlines()[-1])
c = d["config"]
print("1 gpu with the host share of an 8-rank job: %.3f M env-steps/s, blocked on refills %.3f s, frozen env-steps %d, %s" % (d["value"] / 1e6, c["step_blocked_on_refill_s_rank0"], c["env_steps_frozen_waiting_for_host_rank0"], c["emulated_host_share"]))
PY
