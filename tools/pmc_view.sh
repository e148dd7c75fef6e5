#!/bin/bash
# SQ counters of the raster kernel on a short in-phase rollout (all views zoomed in). Run through gpurun; prints per-view averages.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_view
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p -- python tools/step_loop.py 70 > $OUT/run.log 2>&1
python - <<'PY'
import glob, pandas as pd
f = glob.glob("gpurun_out/pmc_view/**/p_counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df = df[df.Kernel_Name.str.contains("k_view")]
piv = df.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
piv = piv[piv.SQ_INSTS_VALU > 0.5 * piv.SQ_INSTS_VALU.max()]          # the main launches
last = piv.tail(10).mean()
views = 8192.0
print("main k_view launches averaged:", len(piv.tail(10)))
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    print(f"{k:>22}: {last[k]/views:9.0f} per view")
print(f"   SQ_WAVE_CYCLES/view: {4*last['SQ_WAVE_CYCLES']/views:9.0f} (x4: quad-cycles -> cycles), WAIT_ANY share {last['SQ_WAIT_ANY']/last['SQ_WAVE_CYCLES']:.2f}, ACTIVE_VALU share {last['SQ_ACTIVE_INST_VALU']/last['SQ_WAVE_CYCLES']:.2f}")
PY
