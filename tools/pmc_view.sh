#!/bin/bash
# SQ counters of the raster kernel(s) on a short in-phase rollout (all views zoomed in). Run through gpurun; prints per-view averages.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_view
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o p -- python tools/step_loop.py 70 4096 2 1 ${DBG:-0} > $OUT/run_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/b -o p -- python tools/step_loop.py 70 4096 2 1 ${DBG:-0} > $OUT/run_b.log 2>&1
python - <<'PY'
import glob, pandas as pd
views = 8192.0
for sub in ("a", "b"):
    fs = glob.glob(f"gpurun_out/pmc_view/{sub}/**/p_counter_collection.csv", recursive=True)
    if not fs: print("no counters in pass", sub); continue
    df = pd.read_csv(fs[0])
    df["K"] = df.Kernel_Name.str.extract(r"^(?:void )?([A-Za-z_0-9]+)")[0]
    for k in ("k_view", "k_view_setup", "k_view_draw"):
        d = df[df.K == k]
        if d.empty: continue
        piv = d.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
        piv = piv[piv.SQ_WAVE_CYCLES > 0.5 * piv.SQ_WAVE_CYCLES.max()]          # the main launches
        last = piv.tail(10).mean()
        print(f"== {k}: {len(piv.tail(10))} main launches averaged, per view:")
        for c in piv.columns:
            print(f"   {c:>24}: {last[c]/views:10.1f}" + (f"   share of wave cycles {last[c]/last['SQ_WAVE_CYCLES']:.3f}" if c.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""))
PY
