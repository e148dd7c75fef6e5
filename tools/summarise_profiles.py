"""Summarise gpurun_out/prof_<round>/ into profiles/<round>_*.md|json (tracked).

A step of the default configuration issues several launches of the same kernel (main chain, contact side stream,
resume stream), so launches are labelled by queue and position in the step: on the caller's queue dynamics, bookkeeping,
view; on the side queue collide, chain(contact envs), view, reset pass(re-spawned envs), view; on the third queue
chain(resume), view.  (Counter passes: the contact pass runs first on the caller's queue, see label().)"""
import json, os, sys
import pandas as pd

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{R}"
os.makedirs("profiles", exist_ok=True)
STEPS = 1000
lines = [f"# {R}: rocprofv3 summary of `python bench.py --no-cpu-baseline` (default configuration: contact side stream on, 1000 timed steps after 64 warm-up steps and the steady-state pre-roll; B=4096, N=2, 1x MI355X)\n"]
try:
    b = json.loads(open(f"{src}/bench_plain.json").read().strip().splitlines()[-1])
    lines.append("Un-profiled bench line of the same command:\n\n```json\n" + json.dumps(b) + "\n```\n")
except Exception as e:
    b = {}
    lines.append(f"(bench_plain missing: {e})\n")

KERNELS = ("k_collide", "k_dynamics", "k_flags_viewprep", "k_viewprep", "k_view", "k_post", "k_await", "k_flags_list", "k_flags", "k_list_chain", "k_reset_list", "k_install", "k_touch", "k_step_begin", "fillBuffer", "copyBuffer")
def kname(s):
    for k in KERNELS:
        if k in s: return k
    return s[:60]

# launches of a step by queue, in launch order (mcr_hip.hip: launch_step, round 3; soft_sync: the streams meet through phase words,
# k_await / k_post are its one-thread kernels — on the event path, which counter-collecting runs take, they do not appear):
#   caller's queue : [collide, when the contact pass runs in front] dynamics (main envs) -> chain (resume + reset pass of the re-spawned envs) [-> bookkeeping] -> view (both lists; ends with the step's join)
#   side queue     : await(begin) -> [collide, beside the dynamics] -> chain (contact envs) [-> bookkeeping] -> view -> post(side done)
#   third queue    : await(dynamics, collide) -> view records + bookkeeping (k_flags_viewprep, N <= 2; k_flags beyond) -> view (main envs) -> post(main done)
CALLER = {"k_collide": ["collide (all envs)"], "k_dynamics": ["dynamics (main envs)"], "k_list_chain": ["chain (resume of deferred envs, caller's stream)"],
          "k_flags_list": ["bookkeeping (deferred envs)"], "k_view": ["view (deferred + re-spawned envs, caller's stream; waits for the join)"], "k_await": ["join (caller's stream, steps without frames)"]}
SIDE = {"k_collide": ["collide (all envs)"], "k_list_chain": ["chain (contact envs, side stream)"], "k_flags_list": ["bookkeeping (contact envs)"],
        "k_reset_list": ["reset pass (re-spawned envs, side stream)"], "k_view": ["view (contact envs, side stream)", "view (re-spawned envs, side stream)"],
        "k_await": ["await: step begun (side stream; spins from the end of its last step)"] * 2, "k_post": ["post: side stream done"]}
AWAIT3 = "await: dynamics + contact pass done (third stream; spins through the dynamics)"
THIRD = {"k_viewprep": ["view records + car polygons (main envs, third stream)"], "k_flags": ["bookkeeping (main envs, third stream)"], "k_view": ["view (main envs)"],
         "k_flags_viewprep": ["view records + bookkeeping (main envs, third stream)"], "k_await": [AWAIT3, AWAIT3], "k_post": ["post: third stream done"]}
STEP_KERNELS = ("k_collide", "k_dynamics", "k_flags_viewprep", "k_viewprep", "k_view", "k_post", "k_await", "k_flags", "k_flags_list", "k_list_chain", "k_reset_list")

def label(df, order_col):
    """adds column Label for the launches of the last STEPS steps.  A step starts with its contact pass (the k_collide launch that
    does not follow a k_install: that one belongs to reset() / reset_envs()) — the first launch of a step on either queue —
    and every launch up to the next such k_collide belongs to it.  Queues: the caller's carries k_dynamics, the side queue
    k_reset_list, the third one the main envs' k_flags."""
    df = df.sort_values(order_col).reset_index(drop=True)
    df["K"] = df["Kernel_Name"].map(kname)
    df["Label"] = None
    dq = df[df.K == "k_dynamics"].Queue_Id.value_counts()
    fq = df[df.K.isin(["k_flags", "k_flags_viewprep"])].Queue_Id.value_counts()
    rq = df[df.K == "k_reset_list"].Queue_Id.value_counts()
    if dq.empty:
        return df
    main_q = dq.index[0]
    third_q = fq.index[0] if len(fq) else None
    side_q = rq.index[0] if len(rq) else None
    if third_q == main_q:                                   # serialised / single-stream runs: the main envs' kernels share the caller's queue
        third_q = None
    col = df.index[df.K == "k_collide"].tolist()
    starts = [i for i in col if i == 0 or df.at[i - 1, "K"] != "k_install"]
    if len(starts) < STEPS + 1:
        return df
    starts = starts[-(STEPS + 1):]
    for a, e in zip(starts[:-1], starts[1:]):
        seen = {}
        for i in range(a, e):
            k = df.at[i, "K"]
            if k not in STEP_KERNELS: continue
            q = df.at[i, "Queue_Id"]
            key = (k, q); n = seen.get(key, 0); seen[key] = n + 1
            if q == main_q:
                names = dict(CALLER)
                if third_q is None: names.update({"k_viewprep": THIRD["k_viewprep"], "k_flags": THIRD["k_flags"], "k_flags_viewprep": THIRD["k_flags_viewprep"], "k_view": THIRD["k_view"] + CALLER["k_view"]})
                names = names.get(k, [])
            elif q == third_q: names = THIRD.get(k, [])
            else: names = SIDE.get(k, [])
            df.at[i, "Label"] = names[n] if n < len(names) else f"{k} #{n} (queue {q})"
    return df

st = pd.read_csv(f"{src}/stats/s_kernel_stats.csv")
st["Kernel"] = st["Name"].map(kname)
lines.append("## `--kernel-trace --stats` (kernel_stats.csv, whole process incl. pre-roll; every launch of a kernel pooled; k_await = the one-wavefront kernels that WAIT for another stream's phase word — their duration is waiting time, not work)\n")
lines.append(st[["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]].head(11).to_markdown(index=False) + "\n")
kt = label(pd.read_csv(f"{src}/stats/s_kernel_trace.csv"), "Start_Timestamp")
kt["us"] = (kt["End_Timestamp"] - kt["Start_Timestamp"]) / 1e3
g = kt[kt.Label.notna()].groupby("Label").us.agg(["count", "mean", "median", "min", "max"]).round(1)
lines.append("## Kernel trace over the timed region (last 1000 steps), launches labelled by their role in the step\n")
lines.append(g.to_markdown() + "\n")
vmain = kt[kt.Label == "view (main envs)"].us
lines.append(f"\nDominant kernel for the roofline: `k_view (main envs)` mean {vmain.mean():.1f} us per launch (bench.py's HIP-event figure: {b.get('roofline', {}).get('avg_launch_ms', float('nan')) * 1e3:.1f} us)\n")

traffic = {}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    c = label(pd.read_csv(f"{src}/{name}/p_counter_collection.csv"), "Dispatch_Id")
    gsel = c[(c.Label == "view (main envs)") & (c.Counter_Name == ctr)]
    traffic[ctr] = float(gsel.Counter_Value.mean())
    side = c[(c.K == "k_view") & c.Label.notna() & (c.Label != "view (main envs)") & (c.Counter_Name == ctr)]     # the list launches of the raster (k_viewprep is a kernel of its own)
    traffic[ctr + "_side"] = float(side.Counter_Value.sum() / STEPS) if len(side) else 0.0
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads exactly 1/2 of
# the bytes of a wide coalesced stream -> x2; WRITE_SIZE is taken as reported (uncalibrated).
fetch_b = (traffic["FETCH_SIZE"] + traffic["FETCH_SIZE_side"]) * 1024 * 2
write_b = (traffic["WRITE_SIZE"] + traffic["WRITE_SIZE_side"]) * 1024
lines.append("\n## HBM traffic of the raster (PMC, separate passes; main launch + the list launches for the contact / deferred / re-spawned envs)\n")
lines.append(f"FETCH_SIZE mean {traffic['FETCH_SIZE']:.0f} + {traffic['FETCH_SIZE_side']:.0f} KB/step (x2 gfx950 correction -> {fetch_b/1e6:.1f} MB), WRITE_SIZE mean {traffic['WRITE_SIZE']:.0f} + {traffic['WRITE_SIZE_side']:.0f} KB/step ({write_b/1e6:.1f} MB)\n")
alg = b.get("roofline", {}).get("algorithmic_bytes_per_launch")
lines.append(f"HBM bytes per step ~ {(fetch_b+write_b)/1e6:.1f} MB vs algorithmic {alg/1e6 if alg else float('nan'):.1f} MB\n")
json.dump({"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
           "raw_kb": traffic, "round": R, "note": "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; main raster launch + the list launches (contact / deferred / re-spawned envs) of a step"},
          open("profiles/view_traffic.json", "w"))
sq = label(pd.read_csv(f"{src}/pmc_sq/p_counter_collection.csv"), "Dispatch_Id")
sq = sq[sq.Label.notna()]
# counters are summed over ALL launches of a kernel inside a step (main + side + reset passes): with several counters
# rocprofv3 serialises and may reorder the queues, so only the per-kernel totals are robust here
# (with more counters than the hardware has slots rocprofv3 rotates counter groups over the dispatches, so each
# counter is a SAMPLE of the launches: mean per sampled launch x launches of that kernel per step)
per_launch = sq.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
launches_per_step = kt[kt.Label.notna()].groupby("K").size() / STEPS
tot = per_launch.mul(launches_per_step, axis=0).dropna(how="all")
lines.append("\n## SQ counters per step, all launches of a kernel pooled (mean over the timed region)\n")
lines.append(tot.round(0).to_markdown() + "\n")
if "k_view" in tot.index:
    nviews = 4096 * 2
    lines.append(f"\nk_view per agent view: {tot.loc['k_view', 'SQ_INSTS_VALU'] / nviews:.0f} VALU, {tot.loc['k_view', 'SQ_INSTS_SALU'] / nviews:.0f} SALU, {tot.loc['k_view', 'SQ_INSTS_LDS'] / nviews:.0f} LDS instructions\n")
open(f"profiles/{R}_rocprof_summary.md", "w").write("\n".join(lines))
print("\n".join(lines))
