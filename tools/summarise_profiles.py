"""Summarise gpurun_out/prof_<round>/ into profiles/<round>_*.md|json (tracked)."""
import json, os, sys
import pandas as pd

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{R}"
os.makedirs("profiles", exist_ok=True)
lines = [f"# {R}: rocprofv3 summary of `python bench.py --no-cpu-baseline` (default: 1000 timed steps after 64 warm-up steps and the steady-state pre-roll) (B=4096, N=2, 1x MI355X)\n"]
try:
    b = json.loads(open(f"{src}/bench_plain.json").read().strip().splitlines()[-1])
    lines.append("Un-profiled bench line of the same command:\n\n```json\n" + json.dumps(b) + "\n```\n")
except Exception as e:
    lines.append(f"(bench_plain missing: {e})\n")

def kname(s):
    for k in ("k_collide", "k_dynamics", "k_view", "k_install", "k_positions", "k_sincos"):
        if k in s: return k
    return s[:60]

st = pd.read_csv(f"{src}/stats/s_kernel_stats.csv")
st["Kernel"] = st["Name"].map(kname)
lines.append("## `--kernel-trace --stats` (kernel_stats.csv)\n")
lines.append(st[["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]].to_markdown(index=False) + "\n")
kt = pd.read_csv(f"{src}/stats/s_kernel_trace.csv")
kt["Kernel"] = kt["Kernel_Name"].map(kname); kt["us"] = (kt["End_Timestamp"] - kt["Start_Timestamp"]) / 1e3
# timed region = last 1000 steps: split the two collide/dynamics passes by order within a step
v = kt[kt.Kernel == "k_view"].tail(1000)
lines.append(f"\nk_view over the timed region (last 1000 launches): mean {v.us.mean():.1f} us, median {v.us.median():.1f} us, min {v.us.min():.1f}, max {v.us.max():.1f}\n")
for k in ("k_collide", "k_dynamics"):
    g = kt[kt.Kernel == k].tail(2000)
    p0, p1 = g.iloc[0::2], g.iloc[1::2]
    lines.append(f"{k}: step pass mean {max(p0.us.mean(), p1.us.mean()):.1f} us, auto-reset pass mean {min(p0.us.mean(), p1.us.mean()):.1f} us\n")
traffic = {}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    c = pd.read_csv(f"{src}/{name}/p_counter_collection.csv")
    c["Kernel"] = c["Kernel_Name"].map(kname)
    g = c[(c.Kernel == "k_view") & (c.Counter_Name == ctr)].tail(1000)
    traffic[ctr] = float(g.Counter_Value.mean())
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads exactly 1/2 of
# the bytes of a wide coalesced stream -> x2; WRITE_SIZE is taken as reported (uncalibrated).
fetch_b = traffic["FETCH_SIZE"] * 1024 * 2
write_b = traffic["WRITE_SIZE"] * 1024
lines.append("\n## HBM traffic of k_view (PMC, separate passes)\n")
lines.append(f"FETCH_SIZE mean {traffic['FETCH_SIZE']:.0f} KB/launch (x2 gfx950 correction -> {fetch_b/1e6:.1f} MB), WRITE_SIZE mean {traffic['WRITE_SIZE']:.0f} KB/launch ({write_b/1e6:.1f} MB)\n")
alg = b["roofline"]["algorithmic_bytes_per_launch"] if "roofline" in b and b["roofline"] else None
lines.append(f"HBM bytes per launch ~ {(fetch_b+write_b)/1e6:.1f} MB vs algorithmic {alg/1e6 if alg else float('nan'):.1f} MB\n")
json.dump({"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
           "raw_kb": traffic, "round": R, "note": "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported"},
          open("profiles/view_traffic.json", "w"))
sq = pd.read_csv(f"{src}/pmc_sq/p_counter_collection.csv"); sq["Kernel"] = sq["Kernel_Name"].map(kname)
piv = sq.pivot_table(index=["Dispatch_Id", "Kernel"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
lines.append("\n## SQ counters per launch (mean over the last 1000 launches of each kernel)\n")
rows = []
for k, g in piv.groupby("Kernel"):
    g = g.sort_values("Dispatch_Id").tail(1000); m = g.drop(columns=["Dispatch_Id", "Kernel"]).mean()
    m["Kernel"] = k; rows.append(m)
lines.append(pd.DataFrame(rows).set_index("Kernel").round(0).to_markdown() + "\n")
open(f"profiles/{R}_rocprof_summary.md", "w").write("\n".join(lines))
print("\n".join(lines))
