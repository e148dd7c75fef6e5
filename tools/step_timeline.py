"""Print the kernel timeline (start/end relative to the step's first kernel, queue id) of a few steady-state steps from a
rocprofv3 --kernel-trace CSV.  usage: step_timeline.py <kernel_trace.csv> [first_step_from_end] [nsteps]"""
import sys, pandas as pd
kt = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
def kname(s):
    for k in ("k_collide", "k_dynamics", "k_view", "k_flags", "k_synth", "k_install", "copyBuffer", "fillBuffer"):
        if k in s: return k
    return s[:30]
kt["K"] = kt["Kernel_Name"].map(kname)
# a step starts at each k_collide launched with the largest grid right after a fill (memset) or a view; find pass-0 collides:
col = kt.index[(kt.K == "k_collide")].tolist()
starts = [i for j, i in enumerate(col) if j == 0 or kt.K[col[j - 1]:i].isin(["k_synth"]).any()]
sel = starts[-back:-back + ns + 1]
qcol = "Queue_Id" if "Queue_Id" in kt.columns else None
for a, b in zip(sel[:-1], sel[1:]):
    t0 = kt.Start_Timestamp[a]
    print(f"--- step (period {(kt.Start_Timestamp[b] - t0) / 1e3:.1f} us)")
    for i in range(a, b):
        r = kt.iloc[i]
        print(f"  {r.K:12s} q={r[qcol] if qcol else '?'}  start {(r.Start_Timestamp - t0) / 1e3:8.1f}  end {(r.End_Timestamp - t0) / 1e3:8.1f}  dur {(r.End_Timestamp - r.Start_Timestamp) / 1e3:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
