"""Print the kernel timeline (start/end relative to the step's first kernel, queue id) of a few steady-state steps from a
rocprofv3 --kernel-trace CSV.  usage: step_timeline.py <kernel_trace.csv> [first_step_from_end] [nsteps]"""
import sys, pandas as pd
kt = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
def kname(s):
    for k in ("k_collide", "k_dynamics", "k_flags_viewprep", "k_viewprep", "k_view", "k_post", "k_await", "k_flags_list", "k_flags", "k_list_chain", "k_reset_list", "k_synth", "k_install", "copyBuffer", "fillBuffer"):
        if k in s: return k
    return s[:30]
kt["K"] = kt["Kernel_Name"].map(kname)
# a step starts at each k_collide launch that does not follow a k_install (that one belongs to reset() / reset_envs())
col = kt.index[(kt.K == "k_collide")].tolist()
starts = [i for i in col if i == 0 or kt.K[i - 1] != "k_install"]
sel = starts[-back:-back + ns + 1]
qcol = "Queue_Id" if "Queue_Id" in kt.columns else None
for a, b in zip(sel[:-1], sel[1:]):
    t0 = kt.Start_Timestamp[a]
    print(f"--- step (period {(kt.Start_Timestamp[b] - t0) / 1e3:.1f} us)")
    for i in range(a, b):
        r = kt.iloc[i]
        print(f"  {r.K:12s} q={r[qcol] if qcol else '?'}  start {(r.Start_Timestamp - t0) / 1e3:8.1f}  end {(r.End_Timestamp - t0) / 1e3:8.1f}  dur {(r.End_Timestamp - r.Start_Timestamp) / 1e3:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
