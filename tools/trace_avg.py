"""Average duration of each kernel's launches in a rocprofv3 --kernel-trace CSV, split into 'main' launches (longer than
`thr` us) and the rest.  usage: trace_avg.py <kernel_trace.csv> [thr_us] [last_n_launches]"""
import sys, pandas as pd
kt = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp")
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kt["dur"] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
kt["K"] = kt.Kernel_Name.str.extract(r"^(?:void )?([A-Za-z_0-9:]+)")[0]
for k, g in kt.groupby("K"):
    if not k.startswith("k_"): continue
    if last: g = g.tail(last)
    big, small = g[g.dur > thr], g[g.dur <= thr]
    print(f"{k:16s} main n={len(big):5d} avg {big.dur.mean():8.1f} us  med {big.dur.median():8.1f} | other n={len(small):5d} avg {small.dur.mean():6.1f} us")
