"""Loops of a kernel in `hipcc -S` output (back edges to earlier labels) with their instruction mix.  Not a test.
usage: isa_loops.py file.s mangled-symbol-substring [min_instructions]"""
import sys, re, collections
path, sym = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ins = []; labels = {}; inside = False; loc = None
for ln in open(path):
    if re.match(r"^_Z\w*:", ln): inside = sym in ln; continue
    if not inside: continue
    t = ln.strip()
    if t.startswith(".Lfunc_end"): inside = False; continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m: loc = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m: labels[m.group(1)] = len(ins); continue
    if not t or t.startswith((".", ";")) or t.endswith(":"): continue
    ins.append((t, loc))
def kind(op):
    return ("acc" if "accvgpr" in op else "scratch" if op.startswith("scratch") else "readlane" if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) else
            "f64" if op.startswith("v_") and "f64" in op else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "waitcnt" if op.startswith("s_waitcnt") else "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_") else "vmem")
loops = []
for i, (t, loc) in enumerate(ins):
    op = t.split()[0]
    if op.startswith(("s_cbranch", "s_branch")):
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] <= i: loops.append((labels[tgt], i, tgt))
print("instructions", len(ins), "loops", len(loops))
for a, b, tgt in sorted(loops):
    n = b - a + 1
    if n < mn: continue
    c = collections.Counter(kind(t.split()[0]) for t, _ in ins[a:b + 1])
    lines = collections.Counter(l for _, l in ins[a:b + 1] if l)
    top = ", ".join(f"{f}:{l}x{m}" for (f, l), m in lines.most_common(6))
    print(f"{tgt:>12} [{a:6d},{b:6d}] n={n:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "   lines " + top)
# innermost loops that touch memory or reload spilled scalars (the things that do not belong into a latency-bound sweep)
print("\ninnermost loops with memory traffic or scalar reloads:")
S = sorted(set((a, b, t) for a, b, t in loops))
for a, b, tgt in S:
    if any(a <= a2 and b2 <= b and (a2, b2) != (a, b) for a2, b2, _ in S): continue
    body = ins[a:b + 1]
    c = collections.Counter(kind(t.split()[0]) for t, _ in body)
    rel = sum(1 for t, _ in body if re.match(r"v_readlane_b32 s\d+, v\d+, \d+", t))
    sl = sum(1 for t, _ in body if t.startswith("s_load"))
    if c["vmem"] or c["scratch"] or rel or sl:
        lines = collections.Counter(l for _, l in body if l)
        print(f"{tgt:>12} n={b - a + 1:5d} vmem={c['vmem']} s_load={sl} scratch={c['scratch']} sgpr_reloads={rel} acc={c['acc']}  lines " + ", ".join(f"{f}:{l}x{m}" for (f, l), m in lines.most_common(4)))
