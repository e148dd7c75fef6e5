"""Static instruction counts of a kernel per source line range, from `hipcc -S -gline-tables-only` output.  Not a test.
usage: isa_by_line.py file.s mangled-symbol-substring [file-substring-of-the-source]  -> per-line VALU / SALU / LDS / VMEM counts"""
import sys, re, collections
path, sym = sys.argv[1], sys.argv[2]
src = sys.argv[3] if len(sys.argv) > 3 else None
files = {}
cnt = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
inside = False; cur = None
for ln in open(path):
    m = re.match(r"\s*\.file\s+(\d+)\s+(\".*\")", ln)
    if m: files[int(m.group(1))] = m.group(2); continue
    if re.match(r"^_Z\w*:", ln): inside = sym in ln; continue
    if not inside: continue
    if ln.strip().startswith(".Lfunc_end"): inside = False; continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    t = ln.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"): continue
    op = t.split()[0]
    k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3 if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else 4
    cnt[cur][k] += 1
rows = sorted(cnt.items())
tot = [0] * 5
for (f, l), c in rows:
    fn = files.get(f, "?")
    if src and src not in fn: 
        for i in range(5): tot[i] += c[i]
        print(f"{fn[-30:]:>30}:{l:<5} valu {c[0]:5d} salu {c[1]:5d} lds {c[2]:4d} vmem {c[3]:4d} other {c[4]:3d}"); continue
    for i in range(5): tot[i] += c[i]
    print(f"{fn[-30:]:>30}:{l:<5} valu {c[0]:5d} salu {c[1]:5d} lds {c[2]:4d} vmem {c[3]:4d} other {c[4]:3d}")
print("total", tot)
