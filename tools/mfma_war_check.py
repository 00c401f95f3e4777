"""tools/mfma_war_check.py -- static look at a gfx950 .s file: how close behind an MFMA does an asynchronous load (LDS / global / scratch)
overwrite one of that MFMA's source registers?  For every load destination that overlaps a source (A, B or C) of one of the previous
N MFMAs (N = 1, 2, 4, 8, 16, 32; straight-line text order inside the named kernel, labels ignored) one hit is counted per (kind of
source, kind of load, distance in MFMAs).  Usage: mfma_war_check.py file.s kernel-name-substring"""
import re, sys, collections
path, kname = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(kname), l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section") and i > start + 10)
def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m: return {int(m.group(1))}
    return set()
recent = collections.deque(maxlen=64)      # (A regs, B regs, C regs, D regs, line)
hits = collections.Counter()
examples = {}
n_mfma = 0
for i in range(start, end):
    l = lines[i].split(";")[0].strip()
    if not l: continue
    op = l.split()[0]
    args = [a.strip() for a in l[len(op):].split(",")]
    if op.startswith("v_mfma"):
        d, a, b, c = (regs(x) for x in args[:4])
        recent.append((a, b, c, d, i))
        n_mfma += 1
        continue
    kind = "lds" if op.startswith("ds_read") else "global" if op.startswith("global_load") else "scratch" if op.startswith("scratch_load") else None
    if kind is None: continue
    dst = regs(args[0])
    for dist, (a, b, c, d, li) in enumerate(reversed(recent), 1):
        for nm, s in (("A", a), ("B", b), ("C", c - d), ("D", d)):
            if dst & s:
                hits[(nm, kind, dist)] += 1
                examples.setdefault((nm, kind, dist), (li + 1, i + 1))
print("%s: %d MFMAs" % (path, n_mfma))
for nm in "ABCD":
    for kind in ("lds", "global", "scratch"):
        row = [sum(v for (a, k, d), v in hits.items() if a == nm and k == kind and d <= N) for N in (1, 2, 4, 8, 16, 32, 64)]
        if any(row): print("  src %s overwritten by %-7s load within 1/2/4/8/16/32/64 MFMAs: %s" % (nm, kind, row))
if "-v" in sys.argv:
    for k in sorted(examples, key=lambda k: k[2])[:12]: print("   e.g.", k, "lines", examples[k])
