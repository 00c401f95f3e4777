#!/usr/bin/env python3
"""tools/kernel_asm.py SRC.hip MANGLED_PREFIX [-- extra hipcc flags]: compile one translation unit of csrc/ to gfx950 assembly with the
library's flags, cut out the first kernel whose mangled name starts with the prefix -> /tmp/kernel.s, and print its register / scratch
figures and, per basic block holding matrix instructions, the instruction mix (MFMA, VALU, LDS, global).  CPU-side (no GPU needed)."""
import os, re, subprocess, sys

def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--"); extra = args[k + 1:]; args = args[:k]
    src, prefix = args[0], args[1]
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "details-in-recommendation_amd", "csrc")
    out = "/tmp/%s.s" % os.path.basename(src)
    cmd = ["/opt/rocm/bin/hipcc", "-S", "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off"] + extra + [os.path.join(here, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:]); sys.exit(1)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    meta = next(i for i in range(end, len(lines)) if "ScratchSize" in lines[i])
    body = lines[start:end + 1]
    open("/tmp/kernel.s", "w").write("\n".join(body))
    for l in lines[end:meta + 12]:
        if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize", l): print(l.strip())
    blocks, cur = [], None
    for i, l in enumerate(body):
        if re.match(r"^(\.LBB|; %bb)", l):
            cur = {"name": l.split()[0] if l.startswith(".") else l.split()[1], "line": i + 1, "mfma": 0, "valu": 0, "lds": 0, "glob": 0, "scratch": 0}
            blocks.append(cur)
        elif cur is not None:
            t = l.strip()
            if t.startswith("v_mfma"): cur["mfma"] += 1
            elif t.startswith("v_"): cur["valu"] += 1
            elif t.startswith("ds_"): cur["lds"] += 1
            elif t.startswith("scratch_"): cur["scratch"] += 1
            elif t.startswith(("global_", "buffer_")): cur["glob"] += 1
    print("blocks with matrix instructions (line in /tmp/kernel.s):")
    for b in blocks:
        if b["mfma"]: print("  ", b)

if __name__ == "__main__":
    main()
