#!/usr/bin/env python3
"""tools/vmcnt_scan.py: compile every csrc/*.hip to gfx950 assembly with the library's flags and list, per kernel, how many vector-memory LOADS are
waited for with `s_waitcnt vmcnt(0)` before the NEXT load is issued (a chain of round trips: the pattern behind NOTES R6.2 / R6.8 / R6.11 / R6.12).
Kernels with at least SCAN_MIN (default 6) such waits are listed.  A long chain matters where nothing else on the CU hides it (one workgroup per
CU: the towers, the packed DIN unit); bag_csr_k shows 15-23 and runs at the one-hot gather's row rate all the same (many resident waves).  CPU-side."""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "details-in-recommendation_amd")
sys.path.insert(0, here)
import build

def scan(src):
    out = "/tmp/scan_%s.s" % src
    cmd = [build.HIPCC, "-S", "--cuda-device-only"] + [f for f in build.FLAGS if f not in ("-fPIC",)] + build.EXTRA_FLAGS.get(src, []) + \
          ["-I" + os.path.join(here, "csrc"), "-I" + os.path.join(here, "..", "include"), os.path.join(here, "csrc", src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(src, "FAILED", r.stderr[-300:]); return
    lines = open(out).read().split("\n")
    name, chain, best, pending = None, 0, {}, False
    for l in lines:
        t = l.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, chain, pending = m.group(1), 0, False
            continue
        if name is None:
            continue
        if t.startswith((".Lfunc_end",)):
            name = None; continue
        op = t.split(" ")[0]
        if op.startswith(("global_load", "flat_load", "buffer_load")) and "lds" not in op:
            if pending is False:
                pending = True            # a load is in flight, not yet waited for
        elif op == "s_waitcnt" and "vmcnt(0)" in t:
            if pending:
                chain += 1                # this wait exposes (at least) one load's round trip
                best[name] = max(best.get(name, 0), chain)
            pending = False
        elif op in ("s_barrier", "s_endpgm") or op.startswith("s_cbranch") and False:
            pass
    for k, v in sorted(best.items(), key=lambda kv: -kv[1]):
        if v >= int(os.environ.get("SCAN_MIN", "6")):
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%-22s %4d load-then-vmcnt(0) waits   %s" % (src, v, d[:150]))

for src in (sys.argv[1:] or [s for s in build.SOURCES if s.endswith(".hip")]):
    scan(src)
