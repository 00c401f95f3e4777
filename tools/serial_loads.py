#!/usr/bin/env python3
"""tools/serial_loads.py DIR: over the gfx950 assembly files in DIR (hipcc -S of csrc/*.hip), per kernel the number of global loads that are
waited for (s_waitcnt vmcnt(0)) before the next global load is issued -- dependent memory round trips the compiler created by sinking a
load into the branch of its only use.  Prints kernels with at least MIN (default 6) such loads."""
import re, sys, os, subprocess
d = sys.argv[1]; mn = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for f in sorted(os.listdir(d)):
    lines = open(os.path.join(d, f)).read().split("\n")
    name, cnt, pending, loads = None, 0, False, 0
    res = []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if name: res.append((name, cnt, loads))
            name, cnt, pending, loads = m.group(1), 0, False, 0
            continue
        t = l.strip()
        if t.startswith(("global_load", "buffer_load")) and "lds" not in t:
            loads += 1; pending = True
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and pending:
            cnt += 1; pending = False
    if name: res.append((name, cnt, loads))
    for n, c, l in res:
        if c >= mn:
            dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
            print("%-22s waits %3d  loads %3d  %s" % (f[:-2], c, l, dem[:110]))
