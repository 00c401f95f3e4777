#!/usr/bin/env python3
"""tools/lds_waits.py DIR [MIN_MFMA]: over hipcc -S output of csrc/*.hip, per basic block with >= MIN_MFMA (24) matrix instructions: the LDS reads,
and how its s_waitcnt lgkmcnt waits are counted -- lgkmcnt(0) right behind freshly issued reads means the prefetch is waited for too."""
import os, re, sys, subprocess
d = sys.argv[1]; mn = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for f in sorted(os.listdir(d)):
    name = None; blk = None; out = []
    def flush():
        if blk and blk["mfma"] >= mn: out.append((name, dict(blk)))
    for l in open(os.path.join(d, f)):
        m = re.match(r"^(_Z\w+):", l)
        if m: flush(); name = m.group(1); blk = None; continue
        if re.match(r"^(\.LBB|; %bb)", l): flush(); blk = {"mfma": 0, "ds_read": 0, "wait0": 0, "waitN": 0, "fresh0": 0, "since": 99}; continue
        if blk is None: continue
        t = l.strip()
        if t.startswith("v_mfma"): blk["mfma"] += 1; blk["since"] += 1
        elif t.startswith("ds_read"): blk["ds_read"] += 1; blk["since"] = 0
        elif t.startswith("s_waitcnt") and "lgkmcnt" in t:
            if "lgkmcnt(0)" in t:
                blk["wait0"] += 1
                if blk["since"] == 0: blk["fresh0"] += 1
            else: blk["waitN"] += 1
    flush()
    for n, b in out:
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        print("%-18s mfma %3d ds_read %3d  lgkmcnt(0) %2d (right behind a read: %2d)  counted %2d  %s" % (f[:-6], b["mfma"], b["ds_read"], b["wait0"], b["fresh0"], b["waitN"], dem[:90]))
