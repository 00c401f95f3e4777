"""tools/asm_patch.py dev.s -- edits of the bf16x3 din_wave_k body selected by $PATCH_MODE (development: hazard bisection)."""
import os, re, sys
path = sys.argv[1]
mode = os.environ.get("PATCH_MODE", "")
L = open(path).read().split("\n")
start = next(i for i, l in enumerate(L) if re.match(r"^_ZN3dir10din_wave_kINS_10DinWaveSh3", l))
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
out = L[:start]
body = L[start:end + 1]
def is_op(l, op): return l.strip().startswith(op)
n = 0
new = []
for i, l in enumerate(body):
    if mode == "nop_before_bperm" and is_op(l, "ds_bpermute_b32"):
        new.append("\ts_nop 7"); n += 1
    new.append(l)
    if mode == "nop_after_last_mfma" and is_op(l, "v_mfma"):
        nxt = [x for x in body[i + 1:i + 120] if x.strip() and not x.strip().startswith(";")]
        k_m = next((k for k, x in enumerate(nxt) if is_op(x, "v_mfma")), 10**9)
        k_b = next((k for k, x in enumerate(nxt) if is_op(x, "ds_bpermute_b32")), 10**9)
        if k_b < k_m:
            new += ["\ts_nop 15"] * 8; n += 1
    if mode == "nop_after_bperm_wait" and is_op(l, "s_waitcnt lgkmcnt(0)") and any(is_op(x, "ds_bpermute_b32") for x in body[max(0, i - 3):i]):
        new.append("\ts_nop 7"); n += 1
open(path, "w").write("\n".join(out + new + L[end + 1:]))
print("asm_patch %s: %d sites" % (mode, n))
