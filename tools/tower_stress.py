"""tools/tower_stress.py (GPU box) -- run-to-run bitwise equality of dir_tower_bf16x3_f32 at the DeepFM tower shape (65536 x 416 -> 400 -> 400
-> 400 [-> 1]): R reruns on identical inputs; rows that differ from the first run are counted."""
import sys
import torch
sys.path.insert(0, ".")
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
gen = torch.Generator(device="cuda").manual_seed(1)
M, dims = 65536, [416, 400, 400, 400]
x = torch.randn((M, dims[0]), generator=gen, device="cuda") * 0.25
Ws = [torch.randn((dims[i + 1], dims[i]), generator=gen, device="cuda") / dims[i] ** 0.5 for i in range(3)]
bs = [torch.randn((dims[i + 1],), generator=gen, device="cuda") * 0.1 for i in range(3)]
hw, hb = torch.randn((1, 400), generator=gen, device="cuda") / 20, torch.full((1,), 0.1, device="cuda")
for name, f in (("3 layers", lambda: ops.tower(x, Ws, bs)), ("3 layers + head", lambda: ops.tower(x, Ws, bs, head=(hw, hb)))):
    y0 = f().clone()
    runs = rows = 0
    for it in range(R):
        y = f()
        bad = (y != y0).any(1)
        n = int(bad.sum())
        runs += n > 0
        rows += n
    print("tower %s: %d / %d reruns differ from the first, %d differing rows in total" % (name, runs, R, rows), flush=True)
