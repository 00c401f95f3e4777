import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dir_amd
from dir_amd import feature_column as fc, _input
dir_amd.load_library()
B, F, V = 65536, 26, 1000000
cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
feats = {"C%d" % i: torch.randint(0, V, (B,), device="cuda") for i in range(F)}
big = torch.randn(4096, 4096, device="cuda")
def run(mode, n=50):
    _input.CHECK_MODE = mode
    for _ in range(5):
        _input.collect_ids(cats, feats, "cuda"); _input.raise_pending()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        _input.collect_ids(cats, feats, "cuda")
        y = big @ big          # stands for the rest of the forward (~1 ms)
        _input.raise_pending()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for m in ("off", "sync", "deferred", "off"):
    print(m, "%.3f ms" % run(m))
