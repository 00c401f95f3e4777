"""tools/libcall_probe.py WORKLOAD [steps]: which library GEMMs (torch.matmul / mm / bmm / addmm / F.linear) a bench.py step still issues, with
operand shapes and the package line that asked -- the shapes the HIP dense kernels do not cover yet.  Runs bench.main() in-process."""
import sys, collections, traceback
sys.path.insert(0, ".")
import torch
import dir_amd  # noqa: F401
import bench

seen = collections.Counter()


def caller():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "details-in-recommendation_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return "%s:%d" % (fr.filename.split("/")[-1], fr.lineno)
    return "?"


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        shapes = tuple(tuple(t.shape) for t in a if isinstance(t, torch.Tensor))
        if any(isinstance(t, torch.Tensor) and t.is_cuda for t in a):
            seen[(label, shapes, caller())] += 1
        return f(*a, **k)
    setattr(obj, name, g)


for n in ("matmul", "mm", "bmm", "addmm", "baddbmm", "einsum", "mv", "addmv", "tensordot", "inner", "addbmm", "chain_matmul"):
    wrap(torch, n, "torch." + n)
for n in ("__matmul__", "__rmatmul__", "matmul", "mm", "bmm", "addmm", "addmm_", "mv", "addmv", "addmv_", "baddbmm", "baddbmm_", "addbmm", "addbmm_"):
    wrap(torch.Tensor, n, "Tensor." + n)
wrap(torch.nn.functional, "linear", "F.linear")
wl = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = ["bench.py", "--workload", wl, "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline"]
bench.main()
n = steps + 2
print("library GEMM calls of %s (per step; setup calls show as fractions):" % wl)
for (label, shapes, where), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("  %6.2f  %-18s %-44s %s" % (c / n, label, " x ".join(str(s) for s in shapes), where))
