"""tools/host_overhead.py WORKLOAD [steps]: host time spent inside the package's Python entry points during bench.py's step (launches are
asynchronous: with the GPU ahead of the host this is the step's launch-bound share).  Wraps every public function of dir_amd.ops and the
autograd nodes of dir_amd.dense / dir_amd.autograd with perf_counter accumulators (inclusive times), runs bench.main() in-process."""
import sys, time, types, collections
sys.path.insert(0, ".")
import dir_amd
from dir_amd import ops, dense, autograd as ag
import bench

acc = collections.defaultdict(lambda: [0, 0.0])
depth = [0]


def wrap(mod, name, label):
    f = getattr(mod, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[label]
            e[0] += 1
            e[1] += time.perf_counter() - t0
    g.__name__ = name
    setattr(mod, name, g)


for n in dir(ops):
    f = getattr(ops, n)
    if isinstance(f, types.FunctionType) and not n.startswith("__") and f.__module__ == ops.__name__:
        wrap(ops, n, "ops." + n)
for cls in (dense._MlpHeadFn, dense._MlpStackFn, dense._DenseFn, dense._DenseBnFn):
    for m in ("forward", "backward"):
        f = getattr(cls, m)

        def mk(f, label):
            def g(*a, **k):
                t0 = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    e = acc[label]
                    e[0] += 1
                    e[1] += time.perf_counter() - t0
            return staticmethod(g)
        setattr(cls, m, mk(f, "dense.%s.%s" % (cls.__name__, m)))
wl = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sys.argv = ["bench.py", "--workload", wl, "--steps", str(steps), "--warmup", "8", "--no-cpu-baseline"]
for e in acc.values():
    e[0] = 0; e[1] = 0.0
bench.main()
n = steps + 8
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
print("%-50s %8s %10s %10s" % ("function (inclusive)", "calls/st", "us/step", "us/call"))
for k, (c, t) in rows[:28]:
    if c:
        print("%-50s %8.1f %10.1f %10.1f" % (k, c / n, t / n * 1e6, t / c * 1e6))
