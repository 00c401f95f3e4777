"""tools/sync_probe.py (GPU box): what the host-clock bracket of bench.py's timed region costs beyond the kernels: torch.cuda.synchronize() on an idle
device, event / stream synchronize, and 20 launches of the headline kernel timed four ways."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dir_amd  # noqa: F401
from dir_amd import ops

dev = torch.device("cuda:0")
B, F, K, V = 65536, 26, 16, 1_000_000
g = torch.Generator(device=dev).manual_seed(0)
ts = ops.TableSet([torch.randn((V, K), generator=g, device=dev) * 0.25 for _ in range(F)])
ids = [torch.randint(0, V, (B, F), generator=g, device=dev) for _ in range(4)]
out = torch.empty((B, F * K), device=dev)
fm = torch.empty((B, 1), device=dev)
step = lambda i: ops.gather_fm(ts, ids[i % 4], out=out, fm=fm)      # noqa: E731
for i in range(200):
    step(i)
torch.cuda.synchronize()

def med(f, n=50):
    v = sorted(f() for _ in range(n))
    return v[len(v) // 2] * 1e6

def t_sync_idle():
    t0 = time.perf_counter(); torch.cuda.synchronize(); return time.perf_counter() - t0
print("torch.cuda.synchronize() on an idle device: %.1f us" % med(t_sync_idle))
def t_stream_sync_idle():
    t0 = time.perf_counter(); torch.cuda.current_stream().synchronize(); return time.perf_counter() - t0
print("stream.synchronize() idle: %.1f us" % med(t_stream_sync_idle))

def region(end):
    torch.cuda.synchronize(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(20):
        step(i)
    e1.record()
    ti = time.perf_counter() - t0
    if end == "device":
        torch.cuda.synchronize(); torch.cuda.synchronize()
    elif end == "event":
        e1.synchronize()
    elif end == "stream":
        torch.cuda.current_stream().synchronize()
    elif end == "spin":
        while not e1.query():
            pass
    el = time.perf_counter() - t0
    torch.cuda.synchronize()
    return el, e0.elapsed_time(e1) * 1e-3, ti
for end in ("device", "event", "stream", "spin", "device"):
    rs = sorted(region(end) for _ in range(30))
    el, ev, ti = rs[len(rs) // 2]
    print("20 steps, end=%-6s host clock %.1f us  (%.2f us/step)   HIP events %.1f us (%.2f us/step)   issue %.1f us" % (end, el * 1e6, el * 1e6 / 20, ev * 1e6, ev * 1e6 / 20, ti * 1e6))
