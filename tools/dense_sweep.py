"""dense_sweep.py -- dir_dense_f32 vs torch (rocBLAS addmm + relu) over layer shapes and input row strides (development tool)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dir_amd  # noqa: E402,F401
from dir_amd import ops  # noqa: E402


def t(M, Kd, N, ld=None, wld=None, relu=True):
    ld = ld or Kd
    wld = wld or Kd
    xf = torch.randn(M, ld, device="cuda")
    x = xf[:, :Kd]
    w = torch.randn(N, wld, device="cuda")[:, :Kd]
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20
    us = run(lambda: ops.dense(x, w, b, relu=relu, out=y))
    us2 = run(lambda: torch.relu(torch.addmm(b, x, w.t())))
    tf = 2 * M * Kd * N / us / 1e6
    print("M %6d Kd %5d (ld %5d wld %5d) N %5d: dense %8.1f us %6.1f TF (%.3f) | torch addmm+relu %8.1f us %6.1f TF" % (
        M, Kd, ld, wld, N, us, tf, tf / 157.3, us2, 2 * M * Kd * N / us2 / 1e6))


if __name__ == "__main__":
    for s in [(65536, 416, 400), (65536, 416, 400, 416, 420), (65536, 416, 400, 416, 432), (65536, 416, 400, 416, 448), (65536, 416, 400, 448, 448),
              (65536, 416, 400, 432, 432), (65536, 400, 400), (65536, 400, 400, 400, 416), (65536, 400, 400, 400, 432), (65536, 384, 400), (65536, 1024, 1024), (65536, 1024, 1024, 1024, 1040)]:
        t(*s)
