"""Shape sweep of the CIN forward / backward kernels and the cross / gather kernels (development tool): prints a markdown
table of time and achieved rate per shape.  python tools/sweep_shapes.py > gpurun_out/sweep.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dir_amd  # noqa: E402,F401
from dir_amd import ops  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 65536
    print("### CIN layer, B = 65536 (forward: dir_cin_layer_f32 and what arith='auto' runs; backward: dir_cin_dw_f32 + dir_cin_dx_f32 or the "
          "forward-kernel form; TFLOP/s are fp32-equivalent = 2*B*D*Hp*m*H / time)\n")
    print("| m | D | Hp | H | fp32-MFMA forward ms | TFLOP/s | of 157.3 | auto | auto forward ms | TFLOP/s | backward ms | TFLOP/s (2 GEMMs) | of 157.3 |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for m, D, Hp, H in [(26, 16, 26, 128), (26, 16, 128, 128), (26, 16, 64, 64), (26, 16, 200, 200), (26, 16, 200, 100), (26, 16, 100, 200), (39, 16, 39, 128), (39, 16, 128, 128),
                        (16, 16, 128, 128), (8, 16, 64, 64), (26, 8, 128, 128), (26, 32, 128, 128), (26, 4, 128, 128), (26, 16, 128, 32)]:
        Bs = B if D * max(Hp, H) * B * 4 < 3e9 else B // 2
        x0 = torch.randn((Bs, m, D), generator=g, device="cuda") * 0.25
        xk = torch.randn((Bs, Hp, D), generator=g, device="cuda") * 0.25
        W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
        G = torch.randn((Bs, H, D), generator=g, device="cuda") * 0.1
        fl = 2.0 * Bs * D * Hp * m * H
        tf = timeit(lambda: ops.cin_layer(x0, xk, W, arith="f32"))
        auto = ops.cin_auto_arith(m, D, Hp, H)
        ta = timeit(lambda: ops.cin_layer(x0, xk, W, arith=auto)) if auto != "f32" else tf
        tb = timeit(lambda: ops.cin_layer_backward(x0, xk, W, G), iters=3, warm=1)
        print("| %d | %d | %d | %d | %.3f | %.1f | %.2f | %s | %.3f | %.1f | %.3f | %.1f | %.2f |" % (
            m, D, Hp, H, tf, fl / tf / 1e9, fl / tf / 1e9 / 157.3, auto, ta, fl / ta / 1e9, tb, 2 * fl / tb / 1e9, 2 * fl / tb / 1e9 / 157.3))
        del x0, xk, W, G
    print("\n### DCN cross, B = 65536, L = 3 (forward / backward)\n")
    print("| d | forward µs | TB/s | backward µs | TB/s |")
    print("|---|---|---|---|---|")
    for d in [51, 64, 128, 256, 416, 429, 512, 1024]:
        x0 = torch.randn((B, d), generator=g, device="cuda") * 0.25
        w = (torch.randn((3, d), generator=g, device="cuda") * 0.1).clamp_(-0.2, 0.2)
        b = (torch.randn((3, d), generator=g, device="cuda") * 0.1).clamp_(-0.2, 0.2)
        go = torch.randn((B, d), generator=g, device="cuda") * 0.01
        out = torch.empty_like(x0)
        tf = timeit(lambda: ops.cross_network(x0, w, b, out=out), iters=50, warm=5)
        tb = timeit(lambda: ops.cross_network_backward(x0, w, b, go), iters=50, warm=5)
        print("| %d | %.1f | %.2f | %.1f | %.2f |" % (d, tf * 1e3, B * 2 * 4 * d / tf / 1e9, tb * 1e3, B * 3 * 4 * d / tb / 1e9))
    print("\n### DIN attention pool, B = 65536, T = 50, lengths U{1..50}\n")
    print("| K | H1 | H2 | ms |")
    print("|---|---|---|---|")
    for K, H1, H2 in [(64, 80, 40), (32, 48, 16), (16, 32, 16), (64, 64, 32)]:
        V = 1000000
        table = torch.randn((V, K), generator=g, device="cuda") * 0.125
        hist = torch.randint(0, V, (B, 50), generator=g, device="cuda")
        hl = torch.randint(1, 51, (B,), generator=g, device="cuda", dtype=torch.int32)
        cand = torch.randint(0, V, (B,), generator=g, device="cuda")
        W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.05
        W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.1
        W3 = torch.randn((H2,), generator=g, device="cuda") * 0.1
        z1, z2, z3 = torch.zeros(H1, device="cuda"), torch.zeros(H2, device="cuda"), torch.zeros(1, device="cuda")
        t = timeit(lambda: ops.din_attention_pool(table, hist, hl, cand, W1, z1, W2, z2, W3, z3, normalize=True), iters=20, warm=3)
        print("| %d | %d | %d | %.3f |" % (K, H1, H2, t))


if __name__ == "__main__":
    main()
