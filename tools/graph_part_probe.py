"""tools/graph_part_probe.py (GPU box): which part of a training step does not survive `capture -> other eager work -> replay`?
DIR_PROBE_PART = sparse (ops.gather_fm + SparseAdagrad.step_fm + SparseFtrl.step, two independent table sets A / B),
                 dense  (three hidden layers + units-1 head forward / backward + HIP dense Adagrad on a fixed input),
                 sort   (SparseAdagrad.step only).
Twin A runs eagerly N times between B's capture and B's one replay."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd  # noqa: E402,F401
from dir_amd import autograd as ag, ops  # noqa: E402

part = os.environ.get("DIR_PROBE_PART", "sparse")
N = int(os.environ.get("DIR_PROBE_N", "60"))
B, F, K, V = 65536, 26, 16, int(os.environ.get("DIR_PROBE_V", "200000"))
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)


def make():
    if part in ("sparse", "sort"):
        tables = [torch.randn((V, K), generator=gen, device=dev) * 0.25 for _ in range(F)]
        ts = ops.TableSet.train_rows(tables) if part == "sparse" else ops.TableSet(tables)
        opt = ops.SparseAdagrad(ts, lr=0.01)
        lin = ops.TableSet([torch.zeros((V,), device=dev) for _ in range(F)])
        ftrl = ops.SparseFtrl(lin, lr=0.2)
        if part == "sparse":
            ops.share_sorted_entries(opt, ftrl)
        ids = torch.randint(0, V, (B, F), generator=gen, device=dev)
        out = torch.empty((B, F * K), device=dev)
        fm = torch.empty((B, 1), device=dev)
        fsum = torch.empty((B, K), device=dev)
        gfm = torch.randn((B, 1), generator=gen, device=dev) * 0.01
        gd = torch.randn((B, F * K), generator=gen, device=dev) * 0.01
        glin = torch.randn((B, 1), generator=gen, device=dev) * 0.01

        def step():
            if part == "sort":
                opt.step(ids, gd)
                return
            ops.gather_fm(ts, ids, out=out, fm=fm, fsum=fsum)
            opt.step_fm(ids, gd, gfm, fsum)
            ftrl.step(ids, glin)
        return step
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    cats = [fc.categorical_column_with_identity("C%d" % i, 100) for i in range(F)]
    torch.manual_seed(3)
    m = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400, 400],
               fm_embedding_size=K).to(dev)
    dense_p = [p for n, p in m.named_parameters() if "embedding" not in n and "linear_w" not in n]
    od = ag.Adagrad(dense_p, lr=0.01, initial_accumulator_value=0.1)
    x = torch.randn((B, F * K), generator=gen, device=dev) * 0.25
    y = (torch.rand((B, 1), generator=gen, device=dev) < 0.25).float()

    def step():
        od.zero_grad(set_to_none=False)
        torch.nn.functional.binary_cross_entropy_with_logits(m.dnn_logit_fn(x), y).backward()
        od.step()
    return step


def main():
    a, b = make(), make()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            b()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        a()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        b()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("%s: captured, first replay ok" % part, flush=True)
    for _ in range(N):
        a()
    torch.cuda.synchronize()
    print("%s: %d eager steps of the twin done" % (part, N), flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("%s: replay after the twin's eager steps ok" % part, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
