"""tools/small_batch_probe.py (GPU box): DeepFM / DCN inference at the reference's batch sizes (100 / 256, DeepCrossNetwork/train.py:16-17) and
1024: latency of one forward, eager and as a HIP-graph replay, under three routings of the hidden layers --
  library   : batches under dense.MIN_ROWS go to nn.Linear (rocBLAS / hipBLASLt): round 4's product routing,
  hip layers: every layer on dir_dense_* (dense.MIN_ROWS = 1, the fused tower off),
  hip tower : the fused tower kernel from row 1 (ops.TOWER_MIN_ROWS = 1; DeepFM: lookups + FM + linear + tower + head in ONE launch)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402
from dir_amd import ops, dense, feature_column as fc  # noqa: E402
from dir_amd.deepfm import DeepFM  # noqa: E402
from dir_amd.dcn import DeepCrossNetwork  # noqa: E402
from dir_amd.serving import GraphedForward  # noqa: E402

dir_amd.load_library()
dev = torch.device("cuda:0")
F, V, K = 26, 100000, 16
cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
torch.manual_seed(0)
deepfm = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400, 400],
                fm_embedding_size=K).to(dev).eval()
nums = [fc.numeric_column("I%d" % i) for i in range(13)]
dcn = DeepCrossNetwork(columns=[fc.embedding_column(c, K) for c in cats] + nums, cross_layer_num=3, dnn_hidden_units=[1024, 1024]).to(dev).eval()


def lat(fn, n=300):
    with torch.no_grad():
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


routes = {"library": (6144, 4096, "auto"), "hip layers": (1, 4096, "0"), "hip tower": (1, 1, "auto")}
for B in (100, 256, 1024, 4096):
    ids = torch.randint(0, V, (B, F), device=dev)
    feats = {"C%d" % i: ids[:, i].contiguous() for i in range(F)}
    feats.update({"I%d" % i: torch.rand(B, 1, device=dev) for i in range(13)})
    for name, model, fwd in (("deepfm", deepfm, lambda: deepfm.forward_ids(ids, ids)), ("dcn", dcn, lambda: dcn(feats))):
        for rname, (mr, tmr, tw) in routes.items():
            dense.MIN_ROWS, ops.TOWER_MIN_ROWS, ops.TOWER = mr, tmr, tw
            dense.reset_routing()
            try:
                e = lat(fwd)
                g = GraphedForward(lambda *a: fwd(), ids)
                gr = lat(lambda: g.graph.replay())
                print("B %5d %-7s %-11s eager %7.1f us  graph %7.1f us   library layers %s" % (B, name, rname, e, gr, dict(dense.ROUTING["library"]) or "{}"), flush=True)
            except Exception as ex:
                print("B %5d %-7s %-11s failed: %r" % (B, name, rname, ex), flush=True)
