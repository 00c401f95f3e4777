"""tools/small_batch_kernels.py (GPU box, under rocprofv3): 50 eager DeepFM forwards and 50 DCN forwards at batch 256 -- which kernels a
forward launches at the reference's batch size, and how long each runs."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402
from dir_amd import feature_column as fc  # noqa: E402
from dir_amd.deepfm import DeepFM  # noqa: E402
from dir_amd.dcn import DeepCrossNetwork  # noqa: E402
dir_amd.load_library()
dev = torch.device("cuda:0")
F, V, K, B = 26, 100000, 16, 256
cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
which = sys.argv[1] if len(sys.argv) > 1 else "deepfm"
if which == "deepfm":
    m = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400, 400],
               fm_embedding_size=K).to(dev).eval()
    ids = torch.randint(0, V, (B, F), device=dev)
    f = lambda: m.forward_ids(ids, ids)      # noqa: E731
else:
    nums = [fc.numeric_column("I%d" % i) for i in range(13)]
    m = DeepCrossNetwork(columns=[fc.embedding_column(c, K) for c in cats] + nums, cross_layer_num=3, dnn_hidden_units=[1024, 1024]).to(dev).eval()
    ids = torch.randint(0, V, (B, F), device=dev)
    feats = {"C%d" % i: ids[:, i].contiguous() for i in range(F)}
    feats.update({"I%d" % i: torch.rand(B, 1, device=dev) for i in range(13)})
    f = lambda: m(feats)                      # noqa: E731
with torch.no_grad():
    for _ in range(50):
        f()
torch.cuda.synchronize()
