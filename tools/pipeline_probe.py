"""tools/pipeline_probe.py (GPU box): DeepFM inference throughput with consecutive batches on alternating streams (the HBM-bound gather of batch
i + 1 under the matrix-bound tower of batch i) against one stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd.deepfm import DeepFM
from dir_amd import feature_column as fc
B, F, K, V = 65536, 26, 16, 1000000
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400, 400],
               fm_embedding_size=K).to(dev).eval()
idsl = [torch.randint(0, V, (B, F), generator=gen, device=dev) for _ in range(4)]
with torch.no_grad():
    ref = [model.forward_ids(ids).clone() for ids in idsl]


def run(n_streams, n=200):
    streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]
    outs = [None] * 4
    with torch.no_grad():
        for it in range(8):
            with torch.cuda.stream(streams[it % len(streams)]):
                model.forward_ids(idsl[it % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(n):
            with torch.cuda.stream(streams[it % len(streams)]):
                outs[it % 4] = model.forward_ids(idsl[it % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    ok = all(torch.equal(outs[i], ref[i]) for i in range(4))
    print("%d stream(s): %.1f us per batch, %.1f M samples/s, outputs bit-equal to the single-stream run: %s" % (n_streams, dt * 1e6, B / dt / 1e6, ok), flush=True)


run(1); run(2); run(3); run(1); run(2)
