"""tools/torch_glue_profile.py [dcn_train|esmm_train|deepfm_train] -- which torch ops (with input shapes) a training step still runs beside the
HIP kernels: torch.profiler over a few steps of the bench.py workload's model, device time per (op, shapes), largest first.  Development only."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import dir_amd  # noqa: F401
from dir_amd import feature_column as fc

wl = sys.argv[1] if len(sys.argv) > 1 else "dcn_train"
B, F, V, K = 65536, 26, 1000000, 16
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(0, V, (B, F), generator=gen, device=dev)
labels = (torch.rand((B, 1), generator=gen, device=dev) < 0.25).float()
if wl == "dcn_train":
    from dir_amd.dcn import DeepCrossNetwork
    cols = [fc.embedding_column(fc.categorical_column_with_identity("C%02d" % i, V), K) for i in range(F)]
    cols += [fc.numeric_column("I%02d" % i) for i in range(13)]
    model = DeepCrossNetwork(columns=cols, cross_layer_num=3, dnn_hidden_units=[1024, 1024], batch_norm=True, optimizer="Adam",
                             optimizer_spec={"epsilon": 1e-4},
                             learning_rate_spec={"learning_rate": 0.001, "decay_method": "cosine_decay", "decay_steps": 3000, "alpha": 0.5}).to(dev)
    dense = torch.rand((B, 13), generator=gen, device=dev)
    feats = {"C%02d" % i: ids[:, i].contiguous() for i in range(F)}
    feats.update({"I%02d" % i: dense[:, i].contiguous() for i in range(13)})
    train_op = model.train_step()

    def step():
        train_op(torch.nn.functional.binary_cross_entropy_with_logits(model(feats), labels))
else:
    raise SystemExit("workload not wired: " + wl)

for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0:
        rows.append((t / N, e.count / N, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("device time per step: %.1f us" % tot)
for t, c, k, s in rows[:70]:
    print("%8.1f us  x%-5.1f %-45s %s" % (t, c, k[:45], s))
