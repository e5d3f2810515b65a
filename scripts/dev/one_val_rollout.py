"""Dev: warm up, then run validation rollouts of one of the reference's shapes (for a kernel trace).
usage: python scripts/dev/one_val_rollout.py gnn_64|gnn_512|cnn_64|c4a [eager]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
key = sys.argv[1] if len(sys.argv) > 1 else "gnn_64"
name, hp, mk = bench.ref_shape_cases()[key]
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**hp, hip_graph=not (len(sys.argv) > 2 and sys.argv[2] == "eager"))).cuda().eval()
batch, nodes = mk(False)
batch = {k: v.cuda() for k, v in batch.items()}
for _ in range(3):
    model.validation_step(batch, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    model.validation_step(batch, 0)
torch.cuda.synchronize()
print(f"{key}: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms per validation rollout")
