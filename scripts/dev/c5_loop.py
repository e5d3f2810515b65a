"""Dev: three eager C5 (magnet_gnn, 256 LR nodes -> 256^2 queries) rollouts (for PMC passes over the chain kernel)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
name = sys.argv[1] if len(sys.argv) > 1 else "magnet_gnn"
torch.manual_seed(0)
m = FACTORY[name](types.SimpleNamespace(**bench.MAGNET_HP[name], hip_graph=False)).cuda().eval()
b, _ = bench.synthetic_magnet_batch(name, 1, 256, 2000, "cuda")
for _ in range(3):
    m.validation_step(b, 0)
torch.cuda.synchronize()
print("done")
