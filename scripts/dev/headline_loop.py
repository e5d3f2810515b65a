"""Dev: five eager mpnn_2d rollouts on the metric batch (for PMC passes over the node-level kernels)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph=False)).cuda().eval()
b = bench.synthetic_batch(256, 4, 1000, "cuda")
for _ in range(5):
    m.validation_step(b, 0)
torch.cuda.synchronize()
print("done")
