"""Timing of one MAgNet[GNN] training step at the C5 size (256 LR nodes -> Wq x Wq queries): rollout with autograd
Functions (fused HIP forward kernels) + HIP backward + FusedAdam.  Correctness-first path: nothing is fused."""
import json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from magnet_amd.models.factory import FACTORY

Wq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NAME = sys.argv[4] if len(sys.argv) > 4 else "magnet_gnn"          # magnet_gnn (C5) | magnet_cnn_2d (C4b)
torch.manual_seed(0)
model = FACTORY[NAME](types.SimpleNamespace(**bench.MAGNET_HP[NAME])).cuda().train()
batch, nodes = bench.synthetic_magnet_batch(NAME, B, Wq, 5, "cuda")
opt = model.configure_optimizers()["optimizer"]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    return float(loss.detach())


l0 = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l1 = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
T = bench.MAGNET_HP[NAME]["time_slice"]
print(json.dumps({"what": NAME + " training step (C5 / C4b size)", "queries": Wq * Wq, "traj": B, "ms_per_step": round(dt * 1e3, 1),
                  "node_updates_per_s": round(B * nodes * (bench.NT - T) / dt, 1), "loss_first_last": [round(l0, 5), round(l1, 5)],
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
