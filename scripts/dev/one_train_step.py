"""Dev: warm up, then run training steps of one workload (for a kernel trace): two warm-up steps, then 2 traced steps; prints host
enqueue vs wall time.  usage: python scripts/dev/one_train_step.py magnet_gnn|magnet_cnn_2d|mpnn_2d"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY

name = sys.argv[1] if len(sys.argv) > 1 else "magnet_gnn"
torch.manual_seed(0)
if name in ("gnn_64", "gnn_512", "cnn_64", "c4a"):                 # the reference's own shapes (bench.ref_shape_cases)
    mname, hp, mk = bench.ref_shape_cases()[name]
    model = FACTORY[mname](types.SimpleNamespace(**hp)).cuda().train()
    batch, nodes = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
elif name.startswith("magnet"):
    model = FACTORY[name](types.SimpleNamespace(**bench.MAGNET_HP[name])).cuda().train()
    batch, nodes = bench.synthetic_magnet_batch(name, 1, 256, 2000, "cuda")
else:
    model = FACTORY[name](types.SimpleNamespace(**bench.HP)).cuda().train()
    batch = bench.synthetic_batch(256, 4, 2000, "cuda")
opt = model.configure_optimizers()["optimizer"]


losses = []


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    losses.append(loss.detach())


for _ in range(2):
    step()
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0):.1f} ms, + drain {1e3 * (t2 - t1):.1f} ms = {1e3 * (t2 - t0):.1f} ms per step")
print("losses:", [round(float(l), 4) for l in losses])
