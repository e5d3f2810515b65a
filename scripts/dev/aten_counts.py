"""Dev: which torch operators (and how many C-ABI calls) does ONE warm training step issue?  usage: python scripts/dev/aten_counts.py gnn_64|cnn_64|c4a|gnn_512"""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY

name = sys.argv[1] if len(sys.argv) > 1 else "gnn_64"
torch.manual_seed(0)
mname, hp, mk = bench.ref_shape_cases()[name]
model = FACTORY[mname](types.SimpleNamespace(**dict(hp, lr=1e-6))).cuda().train()
batch, nodes = mk(True)
batch = {k: v.cuda() for k, v in batch.items()}
opt = model.configure_optimizers()["optimizer"]


def step():
    opt.zero_grad(set_to_none=True)
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
lib = ops.L.lib()
names = [n for n in ops.L.declared_symbols() if hasattr(lib, n)]
calls = collections.Counter()
origs = {n: getattr(lib, n) for n in names}


def counted(n, fn):
    def f(*a):
        calls[n] += 1
        return fn(*a)
    return f


for n, fn in origs.items():
    setattr(ops.L._lib, n, counted(n, fn))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
    step()
for n, fn in origs.items():
    setattr(ops.L._lib, n, fn)
torch.cuda.synchronize()
print(f"{name}: C-ABI calls {sum(calls.values())}")
for n, c in calls.most_common(25):
    print(f"  {c:5d}  {n}")
ev = collections.Counter()
tm = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::"):
        ev[e.name] += 1
        tm[e.name] += e.cpu_time_total
print("aten operators (count, total host us):")
for n, c in ev.most_common(30):
    print(f"  {c:5d}  {tm[n]:9.0f}  {n}")
