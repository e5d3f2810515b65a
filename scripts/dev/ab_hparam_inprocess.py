"""Dev: A/B of one mpnn_2d hparam inside ONE process - two models (same weights), alternating blocks of rollouts on the metric batch
(hip_graph="rebuild", as the headline).  usage: python scripts/dev/ab_hparam_inprocess.py fused_node_stats False True [blocks] [rollouts per block]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
name, va, vb = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3])
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 8
per = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda", 0)
b = bench.synthetic_batch(256, 4, 1000, dev)
models = []
for v in (va, vb):
    torch.manual_seed(0)
    models.append(FACTORY["mpnn_2d"](types.SimpleNamespace(**dict(bench.HP, hip_graph="rebuild", **{name: v}))).to(dev).eval())
for m in models:
    for _ in range(4):
        m.validation_step(b, 0)
torch.cuda.synchronize()
t = [[], []]
for blk in range(blocks):
    for i, m in enumerate(models):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            m.validation_step(b, 0)
        torch.cuda.synchronize()
        t[i].append((time.perf_counter() - t0) / per * 1e3)
for i, v in enumerate((va, vb)):
    xs = sorted(t[i])
    print(f"{name}={v}: median {xs[len(xs) // 2]:.3f} ms, mean {sum(xs) / len(xs):.3f}, min {xs[0]:.3f}, blocks {[round(x, 2) for x in t[i]]}")
ya, yb = models[0].validation_step(b, 0), models[1].validation_step(b, 0)
print("max |difference| of the two rollouts:", float((ya - yb).abs().max()), "relative L2", float((ya - yb).norm() / ya.norm()))
