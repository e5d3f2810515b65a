"""Dev: how much of a graph-replayed headline rollout is host time during which the GPU idles?  (validation_step ends with one host
read of the range-guard status word, so everything the host does between two rollouts is exposed)"""
import cProfile, io, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd.models.factory import FACTORY
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
batch = bench.synthetic_batch(256, 4, 2000, "cuda")
for _ in range(3):
    m.validation_step(batch, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    m.validation_step(batch, 0)
torch.cuda.synchronize()
print(f"wall per rollout {(time.perf_counter() - t0) * 100:.2f} ms")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gpu = []
for _ in range(5):
    torch.cuda.synchronize()
    s.record()
    m.validation_step(batch, 0)
    e.record()
    torch.cuda.synchronize()
    gpu.append(s.elapsed_time(e))
print("GPU time of one isolated rollout (events):", [round(g, 2) for g in gpu])
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.validation_step(batch, 0)
pr.disable()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumtime").print_stats(28)
print("\n".join(l for l in buf.getvalue().splitlines() if l.strip())[:5000])
