"""Dev: what does hip_graph="rebuild" (graph + plan built afresh before every replay, as the bench's headline does) cost over a plain
replay, and where: the device work of the build, and the host round trips inside it."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
dev = torch.device("cuda", 0)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


b = bench.synthetic_batch(256, 4, 1000, dev)
res = {}
for mode in (True, "rebuild", True, "rebuild"):
    torch.manual_seed(0)
    m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP, hip_graph=mode)).to(dev).eval()
    res.setdefault(str(mode), []).append(round(timeit(lambda: m.validation_step(b, 0)), 3))
print("rollout ms:", res)
x = b["x"]
t_topo = timeit(lambda: m._build_topology(x[0], 4), reps=20)
topo = m._build_topology(x[0], 4)
t_plan = timeit(lambda: ops.stream_tiles(topo.csr), reps=20)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
for _ in range(20):
    m._build_topology(x[0], 4)
e.record()
torch.cuda.synchronize()
print(f"_build_topology: {t_topo:.3f} ms wall per call, {s.elapsed_time(e) / 20:.3f} ms between events; stream_tiles {t_plan:.3f} ms wall")
