"""Dev timing of the small BASELINE configs (launch-latency regime): C1 mpnn 1D N=50 B=2 tw=25 nt=250.
(GPU only.  The CPU number quoted for C1 in DESIGN.md came from timing the oracle once; oracle/ is test infrastructure
and only tests/, smoke() and bench.py's cpu_baseline leg may import it, so this script does not.)"""
import sys, time, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from magnet_amd.models.factory import FACTORY

hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=25, teacher_forcing=False, neighbors=3)
torch.manual_seed(0)
B, N, nt = 2, 50, 250
batch = {"u": torch.randn(B, N, nt), "x": torch.linspace(0, 16, N)[None, :, None].repeat(B, 1, 1), "t": torch.linspace(0, 4, nt)[None].repeat(B, 1)}
hp.hip_graph = False
m = FACTORY["mpnn"](hp).eval().cuda()
cb = {k: v.cuda() for k, v in batch.items()}
for _ in range(3): m.validation_step(cb, 0)
torch.cuda.synchronize(); t0 = time.time()
K = 20
for _ in range(K): m.validation_step(cb, 0)
torch.cuda.synchronize(); dt = (time.time() - t0) / K
print(f"C1 GPU rollout (9 forwards): {dt*1e3:.2f} ms -> {B*N*225/dt:.3e} node-updates/s")
ref = m.validation_step(cb, 0).clone()
m.hip_graph = True
for _ in range(3): got = m.validation_step(cb, 0)
print("hipGraph replay equals eager:", bool(torch.equal(got, ref)))
torch.cuda.synchronize(); t0 = time.time()
for _ in range(K): m.validation_step(cb, 0)
torch.cuda.synchronize(); dg = (time.time() - t0) / K
print(f"C1 GPU rollout with hipGraph replay: {dg*1e3:.2f} ms -> {B*N*225/dg:.3e} node-updates/s")
