"""Dev: the C1 rollout (mpnn 1-D, 2 x 50 nodes, 9 forwards) replayed from its hipGraph 200 times (for a kernel trace)."""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd.models.factory import FACTORY
hp = dict(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l1", hidden_features=128, hidden_layer=5, time_window=25,
          teacher_forcing=False, neighbors=3)
B, N, nt = 2, 50, 250
batch = {"u": torch.randn(B, N, nt, generator=torch.Generator().manual_seed(0)), "x": torch.linspace(0, 16, N)[None, :, None].repeat(B, 1, 1),
         "t": torch.linspace(0, 4, nt)[None].repeat(B, 1)}
cb = {k: v.cuda() for k, v in batch.items()}
torch.manual_seed(0)
m = FACTORY["mpnn"](types.SimpleNamespace(**hp, hip_graph=(sys.argv[1] if len(sys.argv) > 1 else "1") == "1")).cuda().eval()
for _ in range(5):
    m.validation_step(cb, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    m.validation_step(cb, 0)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per rollout")
