"""Dev: time k_mpnn_edge_stream32 on the metric graph (4 x 256^2).  MAGNET_LIB_PATH selects a role-ablation build (-DES_ROLES=1: waves 0-3
issue only the tile's MFMAs, their SIMD partners 4-7 only the rest; 2: every wave the matrix stream alone; 3: every wave the rest alone)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from magnet_amd import _lib, ops  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402
from magnet_amd.packing import pack_edge16_fp16x2, pack_edge32t_fp16x2  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, 256)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, 4).csr
N, E, H = csr.N, csr.E, 128
A, Bm = torch.randn(N, H, device="cuda"), torch.randn(N, H, device="cuda")
l2 = m.gnn_layers[0].message_net_2[0]
w32, _ = pack_edge32t_fp16x2(l2.weight, l2.bias)
_, b = pack_edge16_fp16x2(l2.weight, l2.bias)
plan = ops.stream_tiles(csr)
agg = torch.empty(N, H, device="cuda")
fn = lambda: ops.mpnn_edge_stream(A, Bm, csr, plan, w32, b, agg, l2e=True)  # noqa: E731
out = []
for r in range(R):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    out.append(s.elapsed_time(e) / 20)
print(f"{os.path.basename(_lib.LIB_PATH)} k_mpnn_edge_stream32 (E = {E}): " + "  ".join(f"{v:.4f}" for v in out) + " ms")
