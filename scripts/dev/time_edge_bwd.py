"""Dev: time one generation of the fused MPNN edge backward on the metric graph (4 x 256^2).  usage: time_edge_bwd.py [generation] [rounds]
(MAGNET_LIB_PATH selects an ablation build: scripts/dev/build_co_ablations.sh)"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from magnet_amd import _lib, ops  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402
from magnet_amd.packing import pack  # noqa: E402

gen = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, 256)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, 4).csr
N, E, H = csr.N, csr.E, 128
A, Bm = torch.randn(N, H, device="cuda"), torch.randn(N, H, device="cuda")
dagg = torch.randn(N, H, device="cuda") * 1e-7
l2 = m.gnn_layers[0].message_net_2[0]
w2, w2t = pack(l2.weight, l2.bias, "fp16x3"), pack(l2.weight.t(), None, "fp16x3")
partials = torch.empty(((E + 31) // 32) * 2 * H + 1, device="cuda")
lib = _lib.lib()
ws = torch.empty(lib.magnet_mpnn_edge_bwd_w_workspace(), device="cuda")
lib.magnet_mpnn_edge_bwd_w_generation(gen)
dp, dA = torch.empty(E, H, device="cuda"), torch.empty(N, H, device="cuda")
gW, gb = torch.zeros(H, H, device="cuda"), torch.zeros(H, device="cuda")
fn = lambda: ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW, gb, ws=ws, accumulate=False)  # noqa: E731
out = []
for r in range(R):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    out.append(s.elapsed_time(e) / 10)
print(f"{os.path.basename(_lib.LIB_PATH)} generation {gen}: " + "  ".join(f"{v:.3f}" for v in out) + " ms")
