"""Dev: 30 launches of k_mpnn_edge_stream32 on the metric graph (for PMC passes)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack_edge16_fp16x2, pack_edge32t_fp16x2
import bench
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
g = torch.linspace(0, 1, 256)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, 4).csr
A, Bm, agg = (torch.randn(csr.N, 128, device="cuda") for _ in range(3))
m2 = m.gnn_layers[0].message_net_2[0]
w16, w32t = pack_edge16_fp16x2(m2.weight, m2.bias), pack_edge32t_fp16x2(m2.weight, m2.bias)
plan = ops.stream_tiles(csr)
for _ in range(30):
    ops.mpnn_edge_stream(A, Bm, csr, plan, w32t[0], w16[1], agg, l2e=True)
torch.cuda.synchronize()
print("done")
