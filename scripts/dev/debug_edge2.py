import sys, types
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")   # needs the development library: python -m magnet_amd.build --dev
sys.path.insert(0, ".")
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
W = 10
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
topo = m._build_topology(x0, 1)
csr = topo.csr
N, E, H = csr.N, csr.E, 128
P = m._packed(); lyr = P.layers[0]
A, Bm = torch.randn(N, H, device="cuda"), torch.randn(N, H, device="cuda")
res = {}
for v in (0, 100):
    _lib.lib().magnet_debug_set_variant(0, v)
    agg = torch.full((N, H), float("nan"), device="cuda"); partials = torch.zeros(((E + 31) // 32) * 2 * H, device="cuda")
    ops.mpnn_edge(A, Bm, csr, lyr.msg2[0], lyr.msg2[1], agg, partials)
    res[v] = agg.cpu(); res[('p', v)] = partials.cpu().view(-1, 2, 128)
d = (res[0] - res[100]).abs().max(1).values
rp = csr.rowptr.cpu()
print("E", E, "N", N)
for r in range(min(N, 40)):
    print(r, int(rp[r]), int(rp[r + 1]), "tiles", int(rp[r]) // 32, (int(rp[r + 1]) - 1) // 32, "maxdiff %.3e" % float(d[r]), "ratio %.3f" % float((res[100][r, 0] / res[0][r, 0])))

p0, p1 = res[('p', 0)], res[('p', 100)]
for tl in range(3):
    for sl in range(2):
        print("tile", tl, "slot", sl, "v1", p0[tl, sl, :4].tolist(), "v2", p1[tl, sl, :4].tolist(), "v2[32:36]", p1[tl, sl, 32:36].tolist(), "v1[32:36]", p0[tl, sl, 32:36].tolist())
