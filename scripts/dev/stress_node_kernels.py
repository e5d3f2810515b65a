"""Dev: determinism stress of k_node_halves / k_node_update / the generic chain variants (bitwise equality over many launches)."""
import sys, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
lyr = m._packed().layers[0]
H, kin = 128, m.kin
h, agg = torch.randn(N, H, device="cuda") * 2 + 1, torch.randn(N, H, device="cuda")
node_in = torch.randn(N, kin, device="cuda")
st = ops.instance_norm_stats(h, 4, N // 4)
A, B, g = (torch.empty(N, H, device="cuda") for _ in range(3))


def halves():
    ops.mpnn_node_halves(h, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, A, B, stats=st, rows_per_graph=N // 4)
    return torch.cat([A, B])


def update():
    ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, g, stats=st, rows_per_graph=N // 4)
    return g


for name, fn in (("halves", halves), ("update", update)):
    ref = fn().clone()
    bad = 0
    for r in range(reps):
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
            d = (out - ref).abs().amax(1)
            print(name, r, "rows differing", int((d > 0).sum()), "first", int((d > 0).nonzero()[0]))
    print(name, "launches differing from the first:", bad, "of", reps)
