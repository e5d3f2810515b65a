"""Dev A/B: generations of the fused MPNN edge backward (magnet_mpnn_edge_bwd_w_generation: 1 = k_mpnn_edge_bwd_wp, 2 = k_mpnn_edge_bwd_co)
on the metric graph (4 x 256^2), alternating in one process; the results of the two against each other and against fp64 on a sample.
usage: python scripts/dev/ab_edge_bwd_gen3.py [W] [B] [rounds]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from magnet_amd import _lib, ops  # noqa: E402
from magnet_amd.models.factory import FACTORY  # noqa: E402
from magnet_amd.packing import pack  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, B).csr
N, E, H = csr.N, csr.E, 128
print(f"N={N} E={E}")
A, Bm = torch.randn(N, H, device="cuda"), torch.randn(N, H, device="cuda")
dagg = torch.randn(N, H, device="cuda") * 1e-7
l2 = m.gnn_layers[0].message_net_2[0]
w2, w2t = pack(l2.weight, l2.bias, "fp16x3"), pack(l2.weight.t(), None, "fp16x3")
partials = torch.empty(((E + 31) // 32) * 2 * H + 1, device="cuda")
lib = _lib.lib()
ws = torch.empty(lib.magnet_mpnn_edge_bwd_w_workspace(), device="cuda")


def run(gen):
    lib.magnet_mpnn_edge_bwd_w_generation(gen)
    dp, dA = torch.empty(E, H, device="cuda"), torch.empty(N, H, device="cuda")
    gW, gb = torch.zeros(H, H, device="cuda"), torch.zeros(H, device="cuda")
    ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW, gb, ws=ws, accumulate=False)
    return dp, dA, gW, gb


def timeit(gen, reps=10):
    lib.magnet_mpnn_edge_bwd_w_generation(gen)
    dp, dA = torch.empty(E, H, device="cuda"), torch.empty(N, H, device="cuda")
    gW, gb = torch.zeros(H, H, device="cuda"), torch.zeros(H, device="cuda")
    fn = lambda: ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW, gb, ws=ws, accumulate=False)  # noqa: E731
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))  # noqa: E731
r1, r2 = run(1), run(2)
print("generation 3 (co) vs 2 (wp): dp %.2e  dA %.2e  dW2 %.2e  db2 %.2e" % tuple(rel(a, b) for a, b in zip(r2, r1)))
r2b = run(2)
print("repeat bit-identical:", all(torch.equal(a, b) for a, b in zip(r2, r2b)))
# fp64 on the first 4000 edges' rows (dp only) and on the whole graph for dW2 through a chunked fp64 evaluation
W64, b64 = l2.weight.double(), l2.bias.double()
sw = lambda v: v * torch.sigmoid(v)  # noqa: E731
dsw = lambda v: torch.sigmoid(v) * (1 + v * (1 - torch.sigmoid(v)))  # noqa: E731
deg = (csr.rowptr[1:] - csr.rowptr[:-1]).double().clamp(min=1)
gW64, gb64 = torch.zeros(H, H, dtype=torch.float64, device="cuda"), torch.zeros(H, dtype=torch.float64, device="cuda")
dA64 = torch.zeros(N, H, dtype=torch.float64, device="cuda")
worst_dp = 0.0
for c0 in range(0, E, 1 << 20):
    sl = slice(c0, min(E, c0 + (1 << 20)))
    t, s = csr.tgt[sl].long(), csr.src[sl].long()
    p = A[t].double() + Bm[s].double()
    s1 = sw(p)
    q = s1 @ W64.t() + b64
    dq = dagg[t].double() / deg[t][:, None] * dsw(q)
    gW64 += dq.t() @ s1
    gb64 += dq.sum(0)
    dp64 = (dq @ W64) * dsw(p)
    dA64.index_add_(0, t, dp64)
    worst_dp = max(worst_dp, rel(r2[0][sl], dp64))
print("generation 3 vs fp64: dp (worst chunk) %.2e  dA %.2e  dW2 %.2e  db2 %.2e" % (worst_dp, rel(r2[1], dA64), rel(r2[2], gW64), rel(r2[3], gb64)))
print("generation 2 vs fp64:                      dA %.2e  dW2 %.2e  db2 %.2e" % (rel(r1[1], dA64), rel(r1[2], gW64), rel(r1[3], gb64)))
for r in range(R):
    a, b = timeit(1), timeit(2)
    print(f"round {r}: wp {a:.3f} ms   co {b:.3f} ms   ({3 * 2 * E * H * H / b / 1e9:.0f} TFLOP/s fp32-equivalent, frac of 833 = {3 * 2 * E * H * H / b / 1e9 / 833:.3f})")
lib.magnet_mpnn_edge_bwd_w_generation(2)
