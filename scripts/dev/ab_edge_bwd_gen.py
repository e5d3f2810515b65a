"""Same-process alternating A/B of the two generations of the fused edge backward with in-kernel weight gradient
(magnet_mpnn_edge_bwd_w_generation: 0 = shared slabs, 1 = wave-private accumulators), 256^2 x 4 graph, + agreement of all outputs."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
L = _lib.lib()
out = {}
for gen in (0, 1):
    L.magnet_mpnn_edge_bwd_w_generation(gen)
    dp, dA = torch.zeros(E, H, device="cuda"), torch.zeros(N, H, device="cuda")
    gW, gb = torch.zeros(H, H, device="cuda"), torch.zeros(H, device="cuda")
    ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW, gb, accumulate=False)
    torch.cuda.synchronize()
    out[gen] = (dp, dA, gW, gb)
for name, a, b in zip(("dp", "dA", "dW2", "db2"), out[0], out[1]):
    print(f"{name}: gen1 vs gen0  max|diff| / max|ref| = {float((a - b).abs().max() / a.abs().max()):.3e}   bitwise {bool((a == b).all())}")
# against the unfused kernels (s1 / dq in memory + k_wgrad_split)
s1, dq, dp0 = (torch.empty(E, H, device="cuda") for _ in range(3))
dA0 = torch.empty(N, H, device="cuda")
ops.mpnn_edge_bwd(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], s1, dq, dp0, dA0, partials)
gWr = (dq.double().t() @ s1.double()).float()
gbr = dq.double().sum(0).float()
for gen in (0, 1):
    print(f"gen {gen}: dW2 vs fp64 dq^T s1: {float((out[gen][2] - gWr).abs().max() / gWr.abs().max()):.3e}   db2: "
          f"{float((out[gen][3] - gbr).abs().max() / gbr.abs().max()):.3e}")
del s1, dq, dp0


def timeit(fn, reps=5):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


dp, dA, gW, gb = out[1]
ws = torch.empty(L.magnet_mpnn_edge_bwd_w_workspace(), device="cuda")
for rnd in range(4):
    for gen in (0, 1):
        L.magnet_mpnn_edge_bwd_w_generation(gen)
        ms = timeit(lambda: ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW, gb, ws, accumulate=False))
        print(f"round {rnd} generation {gen}: {ms:.3f} ms")
# repeatability of generation 1 (argv[3] launches, default 5: every output bit for bit)
L.magnet_mpnn_edge_bwd_w_generation(1)
ref = None
n_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bad = 0
for _ in range(n_rep):
    ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW, gb, ws, accumulate=False)
    cur = (dp.clone(), dA.clone(), gW.clone(), gb.clone())
    if ref is None: ref = cur
    else:
        same = all(bool((a == b).all()) for a, b in zip(ref, cur))
        bad += not same
        if n_rep <= 5: print("repeat bitwise:", same)
print(f"{n_rep} launches of generation 1: {bad} differ from the first")
