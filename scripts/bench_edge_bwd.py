"""Dev microbench: the fused backward edge kernel and its follow-up kernels in isolation on the 256^2 graph."""
import sys, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
abls = [0, 1, 2, 3, 4] if _lib.DEV else [0]   # MAGNET_DEV_LIB=1: store ablations (include/magnet_hip_dev.h)
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
topo = m._build_topology(x0, B)
csr = topo.csr
N, E, H = csr.N, csr.E, 128
print(f"N={N} E={E}")
A, Bm = torch.randn(N, H, device="cuda"), torch.randn(N, H, device="cuda")
dagg = torch.randn(N, H, device="cuda") * 1e-7
l2 = m.gnn_layers[0].message_net_2[0]
w2, w2t = pack(l2.weight, l2.bias, "fp16x3"), pack(l2.weight.t(), None, "fp16x3")
s1, dq, dp = (torch.empty(E, H, device="cuda") for _ in range(3))
dA, dBm = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
partials = torch.empty(((E + 31) // 32) * 2 * H + 1, device="cuda")


def timeit(fn, reps=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for a in abls:
    if _lib.DEV:
        _lib.lib().magnet_debug_set_edge_bwd_ablation(a)
    ms = timeit(lambda: ops.mpnn_edge_bwd(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], s1, dq, dp, dA, partials))
    print(f"edge_bwd ablation {a:2d}: {ms:.3f} ms   ({4*E*H*H/ms/1e9:.1f} TFLOP/s fp32-equivalent, {3*E*H*4/ms/1e6:.0f} GB/s written)")
ws = ops.wgrad_workspace(E, "cuda")
gW, gb = torch.zeros(H, H, device="cuda"), torch.zeros(H, device="cuda")
if _lib.DEV:
    _lib.lib().magnet_debug_set_edge_bwd_ablation(0)
gW2, gb2 = torch.zeros(H, H, device="cuda"), torch.zeros(H, device="cuda")
ms = timeit(lambda: ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], dp, dA, partials, gW2, gb2, accumulate=False))
print(f"edge_bwd_w (dW2 / db2 in the kernel, no s1 / dq): {ms:.3f} ms   ({E*H*4/ms/1e6:.0f} GB/s written)")
ops.mpnn_edge_bwd(A, Bm, dagg, csr, w2[0], w2t[0], w2[1], s1, dq, dp, dA, partials)
ops.wgrad(dq, s1, 0, H, gW, 0, gb, ws, accumulate=False, M=E)
print("dW2 fused vs separate: rel", float((gW2 - gW).norm() / gW.norm()), " db2:", float((gb2 - gb).norm() / gb.norm()))
ms = timeit(lambda: ops.wgrad(dq, s1, 0, H, gW, 0, gb, ws, M=E))
print(f"wgrad (E rows): {ms:.3f} ms   {2*E*H*H/ms/1e9:.1f} TFLOP/s, {2*E*H*4/ms/1e6:.0f} GB/s read")
bs = ops.csr_by_target(csr.tgt, csr.src, N)
ms = timeit(lambda: ops.segment_sum128(dp, bs.rowptr, N, perm=bs.perm, out=dBm))
print(f"segment_sum by source: {ms:.3f} ms   {E*H*4/ms/1e6:.0f} GB/s read")
h = torch.randn(N, H, device="cuda")
ms = timeit(lambda: ops.wgrad(dA, h, 0, H, gW, 0, gb, ws))
print(f"wgrad (N rows): {ms:.3f} ms")
