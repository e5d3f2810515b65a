"""Dev microbench: time the MPNN hop kernels in isolation on the 256^2 graph (B trajectories)."""
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")   # needs the development library: python -m magnet_amd.build --dev
import sys, types
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.factory import FACTORY

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
topo = m._build_topology(x0, B)
csr = topo.csr
N, E, H = csr.N, csr.E, 128
P = m._packed()
A, Bm, agg = (torch.randn(N, H, device="cuda") for _ in range(3))
partials = torch.empty(ops.edge_scratch_floats(E), device="cuda")
lyr = P.layers[0]

def timeit(fn, reps=10):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

ref = None
from magnet_amd.packing import pack
_m2 = m.gnn_layers[0].message_net_2[0]
w32 = pack(_m2.weight, _m2.bias, "fp32")
for v in variants:
    _lib.lib().magnet_debug_set_variant(0, v)
    agg.fill_(float("nan"))
    ms = timeit(lambda: ops.mpnn_edge(A, Bm, csr, w32[0], w32[1], agg, partials))
    if ref is None:
        ref = agg.clone()
    err = float((agg - ref).norm() / ref.norm())
    print(f"edge variant {v}: {ms:.3f} ms  {2*E*H*H/ms/1e9:.1f} TFLOP/s (incl fixup)  rel-diff vs first {err:.2e} nan={int(torch.isnan(agg).sum())}")
for v3 in (0, 6, 8, 0):
    _lib.lib().magnet_debug_set_variant(1, v3)
    agg.fill_(float("nan"))
    ms = timeit(lambda: ops.mpnn_edge(A, Bm, csr, lyr.msg2[0], lyr.msg2[1], agg, partials))
    err = float((agg - ref).norm() / ref.norm())
    print(f"edge kernel, default split mode, knob [{v3}]: {ms:.3f} ms  {2*E*H*H/ms/1e9:.1f} TFLOP/s-equivalent (incl fixup)  rel-diff vs first {err:.2e} nan={int(torch.isnan(agg).sum())}")
_lib.lib().magnet_debug_set_variant(1, 0)
agg.fill_(float("nan"))
A2, B2 = A * ops.LOG2E, Bm * ops.LOG2E
ms = timeit(lambda: ops.mpnn_edge(A2, B2, csr, lyr.msg2[0], lyr.msg2[1], agg, partials, l2e=True))
print(f"edge kernel, log2(e)-scaled inputs (what the model's inference path runs): {ms:.3f} ms  {2*E*H*H/ms/1e9:.1f} TFLOP/s-equivalent (incl fixup)  "
      f"rel-diff vs first {float((agg - ref).norm() / ref.norm()):.2e}")
from magnet_amd.packing import pack_edge16_fp16x2
w16 = pack_edge16_fp16x2(_m2.weight, _m2.bias)
for l2e in (False, True):
    agg.fill_(float("nan"))
    Ax, Bx = (A2, B2) if l2e else (A, Bm)
    ms = timeit(lambda: ops.mpnn_edge16(Ax, Bx, csr, w16[0], w16[1], agg, partials, l2e=l2e))
    print(f"edge kernel, 16-edge tiles, 2 waves/SIMD, l2e={l2e}: {ms:.3f} ms  {2*E*H*H/ms/1e9:.1f} TFLOP/s-equivalent (incl fixup)  "
          f"rel-diff vs first {float((agg - ref).norm() / ref.norm()):.2e} nan={int(torch.isnan(agg).sum())}")
h = torch.randn(N, H, device="cuda"); node_in = torch.randn(N, 32, device="cuda"); gbuf = torch.empty_like(h)
T = ops.Term
ms = timeit(lambda: ops.mlp(N, ops.ktiles(T(h, H), T(agg, H), T(node_in, 32)), [lyr.upd1, lyr.upd2], nl=2, ht=4, ot=4, act=ops.ACT_SWISH, last_act=True, epi=ops.EPI_STORE_RES, res=h, y=gbuf))
print(f"node update: {ms:.3f} ms  {2*N*(288*128+128*128)/ms/1e9:.1f} TFLOP/s")
ms = timeit(lambda: ops.mlp(N, ops.ktiles(T(h, H), T(node_in, 32)), [lyr.pre_a], nl=1, ht=4, ot=4, act=ops.ACT_NONE, y=A))
print(f"node pre (one of two): {ms:.3f} ms  {2*N*160*128/ms/1e9:.1f} TFLOP/s")
ms = timeit(lambda: ops.instance_norm(gbuf, B, N // B, out=h))
print(f"instance norm: {ms:.3f} ms")
msg = torch.randn(E, H, device="cuda"); outm = torch.empty(N, H, device="cuda")
ms = timeit(lambda: ops.segment_mean(msg, csr.rowptr, N, outm))
nbytes = E * H * 4 + N * H * 4 + 4 * (N + 1)
print(f"segment_mean<128>: {ms:.3f} ms  {nbytes/ms/1e6:.0f} GB/s  ({nbytes/1e9:.3f} GB algorithmic)")
