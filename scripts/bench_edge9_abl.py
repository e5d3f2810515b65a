"""Dev (needs `python -m magnet_amd.build --dev`, run with MAGNET_DEV_LIB=1): where does k_mpnn_edge32t's time go?  Run-time
ablations (INVALID results) on the 4 x 256^2 bench graph.  usage: MAGNET_DEV_LIB=1 python scripts/bench_edge9_abl.py"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MAGNET_DEV_LIB"] = "1"
import torch
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack_edge32t_fp16x2

W, B = 256, 4
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, B).csr
N, E, H = csr.N, csr.E, 128
A, Bm, agg = (torch.randn(N, H, device="cuda") for _ in range(3))
partials = torch.empty(ops.edge_scratch_floats(E), device="cuda")
m2 = m.gnn_layers[0].message_net_2[0]
w = pack_edge32t_fp16x2(m2.weight, m2.bias)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


names = {0: "as shipped", 1: "no MFMA", 2: "no epilogue exp/rcp", 4: "no input Swish + split", 6: "no Swish at all (2+4)",
         7: "no MFMA, no Swish (1+2+4)", 8: "L1-resident gathers", 16: "no LDS weight reads", 17: "no MFMA, no LDS reads",
         24: "L1 gathers + no LDS reads", 32: "no gather loads at all", 33: "no gathers, no MFMA", 48: "no gathers, no LDS reads",
         63: "skeleton without gathers", 30: "everything but the MFMAs removed (2+4+8+16)", 31: "skeleton (all removed)"}
for abl, name in names.items():
    ops.L.lib().magnet_debug_set_variant(2, abl)
    ms = min(timeit(lambda: ops.mpnn_edge32t(A, Bm, csr, w[0], w[1], agg, partials, l2e=True)) for _ in range(3))
    print(f"ABL {abl:2d} {name:46s}: {ms:.4f} ms")
ops.L.lib().magnet_debug_set_variant(2, 0)

# ---- k_mpnn_edge_ws: which role sets the step time?  (1 = P idle, 2 = M idle, 4 = E idle)
for abl, name in ((0, "as shipped"), (6, "P only"), (5, "M only"), (3, "E only"), (4, "P + M"), (2, "P + E"), (1, "M + E"), (7, "barriers only")):
    ops.L.lib().magnet_debug_set_variant(3, abl)
    ms = min(timeit(lambda: ops.mpnn_edge_ws(A, Bm, csr, w[0], w[1], agg, partials, l2e=True)) for _ in range(3))
    print(f"WS ABL {abl} {name:14s}: {ms:.4f} ms")
ops.L.lib().magnet_debug_set_variant(3, 0)

# ---- cycle trace of workgroup 0 (development build): where does a step go, per role?
import ctypes as C
lib = ops.L.lib()
lib.magnet_debug_ws_trace.restype = C.c_int
buf = (C.c_longlong * 32)()
ops.mpnn_edge_ws(A, Bm, csr, w[0], w[1], agg, partials, l2e=True)
torch.cuda.synchronize()
lib.magnet_debug_ws_trace(None, 1)
ops.mpnn_edge_ws(A, Bm, csr, w[0], w[1], agg, partials, l2e=True)
torch.cuda.synchronize()
lib.magnet_debug_ws_trace(buf, 0)
steps = (E + 31) // 32 / 512
names = {0: "P barrier+loop", 1: "P rows wait + Swish (slots 0-7)", 2: "P pieces + LDS writes", 3: "P requests", 5: "P rows wait + Swish (8-15)",
         6: "P pieces + LDS writes", 7: "P requests", 8: "M barrier+loop", 9: "M frag reads + MFMA issue", 10: "M acc sum + Y stores",
         16: "E barrier+loop", 17: "E meta", 18: "E 32 edges"}
for k, nm in names.items():
    print(f"trace {k:2d} {nm:34s}: {buf[k] / steps:8.0f} cycles per step")
