"""Dev microbench: ablations of k_mpnn_edge_rowsp (generation 13) on the 4 x 256^2 bench graph.  Builds mpnn_edge_rowsp.hip ALONE with
-DRP_ABL_BUILD into magnet_amd/csrc/variants/librowsp_abl.so when asked to (--build, CPU side), runs the ablations on the GPU otherwise.
Results of an ablated launch are INVALID; only the times mean something."""
import ctypes as C
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "magnet_amd", "csrc", "variants", "librowsp_abl.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(ROOT, "magnet_amd", "csrc", "mpnn_edge_rowsp.hip")
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-slp-vectorize", "-DRP_ABL_BUILD", *extra, src, "-o", SO])
    print("built", SO)
    sys.exit(0)
import torch
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack_edge16_fp16x2, pack_edge32t_fp16x2

hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128, hidden_layer=5, time_window=10,
                           teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
W, B = 256, 4
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, B).csr
N, E, H = csr.N, csr.E, 128
A, Bm, agg = (torch.randn(N, H, device="cuda") * ops.LOG2E for _ in range(3))
m2 = m.gnn_layers[0].message_net_2[0]
w16 = pack_edge16_fp16x2(m2.weight, m2.bias)
w32t = pack_edge32t_fp16x2(m2.weight, m2.bias)
plan = ops.row_tiles(csr)
partials = torch.empty(ops.edge_scratch_floats(E), device="cuda")
lib = C.CDLL(SO)
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


names = {0: "as shipped", 1: "no MFMA", 2: "no epilogue fillers", 4: "no production fillers", 6: "no fillers", 8: "no gather requests",
         16: "no operand fetches", 32: "no slot barriers", 14: "MFMA + fetch only", 30: "MFMA only", 31: "skeleton", 7: "fetch + gathers only",
         95: "skeleton - straggler parking", 159: "skeleton - A row hand-over", 287: "skeleton - A column reads", 543: "skeleton - finish / rare epilogues",
         1023: "bare loop", 1022: "bare loop + MFMA", 990: "bare loop + MFMA with slot barriers", 1006: "bare loop + MFMA + fetch",
         64: "full - straggler parking", 128: "full - A row hand-over", 512: "full - finish / rare epilogues",
         704: "full - all three"}
print(f"edge16 {timeit(lambda: ops.mpnn_edge16(A, Bm, csr, w16[0], w16[1], agg, partials, l2e=True)):.4f} ms   "
      f"rowsp (product) {timeit(lambda: ops.mpnn_edge_rows(A, Bm, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=True, wreg=3)):.4f} ms")
for abl, name in names.items():
    def run():
        rc = lib.magnet_mpnn_edge_rowsp_abl_f32(ops._p(A), ops._p(Bm), ops._p(csr.src), ops._p(plan.tiles), plan.n_tiles, ops._p(w32t[0]), ops._p(w16[1]),
                                                ops._p(agg), abl, ncu, ops._stream())
        assert rc == 0, rc
    print(f"  ablation {abl:2d} ({name}): {min(timeit(run) for _ in range(2)):.4f} ms", flush=True)
