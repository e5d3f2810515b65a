"""Dev: cycle trace (s_memtime) of one wave of the 5-Linear edge chain kernel.  Needs the development library:
python -m magnet_amd.build --dev.  usage: python scripts/dev/trace_chain.py [D] [H] [factored 0|1] [wg]"""
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")
import sys, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.magnet_common import _mlp_ln, pack_mlp_ln, pack_interaction_edge, run_mlp

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
fact = int(sys.argv[3]) if len(sys.argv) > 3 else 1
wg = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
N, deg = 65792, 32
E = N * deg
torch.manual_seed(0)
net = _mlp_ln(3 * D, H, 4, D).cuda()
x = torch.randn(N, D, device="cuda"); e0 = torch.randn(E, D, device="cuda")
tgt = torch.arange(N, device="cuda", dtype=torch.int32).repeat_interleave(deg)
src = torch.randint(0, N, (E,), device="cuda", dtype=torch.int32)
rowptr = torch.arange(0, E + 1, deg, device="cuda", dtype=torch.int32)
agg = torch.empty(N, D, device="cuda"); partials = torch.empty(((E + 31) // 32) * 2 * D + 1, device="cuda")
T_ = ops.Term
if fact:
    chain_for, pi, pj = pack_interaction_edge(net, D)
    chain = chain_for(1)
    PQ = torch.randn(2, N, H, device="cuda")
    run = lambda: run_mlp(E, ops.ktiles(T_(e0, D)), chain, agg, epi=ops.EPI_SEGMEAN, tgt=tgt, rowptr=rowptr, partials=partials, pre=[(PQ[0], tgt), (PQ[1], src)])
else:
    pk = pack_mlp_ln(net)
    run = lambda: run_mlp(E, ops.ktiles(T_(x, D, idx=tgt), T_(x, D, idx=src), T_(e0, D)), pk, agg, epi=ops.EPI_SEGMEAN, tgt=tgt, rowptr=rowptr, partials=partials)
for _ in range(3): run()
buf = torch.zeros(1000, dtype=torch.int64, device="cuda")
_lib.lib().magnet_debug_set_mlp_trace(C.c_void_p(buf.data_ptr()), wg)
run(); torch.cuda.synchronize()
_lib.lib().magnet_debug_set_mlp_trace(C.c_void_p(0), 0)
b = buf.cpu().view(-1, 2)
b = b[b[:, 1] > 0]
names = {1: "start", 10: "L0 iter top", 11: "L0 vmcnt done", 12: "L0 barrier done", 13: "L0 dma+gathers issued", 14: "L0 mma done",
         20: "Lx iter top", 21: "Lx vmcnt done", 22: "Lx barrier done", 23: "Lx dma issued", 24: "Lx mma done", 30: "chain done", 31: "LN done", 32: "epilogue done"}
t0 = int(b[0, 0]); prev = t0
agg_t = {}
for t, tag in b.tolist():
    agg_t[tag] = agg_t.get(tag, 0) + (t - prev)
    prev = t
print(f"workgroup {wg}, wave 0: total {int(b[-1, 0]) - t0} clocks (s_memtime @100MHz? see ratios), {len(b)} stamps")
for tag, dt in sorted(agg_t.items()):
    print(f"  time ending at [{names.get(tag, tag):24s}]: {dt:8d}  ({100 * dt / (int(b[-1, 0]) - t0):5.1f} %)")
