"""Dev microbench: the 5-Linear chain kernel with two accumulators per output block (fp16x3) against one (fp16x3s) - time and
error against an fp64 evaluation of the same chain.   python scripts/bench_chain_single.py [D H [deg]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import ops
from magnet_amd.models.magnet_common import _mlp_ln, pack_mlp_ln, run_mlp

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
deg = int(sys.argv[3]) if len(sys.argv) > 3 else 32
N = 65792
E = N * deg
torch.manual_seed(0)
net = _mlp_ln(3 * D, H, 4, D).cuda()
packs = {}
for mode in ("fp16x3", "fp16x3s") if H == 64 else ("fp16x3",):
    os.environ["MAGNET_CHAIN_GEMM"] = mode
    packs[mode] = pack_mlp_ln(net)
x = torch.randn(N, D, device="cuda")
e0 = torch.randn(E, D, device="cuda")
tgt = torch.arange(N, device="cuda", dtype=torch.int32).repeat_interleave(deg)
src = torch.randint(0, N, (E,), device="cuda", dtype=torch.int32)
rowptr = torch.arange(0, E + 1, deg, device="cuda", dtype=torch.int32)
partials = torch.empty(((E + 31) // 32) * 2 * D + 1, device="cuda")
T_ = ops.Term
kts = ops.ktiles(T_(x, D, idx=tgt), T_(x, D, idx=src), T_(e0, D))
kts1 = ops.ktiles(T_(e0, D))


def timeit(run, reps=20):
    t0 = __import__("time").time()
    while __import__("time").time() - t0 < 0.7:            # a GPU that idled through start-up ramps its clock over several 100 ms
        run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


# fp64 reference on a slice of rows (plain store epilogue)
ns = 1 << 16
rows = slice(0, ns)
with torch.no_grad():
    xin = torch.cat([x[tgt[rows].long()], x[src[rows].long()], e0[rows]], 1).double()
    lins = net[0].linears()
    hcur = xin
    for i, lin in enumerate(lins):
        hcur = hcur @ lin.weight.double().t() + lin.bias.double()
        if i + 1 < len(lins):
            hcur = torch.relu(hcur)
    ref = torch.nn.functional.layer_norm(hcur, (D,), net[1].weight.double(), net[1].bias.double(), net[1].eps)
for mode, pk in packs.items():
    agg = torch.empty(N, D, device="cuda")
    ms = timeit(lambda: run_mlp(E, kts, pk, agg, epi=ops.EPI_SEGMEAN, tgt=tgt, rowptr=rowptr, partials=partials))
    y = torch.empty(E, D, device="cuda")
    ms_store = timeit(lambda: run_mlp(E, kts, pk, y, epi=ops.EPI_STORE))
    ms_k1 = timeit(lambda: run_mlp(E, kts1, pk, y, epi=ops.EPI_STORE)) if False else float("nan")
    err = float((y[rows].double() - ref).norm() / ref.norm())
    print(f"{mode:8s} D={D} H={H} E={E}: CSR-mean epilogue {ms:.3f} ms, store epilogue {ms_store:.3f} ms, rel-L2 vs fp64 {err:.2e}")
