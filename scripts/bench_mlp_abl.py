"""Dev microbench: run-time ablations of the 5-Linear InteractionNetwork edge kernel (C5 joint graph shape)."""
__import__("os").environ.setdefault("MAGNET_DEV_LIB", "1")   # needs the development library: python -m magnet_amd.build --dev
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from magnet_amd import ops, _lib
from magnet_amd.models.magnet_common import _mlp_ln, pack_mlp_ln, run_mlp

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N, deg = 65792, 32
E = N * deg
torch.manual_seed(0)
net = _mlp_ln(3 * D, H, 4, D).cuda()
pk = pack_mlp_ln(net)
x = torch.randn(N, D, device="cuda")
e0 = torch.randn(E, D, device="cuda")
tgt = torch.arange(N, device="cuda", dtype=torch.int32).repeat_interleave(deg)
src = torch.randint(0, N, (E,), device="cuda", dtype=torch.int32)
rowptr = torch.arange(0, E + 1, deg, device="cuda", dtype=torch.int32)
agg = torch.empty(N, D, device="cuda")
partials = torch.empty(((E + 31) // 32) * 2 * D + 1, device="cuda")
T_ = ops.Term
kts = ops.ktiles(T_(x, D, idx=tgt), T_(x, D, idx=src), T_(e0, D))


def run():
    run_mlp(E, kts, pk, agg, epi=ops.EPI_SEGMEAN, tgt=tgt, rowptr=rowptr, partials=partials)


def timeit(reps=5):
    for _ in range(2): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


flops = 2.0 * E * (3 * D * H + 3 * H * H + H * D)
for abl in [0, 1, 2, 4, 8, 16, 3, 1 | 2 | 16, 1 | 2 | 4 | 16, 31, 0]:
    _lib.lib().magnet_debug_set_mlp_ablation(abl)
    ms = timeit()
    print(f"abl {abl:2d}: {ms:.3f} ms  {flops/ms/1e9:.0f} TFLOP/s-equivalent")
_lib.lib().magnet_debug_set_mlp_ablation(0)
