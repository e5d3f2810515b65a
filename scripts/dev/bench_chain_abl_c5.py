"""Dev (development library MAGNET_DEV_LIB=1: run-time flags; or MAGNET_LIB_PATH=<a -DMLP_ABL_STATIC=n build of the product kernel>): run-time ablations of the C5 E-level chain exactly as the rollout launches it - the FACTORED
InteractionNetwork edge chain (5 Linears over e_0 alone, the two per-node products gathered into the first pre-activation, LayerNorm + CSR
mean) on the 65 792-node joint graph.  Flags (csrc/mlp_bf16_kernels.hip g_mlp_abl): 1 no MFMAs, 2 no weight DMA, 4 no epilogue, 8 NO
BARRIERS, 16 no input loads.  Timing only - results invalid."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from magnet_amd import _lib, ops  # noqa: E402
from magnet_amd.models.magnet_common import _mlp_ln, pack_interaction_edge, run_mlp  # noqa: E402

D = H = 128
N, deg = 65792, 32
E = N * deg
torch.manual_seed(0)
net = _mlp_ln(3 * D, H, 4, D).cuda()
chain_for, pre_i, pre_j = pack_interaction_edge(net, D)
pk = chain_for(1.0)
x = torch.randn(N, D, device="cuda")
e0 = torch.randn(E, D, device="cuda")
tgt = torch.arange(N, device="cuda", dtype=torch.int32).repeat_interleave(deg)
src = torch.randint(0, N, (E,), device="cuda", dtype=torch.int32)
rowptr = torch.arange(0, E + 1, deg, device="cuda", dtype=torch.int32)
agg = torch.empty(N, D, device="cuda")
partials = torch.empty(((E + 31) // 32) * 2 * D + 1, device="cuda")
PQ = torch.randn(2, N, H, device="cuda")
kts = ops.ktiles(ops.Term(e0, D))


TAPE = "--tape" in sys.argv          # the training forward: four hidden activations + the pre-LayerNorm row written per edge (2 560 B)
hid = [torch.empty(E, H, device="cuda") for _ in range(5)] if TAPE else None


def run():
    run_mlp(E, kts, pk, agg, epi=ops.EPI_SEGMEAN, tgt=tgt, rowptr=rowptr, partials=partials, pre=[(PQ[0], tgt), (PQ[1], src)], hid=hid)


def timeit(reps=10):
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


flops = 2.0 * E * (D * H + 3 * H * H + H * D)
if not _lib.DEV:                      # a compile-time ablation build (-DMLP_ABL_STATIC=..., MAGNET_LIB_PATH) or the product library
    for rnd in range(3):
        ms = timeit()
        print(f"{os.path.basename(_lib.LIB_PATH)}{' tape' if TAPE else ''}: {ms:.4f} ms  {flops / ms / 1e9:.0f} TFLOP/s fp32-equivalent = {flops / ms / 1e9 / 833:.3f} of 833")
    sys.exit(0)
names = {0: "as shipped", 8: "no barriers", 2: "no weight DMA", 10: "no barriers, no weight DMA", 1: "no MFMAs", 4: "no epilogue", 16: "no input loads",
         31: "everything off"}
for rnd in range(2):
    for abl in [0, 8, 2, 10, 1, 4, 16, 31, 0]:
        _lib.lib().magnet_debug_set_mlp_ablation(abl)
        ms = timeit()
        print(f"abl {abl:2d} ({names[abl]:26s}): {ms:.4f} ms  {flops / ms / 1e9:.0f} TFLOP/s fp32-equivalent = {flops / ms / 1e9 / 833:.3f} of 833")
_lib.lib().magnet_debug_set_mlp_ablation(0)
