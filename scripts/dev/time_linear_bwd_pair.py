"""Dev: two hidden 128 -> 128 Linears walked back as two magnet_linear_bwd128_f32 launches against one magnet_linear_bwd128_pair_f32
(k_linear_bwd_r2) at the C5 edge-chain size (2.1 M rows) and at gnn_64's (218 k rows); operands from HBM (1 GiB written in between)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from magnet_amd import _lib as L, ops  # noqa: E402
from magnet_amd.packing import pack  # noqa: E402

ws = torch.empty(L.lib().magnet_linear_bwd128_workspace(), device="cuda")
big = torch.empty(1 << 28, device="cuda")
for M in (2105344, 218752):
    dz = torch.randn(M, 128, device="cuda") * 1e-7
    x1, x2 = torch.relu(torch.randn(M, 128, device="cuda")), torch.relu(torch.randn(M, 128, device="cuda"))
    WA, WB = torch.randn(128, 128, device="cuda") * 0.1, torch.randn(128, 128, device="cuda") * 0.1
    wtA, wtB = pack(WA.t().contiguous(), None, "fp16x3")[0], pack(WB.t().contiguous(), None, "fp16x3")[0]
    mid, dx = torch.empty(M, 128, device="cuda"), torch.empty(M, 128, device="cuda")
    dW = [torch.zeros(128, 128, device="cuda") for _ in range(2)]
    db = [torch.zeros(128, device="cuda") for _ in range(2)]
    amx = torch.zeros(256, device="cuda")
    amx[0] = dz.abs().max()

    def two():
        a = ops.linear_bwd128(dz, x1, wtA, mid, dW[0], db[0], ws, mask=True, amax_in=(amx, 256))
        ops.linear_bwd128(mid, x2, wtB, dx, dW[1], db[1], ws, mask=True, amax_in=a)

    def one():
        ops.linear_bwd128_pair(dz, x1, x2, wtA, wtB, dx, dW[0], db[0], dW[1], db[1], ws, amax_in=(amx, 256))

    def t(f, reps=10):
        for _ in range(2):
            f()
        tot = 0.0
        for _ in range(reps):
            big.fill_(1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            f()
            e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / reps * 1e3

    for r in range(2):
        print(f"M = {M}: two launches {t(two):.1f} us   pair {t(one):.1f} us")
