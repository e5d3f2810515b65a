"""Dev microbench: magnet_linear_bwd128_f32 / 64 at the E-level row count of C5 / C4b"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import _lib as L, ops
from magnet_amd.packing import pack
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2105344
for Wd in (128, 64):
    g = torch.Generator().manual_seed(1)
    W = (torch.randn(Wd, Wd, generator=g) * 0.1).cuda()
    x = torch.relu(torch.randn(M, Wd, device="cuda"))
    dz = torch.randn(M, Wd, device="cuda") * 1e-7
    wt = pack(W.t().contiguous(), None, "fp16x3")[0]
    ws = torch.empty(L.lib().magnet_linear_bwd128_workspace(), device="cuda")
    dx, dW, db = torch.empty(M, Wd, device="cuda"), torch.empty(Wd, Wd, device="cuda"), torch.empty(Wd, device="cuda")
    amax = ops.linear_bwd128(dz, x, wt, dx, dW, db, ws)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, amax_in=amax)
    s.record()
    for _ in range(10):
        ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, amax_in=amax)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"W={Wd} M={M}: {ms:.3f} ms  ({3 * M * Wd * 4 / ms / 1e6:.0f} GB/s of the 3 arrays)")
