"""debug: k_mpnn_edge32t on one tile of single-edge rows, W2 = identity / random: where do the errors sit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import ops
from magnet_amd.packing import pack_edge32t_fp16x2

torch.manual_seed(0)
H = 128
sw = lambda x: x * torch.sigmoid(x)  # noqa: E731
for E, wkind in ((32, "identity"), (32, "random"), (13, "random"), (96, "random")):
    n = E
    rowptr = torch.arange(n + 1, dtype=torch.int32)
    tgt = torch.arange(E, dtype=torch.int32)
    src = torch.randperm(n)[:E].int()
    A, Bm = torch.randn(n, H), torch.randn(n, H)
    W = torch.eye(H) if wkind == "identity" else torch.randn(H, H) / 11
    b = torch.randn(H) * 0.1
    x = sw((A[tgt.long()] + Bm[src.long()]).double())
    want = sw(x @ W.double().T + b.double())
    csr = ops.CSR(rowptr.cuda(), src.cuda(), tgt.cuda(), n, E)
    w, bp = pack_edge32t_fp16x2(W.cuda(), b.cuda())
    agg = torch.full((n, H), float("nan"), device="cuda")
    partials = torch.zeros(ops.edge_scratch_floats(E), device="cuda")
    ops.mpnn_edge32t(A.cuda(), Bm.cuda(), csr, w, bp, agg, partials, l2e=False)
    got = agg.cpu().double()
    err = (got - want).abs()
    print(f"E={E} W={wkind}: rel {float((got - want).norm() / want.norm()):.3e}")
    print("  max err per 16-feature block:", [f"{float(err[:, i:i + 16].max()):.1e}" for i in range(0, H, 16)])
    print("  max err per edge:", [f"{float(err[e].max()):.0e}" for e in range(min(E, 32))])
    if wkind == "identity":   # which input column does each output feature carry?
        xin = x.float()
        inv = torch.log(got / (1 - got + 1e-30) + 1e-30)      # not invertible in general; instead match by brute force on edge 0
        e = 0
        cand = sw(xin[e].double()[None, :] + b.double()[:, None])       # [f, k]: value if feature f read column k
        k_of_f = (cand - got[e][:, None]).abs().argmin(1)
        print("  edge 0: output feature f carries input column:", k_of_f.tolist()[:64])
        print("                                               ", k_of_f.tolist()[64:])
