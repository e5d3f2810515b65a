"""Dev: which output of magnet_mpnn_edge_bwd_w_f32 is off (dp / dA = the data path, dW2 / db2 = the in-kernel weight gradient)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import ops
from magnet_amd.packing import pack
torch.manual_seed(11)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for N, E in ((9, 20), (64, 128), (500, 7001)):
    src = torch.randint(0, N, (E,), dtype=torch.int32, device="cuda")
    tgt = torch.randint(0, max(N - 3, 1), (E,), dtype=torch.int32, device="cuda")
    csr = ops.csr_by_target(src, tgt, N)
    A, Bm = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
    W2, b2 = torch.randn(128, 128, device="cuda") / 11, torch.randn(128, device="cuda") * 0.1
    dagg = torch.randn(N, 128, device="cuda") * 1e-7
    w2p, w2tp = pack(W2, b2, "fp16x3"), pack(W2.t(), None, "fp16x3")
    partials = torch.empty(((E + 31) // 32) * 2 * 128 + 1, device="cuda")
    s1, dq, dp0 = (torch.empty(E, 128, device="cuda") for _ in range(3))
    dA0 = torch.empty(N, 128, device="cuda")
    ops.mpnn_edge_bwd(A, Bm, dagg, csr, w2p[0], w2tp[0], w2p[1], s1, dq, dp0, dA0, partials)
    gW0, gb0 = torch.zeros(128, 128, device="cuda"), torch.zeros(128, device="cuda")
    ops.wgrad(dq, s1, 0, 128, gW0, 0, gb0, ops.wgrad_workspace(E, "cuda"), accumulate=False)
    dp, dA = torch.empty(E, 128, device="cuda"), torch.empty(N, 128, device="cuda")
    gW, gb = torch.zeros(128, 128, device="cuda"), torch.zeros(128, device="cuda")
    ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2p[0], w2tp[0], w2p[1], dp, dA, partials, gW, gb, accumulate=False)
    torch.cuda.synchronize()
    print(f"N={N} E={E}: dp {rel(dp, dp0):.2e} dA {rel(dA, dA0):.2e} dW2 {rel(gW, gW0):.2e} db2 {rel(gb, gb0):.2e}")
    blk = lambda m: [[round(rel(m[32 * a:32 * a + 32, 32 * b:32 * b + 32], gW0[32 * a:32 * a + 32, 32 * b:32 * b + 32]), 3) for b in range(4)] for a in range(4)]
    print("   dW2 error by 32x32 block (row = dq block):", blk(gW))
    print("   db2 error by block:", [round(rel(gb[32 * a:32 * a + 32], gb0[32 * a:32 * a + 32]), 3) for a in range(4)])
