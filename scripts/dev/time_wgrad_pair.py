"""Dev: dY^T [h | node_in] at 262 144 rows as two launches (magnet_wgrad_f32 x 2) against one (magnet_wgrad_pair_f32)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from magnet_amd import ops  # noqa: E402

M = 262144
dY = torch.randn(M, 128, device="cuda") * 1e-7
X, X2 = torch.randn(M, 128, device="cuda"), torch.randn(M, 32, device="cuda")
dW, db = torch.zeros(128, 288, device="cuda"), torch.zeros(128, device="cuda")
ws = ops.wgrad_workspace(M, "cuda")
big = torch.empty(1 << 28, device="cuda")      # 1 GiB written between repetitions: the operands come from HBM as in the step


def two():
    ops.wgrad(dY, X, 0, 128, dW, 0, db, ws)
    ops.wgrad(dY, X2, 0, 32, dW, 256, None, ws)


def one():
    ops.wgrad_pair(dY, X, 0, X2, 0, dW, 0, 256, db, ws)


def t(f, reps=20):
    for _ in range(3):
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


for r in range(3):
    print(f"two launches {t(two):.1f} us   pair {t(one):.1f} us")
