"""Dev: long repeat-launch stress of k_linear_bwd_r (one barrier per step over double-buffered LDS operands: a race would show as a rare
mismatch): every shape x several row counts x N launches, each compared bit for bit with the first launch of its configuration;
other kernels run in between on a second stream to vary the timing.  usage: python scripts/dev/stress_linear_bwd.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import _lib as L, ops
from magnet_amd.packing import pack

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = L.lib()
g = torch.Generator(device="cuda").manual_seed(11)
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device="cuda")
total = bad = 0
for W, gens in ((128, (1, 2, 3)), (64, (1,))):
    Wt = torch.randn(W, W, device="cuda", generator=g) * 0.1
    wt = pack(Wt.t().contiguous(), None, "fp16x3")[0]
    ws = torch.empty(lib.magnet_linear_bwd128_workspace(), device="cuda")
    for M in (33, 4097, 70001, 218752, 1048576 + 17):
        dz = torch.randn(M, W, device="cuda", generator=g) * 1e-7
        x = torch.relu(torch.randn(M, W, device="cuda", generator=g))
        for gen in gens:
            lib.magnet_linear_bwd128_generation(gen)
            L._QUERIES.clear()
            ref = None
            n = N if M < 500000 else max(20, N // 10)
            for it in range(n):
                if it % 3 == 0:
                    with torch.cuda.stream(side):
                        noise_a @ noise_a                      # (a GEMM on another stream: different co-residency each time)
                dx, dW, db = torch.empty(M, W, device="cuda"), torch.empty(W, W, device="cuda"), torch.empty(W, device="cuda")
                am, k = ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, mask=True)
                cur = (dx, dW, db, am[:k].clone())
                if ref is None:
                    ref = cur
                else:
                    total += 1
                    bad += int(not all(torch.equal(a, b) for a, b in zip(ref, cur)))
            print(f"W={W} M={M} generation {gen}: {n} launches, mismatches so far {bad} of {total}", flush=True)
lib.magnet_linear_bwd128_generation(1)
print(f"TOTAL: {bad} of {total} repeat launches differ from the first launch of their configuration")
