"""Dev: the two generations of magnet_linear_bwd128_f32 in ONE process (magnet_linear_bwd128_generation: 0 = k_linear_bwd<4>, 128-row steps,
one wave per SIMD; 1 = k_linear_bwd_r, 32-row steps, two waves per SIMD): agreement with fp64 and with each other on ragged sizes,
strided inputs, both mask settings, rows of very different magnitude; alternating timing at the E-level row counts of C5 / gnn_64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magnet_amd import _lib as L, ops
from magnet_amd.packing import pack

lib = L.lib()
NEW = int(os.environ.get('LB_NEW', '1'))        # the generation checked against 0 (2 = <128, 2>: four waves, two workgroups per CU)


def use(gen):
    lib.magnet_linear_bwd128_generation(gen)
    L._QUERIES.clear()                      # (magnet_linear_bwd128_grid follows the generation)


def run(gen, dz, x, wt, mask, amax_in=None):
    use(gen)
    M, Wd = dz.shape
    ws = torch.empty(lib.magnet_linear_bwd128_workspace(), device="cuda")
    dx, dW, db = torch.full((M, Wd), 7.0, device="cuda"), torch.empty(Wd, Wd, device="cuda"), torch.empty(Wd, device="cuda")
    am = ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, mask=mask, amax_in=amax_in)
    return dx, dW, db, am


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()


g = torch.Generator(device="cuda").manual_seed(3)
worst = 0.0
for WD in (128, 64):
    W = (torch.randn(WD, WD, device="cuda", generator=g) * 0.1)
    wt = pack(W.t().contiguous(), None, "fp16x3")[0]
    for M in (1, 5, 31, 32, 33, 100, 127, 129, 4097, 33000, 218752):
        for mask in (True, False):
            for variant in ("plain", "strided", "rows"):
                dz = torch.randn(M, WD, device="cuda", generator=g) * 1e-7
                xfull = torch.relu(torch.randn(M, 160, device="cuda", generator=g))
                x = xfull[:, 16:16 + WD] if variant == "strided" else xfull[:, :WD].contiguous()
                if variant == "rows":                                   # rows of very different magnitude (late rollout steps ~ 1e-4 of the early ones)
                    dz = dz * torch.logspace(0, -5, M, device="cuda")[:, None]
                ref_dx = (dz.double() @ W.double()) * ((x > 0).double() if mask else 1.0)
                ref_dW, ref_db = dz.double().t() @ x.double(), dz.double().sum(0)
                out = {}
                for gen in (0, NEW):
                    dx, dW, db, am = run(gen, dz, x, wt, mask)
                    e = (rel(dx, ref_dx), rel(dW, ref_dW), rel(db, ref_db))
                    amax_true = dx.abs().max().item()
                    am_ok = abs(am[0][:am[1]].max().item() - amax_true) <= 1e-12 + 1e-6 * amax_true
                    out[gen] = (dx, dW, db, e, am_ok)
                    # the maxima of one call feed the next: run again with amax_in = a previous layer's maxima of |dz|
                a0, a1 = out[0], out[NEW]
                flag = "" if max(a1[3]) < 2e-6 and a1[4] else "   <-- CHECK"
                worst = max(worst, *a1[3])
                print(f"W={WD:3d} M={M:7d} mask={int(mask)} {variant:8s} gen0 dx/dW/db {a0[3][0]:.1e} {a0[3][1]:.1e} {a0[3][2]:.1e} | gen1 {a1[3][0]:.1e} {a1[3][1]:.1e} "
                      f"{a1[3][2]:.1e} amax_ok {a0[4]} {a1[4]} | mask zeros identical {bool(((a0[0] == 0) == (a1[0] == 0)).all())}{flag}")
print("worst gen1 error", worst)
W = (torch.randn(128, 128, device="cuda", generator=g) * 0.1)
wt = pack(W.t().contiguous(), None, "fp16x3")[0]
# per-row relative error of dx on the 'rows' case (generation 0 scales every row by itself, generation 1 by the launch maximum)
M = 33000
dz = torch.randn(M, 128, device="cuda", generator=g) * 1e-7 * torch.logspace(0, -5, M, device="cuda")[:, None]
x = torch.relu(torch.randn(M, 128, device="cuda", generator=g))
ref = dz.double() @ W.double()
for gen in (0, NEW):
    dx = run(gen, dz, x, wt, False)[0]
    rr = (dx.double() - ref).norm(dim=1) / ref.norm(dim=1)
    print(f"gen{gen}: per-row relative error of dx: max {rr.max().item():.2e} at row {rr.argmax().item()}, median {rr.median().item():.2e}, rows 30000+: {rr[30000:].max().item():.2e}")
# chained amax: layer 2's amax_in = layer 1's amax_out
for gen in (0, NEW):
    dz = torch.randn(5000, 128, device="cuda", generator=g) * 1e-6
    x = torch.relu(torch.randn(5000, 128, device="cuda", generator=g))
    dx1, _, _, am = run(gen, dz, x, wt, True)
    dx2, dW2, db2, _ = run(gen, dx1, x, wt, True, amax_in=am)
    ref = ((dx1.double() @ W.double()) * (x > 0).double())
    print(f"gen{gen}: chained amax_in: dx {rel(dx2, ref):.1e} dW {rel(dW2, dx1.double().t() @ x.double()):.1e}")
# determinism
use(NEW)
dz = torch.randn(100000, 128, device="cuda", generator=g) * 1e-7
x = torch.relu(torch.randn(100000, 128, device="cuda", generator=g))
a = run(NEW, dz, x, wt, True)
bad = 0
for _ in range(20):
    b = run(NEW, dz, x, wt, True)
    bad += int(not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])))
print("gen1 repeat launches that differ:", bad, "of 20")
# timing (generations 2 / 3 = generation 1 with two / three steps of rows in flight)
GENS = tuple(int(v) for v in os.environ.get('LB_GENS', '0,1').split(','))
for WD, M in ((128, 2105344), (128, 218752), (128, 18432), (64, 2105344), (64, 163840), (64, 40960)):
    Wt = (torch.randn(WD, WD, device="cuda") * 0.1)
    wt = pack(Wt.t().contiguous(), None, "fp16x3")[0]
    dz = torch.randn(M, WD, device="cuda") * 1e-7
    x = torch.relu(torch.randn(M, WD, device="cuda"))
    ws = torch.empty(lib.magnet_linear_bwd128_workspace(), device="cuda")
    dx, dW, db = torch.empty(M, WD, device="cuda"), torch.empty(WD, WD, device="cuda"), torch.empty(WD, device="cuda")
    line = f"W={WD:3d} M={M:8d}:"
    for rep in range(3):
        for gen in GENS:
            use(gen)
            am = ops.linear_bwd128(dz, x, wt, dx, dW, db, ws)
            for _ in range(3):
                ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, amax_in=am)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, amax_in=am)
            e.record(); torch.cuda.synchronize()
            line += f"  gen{gen} {s.elapsed_time(e) / 20 * 1000:7.1f} us"
    print(line)
use(1)
