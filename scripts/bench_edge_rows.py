"""Dev microbench: the row-aligned edge kernel (k_mpnn_edge_rows32 + the remainder pass) against k_mpnn_edge16 on the 256^2 bench graph,
alternating rounds in one process.  usage: python scripts/bench_edge_rows.py [W] [B] [rounds]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import ops
from magnet_amd.models.factory import FACTORY
from magnet_amd.packing import pack, pack_edge16_fp16x2, pack_edge32t_fp16x2

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hp = types.SimpleNamespace(lr=1e-3, weight_decay=0, factor=0.3, step_size=50, loss="l1", hidden_features=128,
                           hidden_layer=5, time_window=10, teacher_forcing=False, neighbors=4)
torch.manual_seed(0)
m = FACTORY["mpnn_2d"](hp).cuda().eval()
g = torch.linspace(0, 1, W)
x0 = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2).cuda()
csr = m._build_topology(x0, B).csr
N, E, H = csr.N, csr.E, 128
A, Bm, agg = (torch.randn(N, H, device="cuda") for _ in range(3))
partials = torch.empty(ops.edge_scratch_floats(E), device="cuda")
m2 = m.gnn_layers[0].message_net_2[0]
w32f = pack(m2.weight, m2.bias, "fp32")
w16 = pack_edge16_fp16x2(m2.weight, m2.bias)
w32t = pack_edge32t_fp16x2(m2.weight, m2.bias)


FLUSH = int(os.environ.get("FLUSH", "0"))          # MB written between two timed launches (cold L2 / MALL, as inside a rollout)
junk = torch.empty(max(FLUSH, 1) * 1024 * 256, device="cuda") if FLUSH else None


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    if FLUSH:                                        # every launch timed by itself, a big fill in front of it
        tot = 0.0
        for _ in range(reps):
            junk.fill_(1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / reps
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


ops.mpnn_edge(A, Bm, csr, w32f[0], w32f[1], agg, partials)
ref = agg.clone()
A2, B2 = A * ops.LOG2E, Bm * ops.LOG2E
deg = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
print(f"graph: N={N} E={E} degree histogram (top): {torch.bincount(deg)[-6:].tolist()} max={int(deg.max())}")
for pmin in [int(v) for v in os.environ.get("PMIN", "20").split(",")]:
    t_plan = timeit(lambda: ops.row_tiles(csr, pmin), reps=5)
    plan = ops.row_tiles(csr, pmin)
    t_splan = timeit(lambda: ops.stream_tiles(csr), reps=5)
    sp = ops.stream_tiles(csr)
    print(f"stream plan: {sp.n_chunks} chunks x <= {sp.tpc} tiles, {int(sp.ntiles.sum())} tiles ({int(sp.ntiles.min())}-{int(sp.ntiles.max())} per chunk), zero_fill {sp.zero_fill}, build {t_splan * 1e3:.0f} us")
    print(f"pmin={pmin}: {plan.n_tiles} tiles, {plan.n_strag} stragglers, {plan.rem.E} remainder edges ({100.0 * plan.rem.E / E:.2f} %), "
          f"remainder pass {plan.rem_pass}, plan build {t_plan * 1e3:.0f} us")
    for l2e in (True, False):
        Ax, Bx = (A2, B2) if l2e else (A, Bm)
        agg.fill_(float("nan"))
        ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e)
        err = float((agg - ref).norm() / ref.norm())
        nan = int(torch.isnan(agg).sum())
        res = {"edge16": [], "rows": [], "rows_tiles_only": [], "rowsr": [], "rowsr_tiles_only": [], "rowsg": [], "rowsp": [], "stream": [], "stream_deep": [], "stream_deep2": [], "stream_prio": []}
        splan = ops.stream_tiles(csr)
        agg.fill_(float("nan"))
        ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=0)
        err_s, nan_s = float((agg - ref).norm() / ref.norm()), int(torch.isnan(agg).sum())
        keep = agg.clone()
        agg.fill_(float("nan"))
        ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=1)
        same1 = bool(torch.equal(agg, keep))
        agg.fill_(float("nan"))
        ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=2)
        print(f"  stream deep 1 / 2 == stream bitwise: {same1} / {bool(torch.equal(agg, keep))}")
        agg.fill_(float("nan"))
        ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e, wreg=3)
        err_p, nan_p = float((agg - ref).norm() / ref.norm()), int(torch.isnan(agg).sum())
        agg.fill_(float("nan"))
        ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e, wreg=2)
        err_g, nan_g = float((agg - ref).norm() / ref.norm()), int(torch.isnan(agg).sum())
        agg.fill_(float("nan"))
        ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e, wreg=True)
        err_r, nan_r = float((agg - ref).norm() / ref.norm()), int(torch.isnan(agg).sum())

        def tiles_only_r():
            ops.L.check(ops.L.lib().magnet_mpnn_edge_rowsr_fp16x3_f32(
                ops._p(Ax), ops._p(Bx), ops._p(csr.src), ops._p(plan.tiles), plan.n_tiles, None, None, None, ops._p(w32t[0]), ops._p(w16[1]),
                ops._p(agg), int(l2e), ops._stream()), "rowsr")

        def tiles_only():
            ops.L.check(ops.L.lib().magnet_mpnn_edge_rows_fp16x3_f32(
                ops._p(Ax), ops._p(Bx), ops._p(csr.src), ops._p(plan.tiles), plan.n_tiles, None, None, None, ops._p(w32t[0]), ops._p(w16[1]),
                ops._p(agg), int(l2e), ops._stream()), "rows")

        for _ in range(R):
            res["edge16"].append(timeit(lambda: ops.mpnn_edge16(Ax, Bx, csr, w16[0], w16[1], agg, partials, l2e=l2e)))
            res["rows"].append(timeit(lambda: ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e)))
            res["rows_tiles_only"].append(timeit(tiles_only))
            res["rowsr"].append(timeit(lambda: ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e, wreg=True)))
            res["rowsr_tiles_only"].append(timeit(tiles_only_r))
            res["rowsg"].append(timeit(lambda: ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e, wreg=2)))
            res["rowsp"].append(timeit(lambda: ops.mpnn_edge_rows(Ax, Bx, csr, plan, w16[0], w32t[0], w16[1], agg, partials, l2e=l2e, wreg=3)))
            res["stream"].append(timeit(lambda: ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=0)))
            res["stream_deep"].append(timeit(lambda: ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=1)))
            res["stream_deep2"].append(timeit(lambda: ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=2)))
            res["stream_prio"].append(timeit(lambda: ops.mpnn_edge_stream(Ax, Bx, csr, splan, w32t[0], w16[1], agg, l2e=l2e, deep=5)))
        fl = 2 * E * H * H * 3
        line = "  ".join(f"{k} " + "/".join(f"{v:.4f}" for v in vs) for k, vs in res.items())
        best = min(res["rows"])
        print(f"  l2e={l2e}: rel-diff vs fp32 kernel {err:.2e} nan={nan} (rowsr {err_r:.2e} nan={nan_r}, rowsg {err_g:.2e} nan={nan_g}, rowsp {err_p:.2e} nan={nan_p}, stream {err_s:.2e} nan={nan_s}) | ms per launch: {line} | frac of 2.5 PF: "
              f"stream {fl / min(res['stream']) / 1e9 / 2500:.3f}, rows {fl / best / 1e9 / 2500:.3f}, rowsr {fl / min(res['rowsr']) / 1e9 / 2500:.3f}, rowsg {fl / min(res['rowsg']) / 1e9 / 2500:.3f}, rowsp {fl / min(res['rowsp']) / 1e9 / 2500:.3f}, edge16 {fl / min(res['edge16']) / 1e9 / 2500:.3f}")

if os.environ.get("MAGNET_DEV_LIB", "0") == "1":          # ablations of the main kernel (development library; results invalid)
    import ctypes as C
    lib = ops.L.lib()
    plan = ops.row_tiles(csr, 20)
    knob = C.c_int.in_dll(lib, "g_edge_rows_ablation")
    names = {0: "as shipped", 1: "no MFMA", 2: "no epilogue Swish", 4: "no input Swish / pieces", 6: "no Swish at all",
             16: "no LDS weight reads", 23: "skeleton + gathers", 32: "S tiles: no stores", 64: "S tiles: no target loads", 96: "S tiles: neither"}
    for with_s in (False, True):
        for a, name in names.items():
            knob.value = a
            ms = min(timeit(lambda: ops.L.check(lib.magnet_mpnn_edge_rows_fp16x3_f32(
                ops._p(A2), ops._p(B2), ops._p(csr.src), ops._p(plan.tiles), plan.n_tiles, ops._p(plan.strag_ptr) if with_s else None,
                ops._p(plan.strag_row), ops._p(plan.strag_src), ops._p(w32t[0]), ops._p(w16[1]), ops._p(agg), 1, ops._stream()), "rows"))
                for _ in range(3))
            print(f"  k_mpnn_edge_rows32 {'with' if with_s else 'without'} stragglers, ablation {a:2d} ({name}): {ms:.4f} ms")
    knob.value = 0
