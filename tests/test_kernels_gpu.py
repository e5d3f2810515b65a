"""GPU unit parity of the individual C-ABI ops against the oracle's restatements."""
import pytest
import torch

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


def _torch_mlp(m, x):
    """host fp64 evaluation of a magnet_amd MLP parameter container"""
    m = m.double().cpu()
    for layer in m.layers:
        x = layer(x)
    return x


def _random_csr(n, max_deg, seed, empty_frac=0.2, long_rows=2):
    g = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, max_deg + 1, (n,), generator=g)
    deg[torch.rand(n, generator=g) < empty_frac] = 0
    for r in torch.randint(0, n, (long_rows,), generator=g):
        deg[r] = 200 + int(torch.randint(0, 100, (1,), generator=g))
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(deg, 0)
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    return rowptr.int(), tgt


@pytest.mark.parametrize("h", [32, 64, 128])
def test_segment_mean_matches_scatter_mean(h):
    from magnet_amd import ops
    from oracle.pyg_semantics import scatter_mean
    n = 777
    rowptr, tgt = _random_csr(n, 40, seed=h)
    msg = torch.randn(tgt.numel(), h, generator=torch.Generator().manual_seed(1))
    out = ops.segment_mean(msg.cuda(), rowptr.cuda(), n)
    want = scatter_mean(msg, tgt, n)
    assert rel_l2(out, want) < 1e-6
    assert float(out.cpu()[(rowptr[1:] == rowptr[:-1])].abs().max()) == 0.0      # empty rows -> 0


def test_segment_mean_empty_and_single():
    from magnet_amd import ops
    rowptr = torch.tensor([0, 0, 1, 1], dtype=torch.int32).cuda()
    msg = torch.arange(128, dtype=torch.float32)[None].cuda()
    out = ops.segment_mean(msg, rowptr, 3).cpu()
    assert torch.equal(out[1], msg[0].cpu()) and out[0].abs().sum() == 0 and out[2].abs().sum() == 0


@pytest.mark.parametrize("dim,loop", [(1, False), (2, False), (2, True)])
def test_radius_graph_matches_torch_cluster_restatement(dim, loop):
    from magnet_amd import ops
    from oracle.pyg_semantics import radius_graph
    g = torch.Generator().manual_seed(dim * 7 + loop)
    B, N = 3, 150
    x = torch.rand(B * N, dim, generator=g)
    x[:40] *= 0.1                                        # dense cluster: the 32-neighbour cap is hit
    r = 0.25
    batch = torch.arange(B).repeat_interleave(N)
    want = radius_graph(x, r, batch, loop=loop)
    for brute in (False, True):                            # bounding-box pruned and brute-force kernels
        found, query, qptr, E = ops.radius_pairs(x.cuda(), r, ops.graph_ptr(B, N, "cuda"), loop=loop, brute_force=brute)
        got = torch.stack([found.long(), query.long()]).cpu()
        assert E == want.shape[1]
        assert torch.equal(got, want)
    deg = torch.bincount(want[1], minlength=B * N)
    assert int(deg.max()) >= 32                            # the cap was exercised


def test_csr_by_target_is_stable_transpose():
    from magnet_amd import ops
    g = torch.Generator().manual_seed(3)
    n, E = 300, 5000
    src = torch.randint(0, n, (E,), generator=g).int()
    tgt = torch.randint(0, n // 3, (E,), generator=g).int()       # leaves empty rows
    csr = ops.csr_by_target(src.cuda(), tgt.cuda(), n)
    order = torch.sort(tgt.long(), stable=True).indices
    assert torch.equal(csr.perm.cpu().long(), order)
    assert torch.equal(csr.src.cpu(), src[order]) and torch.equal(csr.tgt.cpu(), tgt[order])
    assert torch.equal(csr.rowptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.bincount(tgt.long(), minlength=n), 0)]))


def test_csr_by_target_long_rows_take_the_radix_path():
    """rows beyond 1024 entries (the inverse CSR of an INR gather index has rows of 10^4..10^5) are sorted by the per-row
    binary radix kernel instead of the O(d^2) rank sort: same stable transpose, including a row that owns half the edges,
    one exactly at the threshold and an edge count that is not a power of two"""
    from magnet_amd import ops
    g = torch.Generator().manual_seed(5)
    n, E = 60, 70001
    tgt = torch.randint(0, n, (E,), generator=g)
    tgt[torch.rand(E, generator=g) < 0.5] = 7                      # one row owns ~half of the edges
    tgt[:1024] = 11                                                # ... and row 11 gets at least the threshold
    tgt = tgt[torch.randperm(E, generator=g)].int()
    src = torch.randint(0, n, (E,), generator=g).int()
    csr = ops.csr_by_target(src.cuda(), tgt.cuda(), n)
    order = torch.sort(tgt.long(), stable=True).indices
    assert int(torch.bincount(tgt.long()).max()) > 30000
    assert torch.equal(csr.perm.cpu().long(), order)
    assert torch.equal(csr.src.cpu(), src[order]) and torch.equal(csr.tgt.cpu(), tgt[order])


def test_knn_matches_restatement():
    from magnet_amd import ops
    from oracle.pyg_semantics import knn
    g = torch.Generator().manual_seed(5)
    B, L, Nq = 2, 37, 91
    x, y = torch.rand(B * L, 2, generator=g), torch.rand(B * Nq, 2, generator=g)
    y[:5] = x[:5]                                         # exact hits / ties with distance 0
    bx, by = torch.arange(B).repeat_interleave(L), torch.arange(B).repeat_interleave(Nq)
    want = knn(x, y, 4, bx, by)[1].view(B * Nq, 4)
    got = ops.knn(x.cuda(), ops.graph_ptr(B, L, "cuda"), y.cuda(), ops.graph_ptr(B, Nq, "cuda"), 4)
    assert torch.equal(got.cpu().long(), want)


@pytest.mark.parametrize("L,Nq,dim,k", [(1024, 256, 2, 4), (3, 50, 2, 4), (200, 33, 3, 8), (64, 64, 1, 2), (130, 7, 2, 1)])
def test_knn_wave_kernel_on_lattices_with_ties_and_short_graphs(L, Nq, dim, k):
    """one wave per query (k_knn_wave): candidates are spread over the lanes and merged - ties (lattice points at equal distance, in
    different lanes), fewer than k candidates (-1 fill), k up to 8, 1-D / 3-D, the reference's training shape 1024 x 256"""
    from magnet_amd import ops
    from oracle.pyg_semantics import knn
    g = torch.Generator().manual_seed(L + Nq)
    B = 3
    side = max(2, int(round(L ** (1.0 / dim))))
    lat = torch.stack(torch.meshgrid(*([torch.arange(side, dtype=torch.float32)] * dim), indexing="ij"), -1).reshape(-1, dim)
    x = torch.cat([lat[torch.randperm(lat.shape[0], generator=g)][:L] if lat.shape[0] >= L else torch.rand(L, dim, generator=g) * side
                   for _ in range(B)])
    y = torch.cat([(torch.randint(0, 2 * side, (Nq, dim), generator=g).float() * 0.5) for _ in range(B)])    # cell centres and corners: exact ties
    bx, by = torch.arange(B).repeat_interleave(L), torch.arange(B).repeat_interleave(Nq)
    want = knn(x, y, k, bx, by)
    got = ops.knn(x.cuda(), ops.graph_ptr(B, L, "cuda"), y.cuda(), ops.graph_ptr(B, Nq, "cuda"), k).cpu().long()
    if L >= k:
        assert torch.equal(got, want[1].view(B * Nq, k))
    else:                                                 # the restatement drops missing neighbours, the kernel pads with -1
        assert torch.equal(got[:, :L], want[1].view(B * Nq, L)) and bool((got[:, L:] == -1).all())


def test_instance_norm_matches_pyg_restatement():
    from magnet_amd import ops
    from oracle.pyg_semantics import instance_norm
    g = torch.Generator().manual_seed(9)
    B, N, H = 3, 333, 128
    x = torch.randn(B * N, H, generator=g) * 3 + 5
    want = instance_norm(x, torch.arange(B).repeat_interleave(N))
    got = ops.instance_norm(x.cuda(), B, N)
    assert rel_l2(got, want) < 1e-6


def _two_accumulator_form(monkeypatch, mode, hidden):
    """"fp16x3-2acc": the 64-wide 5-Linear chains in the two-accumulator form of fp16x3 (their default is the one-accumulator form
    "fp16x3s", packing.chain_gemm)"""
    if mode != "fp16x3-2acc":
        return mode
    if hidden != 64:
        pytest.skip("the 128-wide chains have one form")
    monkeypatch.setenv("MAGNET_CHAIN_GEMM", "fp16x3")
    return "fp16x3"


@pytest.mark.parametrize("mode", ["fp16x3", "fp16x3-2acc", "bf16x6", "fp32"])
@pytest.mark.parametrize("hidden,out,k_in,ln", [(128, 128, 45, True), (64, 32, 96, True), (128, 7, 128, False), (64, 64, 40, True)])
def test_generic_mlp_matches_torch(hidden, out, k_in, ln, mode, monkeypatch):
    """5 Linears + ReLU (+LayerNorm) through magnet_mlp_f32 vs torch fp64 on the host."""
    from magnet_amd import ops
    mode = _two_accumulator_form(monkeypatch, mode, hidden)
    from magnet_amd.models.backbones.mlp import MLP
    from magnet_amd.models.magnet_common import pack_mlp, pack_mlp_ln, run_mlp
    from magnet_amd.packing import pad_to
    torch.manual_seed(0)
    m = MLP(k_in, [hidden] * 4, out)
    norm = torch.nn.LayerNorm(out)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5), norm.bias.uniform_(-0.5, 0.5)
    M = 1000 + 13
    kp = pad_to(k_in)
    x = torch.zeros(M, kp)
    x[:, :k_in] = torch.randn(M, k_in)
    seq = torch.nn.Sequential(m, norm).cuda()
    pk = pack_mlp_ln(seq, k_pad=kp, mode=mode) if ln else pack_mlp(seq[0], k_pad=kp, mode=mode)
    y = torch.full((M, out), float("nan"), device="cuda")
    run_mlp(M, ops.ktiles(ops.Term(x.cuda(), kp)), pk, y)
    with torch.no_grad():
        ref = _torch_mlp(m, x[:, :k_in].double())
        if ln:
            ref = norm.double().cpu()(ref)
    assert rel_l2(y, ref) < 2e-6


@pytest.mark.parametrize("mode", ["fp16x3", "fp16x3-2acc", "bf16x6", "fp32"])
def test_mlp_gather_difference_and_segmean_epilogue(mode, monkeypatch):
    """the edge path: gathered / differenced sources -> MLP -> LayerNorm -> fused CSR mean == unfused."""
    from magnet_amd import ops
    mode = _two_accumulator_form(monkeypatch, mode, 64)
    from magnet_amd.models.backbones.mlp import MLP
    from magnet_amd.models.magnet_common import pack_mlp_ln, run_mlp
    from oracle.pyg_semantics import scatter_mean
    torch.manual_seed(1)
    n, D = 500, 32
    rowptr, tgt = _random_csr(n, 20, seed=11)
    E = tgt.numel()
    src = torch.randint(0, n, (E,))
    x = torch.randn(n, D)
    e0 = torch.randn(E, D)
    seq = torch.nn.Sequential(MLP(3 * D, [64] * 4, D), torch.nn.LayerNorm(D)).cuda()
    pk = pack_mlp_ln(seq, mode=mode)
    xc, ec = x.cuda(), e0.cuda()
    kts = ops.ktiles(ops.Term(xc, D, idx=tgt.int().cuda()), ops.Term(xc, D, idx=src.int().cuda()), ops.Term(ec, D, scale=4.0))
    agg = torch.full((n, D), float("nan"), device="cuda")
    partials = torch.empty(((E + 31) // 32) * 2 * D, device="cuda")
    run_mlp(E, kts, pk, agg, epi=ops.EPI_SEGMEAN, tgt=tgt.int().cuda(), rowptr=rowptr.cuda(), partials=partials)
    ops.segmean_fixup(rowptr.cuda(), n, D, partials, agg)
    with torch.no_grad():
        m = seq[1].double().cpu()(_torch_mlp(seq[0], torch.cat([x[tgt], x[src], 4.0 * e0], -1).double()))
    want = scatter_mean(m, tgt, n)
    assert not torch.isnan(agg).any()
    assert rel_l2(agg, want) < 2e-6


@pytest.mark.parametrize("mode", ["fp16x3", "bf16x6", "fp32"])
def test_mlp_column_groups_in_block_equal_separate_launches(mode):
    """groups_in_block / y_group_stride of magnet_mlp_args_t: two Linears sharing their input rows as the column groups of
    one launch (what MPNN does for the per-node halves of message_net_1) write exactly what two launches write - rows that
    are not a multiple of the tile, outputs a full buffer apart."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    torch.manual_seed(4)
    n, k = 1000, 160
    x = torch.randn(n, k, device="cuda")
    la, lb = torch.nn.Linear(k, 128).cuda(), torch.nn.Linear(k, 128, bias=False).cuda()
    pa, pb = pack(la.weight, la.bias, mode), pack(lb.weight, None, mode)
    kts = ops.ktiles(ops.Term(x, k))
    want = torch.empty(2, n, 128, device="cuda")
    ops.mlp(n, kts, [pa], nl=1, ht=4, ot=4, act=ops.ACT_NONE, y=want[0])
    ops.mlp(n, kts, [pb], nl=1, ht=4, ot=4, act=ops.ACT_NONE, y=want[1])
    got = torch.full((2, n, 128), float("nan"), device="cuda")
    ops.mlp(n, kts, [pa, pb], nl=1, ht=4, ot=4, act=ops.ACT_NONE, y=got, ldy=128, colgroups=2, groups_in_block=True,
            y_group_stride=n * 128)
    assert torch.equal(got, want)
    with torch.no_grad():
        ref = torch.stack([la.double()(x.double()), lb.double()(x.double())])
    assert rel_l2(got, ref) < 1e-6


def test_mpnn_edge_kernels_agree_on_ragged_graph():
    """fp32-MFMA and bf16-split edge kernels vs a host evaluation, on a graph with empty rows, rows longer
    than several tiles and an edge count that is not a multiple of 32."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    from oracle.pyg_semantics import scatter_mean
    torch.manual_seed(2)
    n, H = 301, 128
    rowptr, tgt = _random_csr(n, 45, seed=5, long_rows=3)
    E = tgt.numel()
    assert E % 32 != 0
    src = torch.randint(0, n, (E,))
    A, Bm = torch.randn(n, H), torch.randn(n, H)
    lin = torch.nn.Linear(H, H)
    csr = ops.CSR(rowptr.cuda(), src.int().cuda(), tgt.int().cuda(), n, E)
    sw = lambda x: x * torch.sigmoid(x)  # noqa: E731
    with torch.no_grad():
        m = sw(lin.double()(sw((A[tgt] + Bm[src]).double())))
    want = scatter_mean(m, tgt, n)
    lin = lin.float().cuda()
    for mode in ("fp16x3", "bf16x6", "fp32"):
        w, b = pack(lin.weight, lin.bias, mode)
        agg = torch.full((n, H), float("nan"), device="cuda")
        partials = torch.zeros(((E + 31) // 32) * 2 * H, device="cuda")
        ops.mpnn_edge(A.cuda(), Bm.cuda(), csr, w, b, agg, partials)
        assert not torch.isnan(agg).any()
        assert rel_l2(agg, want) < 1e-6, mode


def test_mpnn_edge_log2e_scaled_inputs_give_the_same_mean():
    """magnet_mpnn_edge_*_l2e_f32: A and Bm multiplied by log2(e) (what the model folds into the packed message_net_1
    halves), unscaled W2 / b2 - the aggregated messages are the same quantity as from the plain entry point."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    torch.manual_seed(11)
    n, H = 257, 128
    rowptr, tgt = _random_csr(n, 40, seed=9, long_rows=2)
    E = tgt.numel()
    src = torch.randint(0, n, (E,))
    A, Bm = torch.randn(n, H).cuda(), torch.randn(n, H).cuda()
    lin = torch.nn.Linear(H, H).cuda()
    csr = ops.CSR(rowptr.cuda(), src.int().cuda(), tgt.int().cuda(), n, E)
    for mode, tol in (("fp16x3", 1e-6), ("bf16", 2e-2)):
        w, b = pack(lin.weight, lin.bias, mode)
        ref = torch.full((n, H), float("nan"), device="cuda")
        got = torch.full((n, H), float("nan"), device="cuda")
        partials = torch.zeros(((E + 31) // 32) * 2 * H, device="cuda")
        ops.mpnn_edge(A, Bm, csr, w, b, ref, partials)
        ops.mpnn_edge(A * ops.LOG2E, Bm * ops.LOG2E, csr, w, b, got, partials, l2e=True)
        assert not torch.isnan(got).any()
        assert rel_l2(got, ref) < tol, (mode, rel_l2(got, ref))


@pytest.mark.parametrize("case", ["ragged", "no_edges", "under_one_tile", "one_row_owns_everything", "every_row_one_edge",
                                  "small_values", "mpnn_like"])
@pytest.mark.parametrize("l2e", [False, True])
@pytest.mark.parametrize("kernel", ["edge16", "edge16s", "edge32t", "edge32p", "edgews", "rows", "rowsr", "rowsg", "rowsp", "stream"])
def test_mpnn_edge16_matches_host_evaluation(case, l2e, kernel):
    """k_mpnn_edge16 (16-edge tiles, v_mfma_f32_16x16x32_f16, two waves per SIMD, its own weight fragment order) and k_mpnn_edge32t
    (generation 9: transposed 32-edge tiles, lane = output feature, one accumulator, unscaled fp16-subnormal low pieces) against a
    host evaluation on a ragged graph (empty rows, rows across many tiles, E % 16 != 0), the range-partials corner cases, a
    graph whose every tile is full of run ends, and inputs small enough that every low piece is an fp16 subnormal."""
    from magnet_amd import _lib, ops
    from magnet_amd.packing import pack_edge16_fp16x2, pack_edge32t_fp16x2
    from oracle.pyg_semantics import scatter_mean
    if kernel not in ("edge16", "edge16s", "stream"):
        if not _lib.DEV:                       # generations 9-13 live in the development library only (include/magnet_hip_dev.h)
            pytest.skip("development-library kernel: MAGNET_DEV_LIB=1 after `python -m magnet_amd.build --dev`")
        from magnet_amd import dev_ops
    torch.manual_seed(3)
    H = 128
    if case in ("ragged", "small_values"):
        n = 301
        rowptr, tgt = _random_csr(n, 45, seed=5, long_rows=3)
    elif case == "mpnn_like":                  # the degrees of an MPNN_2d radius graph: 33 inside, 14 .. 32 on the frame, a few empty rows
        g = torch.Generator().manual_seed(17)
        n = 2500
        deg = torch.where(torch.rand(n, generator=g) < 0.7, torch.full((n,), 33), torch.randint(14, 33, (n,), generator=g))
        deg[torch.randint(0, n, (9,), generator=g)] = 0
        rowptr = torch.zeros(n + 1, dtype=torch.int32)
        rowptr[1:] = deg.cumsum(0)
        tgt = torch.repeat_interleave(torch.arange(n), deg)
    else:
        n, deg = {"no_edges": (5, [0] * 5), "under_one_tile": (6, [3, 0, 7, 1, 0, 2]),
                  "one_row_owns_everything": (3, [0, 40001, 0]), "every_row_one_edge": (3001, [1, 2] * 1500 + [1])}[case]
        rowptr = torch.zeros(n + 1, dtype=torch.int32)
        rowptr[1:] = torch.tensor(deg).cumsum(0)
        tgt = torch.repeat_interleave(torch.arange(n), torch.tensor(deg))
    E = tgt.numel()
    src = torch.randint(0, n, (E,))
    A, Bm = torch.randn(n, H), torch.randn(n, H)
    if case == "small_values":                 # Swish(A + Bm) ~ 1e-3: the low pieces of the activations are fp16 subnormals
        A, Bm = A * 2e-3, Bm * 2e-3
    lin = torch.nn.Linear(H, H)
    sw = lambda x: x * torch.sigmoid(x)  # noqa: E731
    with torch.no_grad():
        m = sw(lin.double()(sw((A[tgt.long()] + Bm[src.long()]).double())))
    want = scatter_mean(m, tgt.long(), n) if E else torch.zeros(n, H, dtype=torch.float64)
    csr = ops.CSR(rowptr.int().cuda(), src.int().cuda(), tgt.int().cuda(), n, E)
    agg = torch.full((n, H), float("nan"), device="cuda")
    partials = torch.zeros(ops.edge_scratch_floats(E), device="cuda")
    scale = ops.LOG2E if l2e else 1.0
    if kernel in ("edge16", "edge16s"):       # edge16s: the one-accumulator variant (planes pre-scaled by 2^11)
        w, b = pack_edge16_fp16x2(lin.weight.cuda(), lin.bias.cuda(), scaled=kernel == "edge16s")
        ops.mpnn_edge16(A.cuda() * scale, Bm.cuda() * scale, csr, w, b, agg, partials, l2e=l2e, single=kernel == "edge16s")
    elif kernel in ("rows", "rowsr", "rowsg", "rowsp"):         # generation 11: tiles aligned to CSR rows + the remainder edges through k_mpnn_edge16
        # (rowsr: generation 12, the same tiles with W2's planes resident in the accumulation registers; rowsg: 12b, + grouped stragglers)
        w16, b = pack_edge16_fp16x2(lin.weight.cuda(), lin.bias.cuda())
        w32, _ = pack_edge32t_fp16x2(lin.weight.cuda(), lin.bias.cuda())
        plan = dev_ops.row_tiles(csr)
        deg = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
        has = deg >= plan.pmin
        left = deg - torch.where(has, deg.clamp(max=32), 0)
        strag = has & (left == 1)
        assert plan.n_tiles == int(has.sum()) and plan.n_strag == int(strag.sum()) and plan.rem.E == int(left[~strag].sum())
        dev_ops.mpnn_edge_rows(A.cuda() * scale, Bm.cuda() * scale, csr, plan, w16, w32, b, agg, partials, l2e=l2e, wreg={"rows": 0, "rowsr": 1, "rowsg": 2, "rowsp": 3}[kernel])
    elif kernel == "stream":                  # generation 14: 32-edge tiles cut from the edge stream of row-aligned chunks (two rows per tile)
        w16, b = pack_edge16_fp16x2(lin.weight.cuda(), lin.bias.cuda())
        w32, _ = pack_edge32t_fp16x2(lin.weight.cuda(), lin.bias.cuda())
        plan = ops.stream_tiles(csr)
        # (no plan: no edges, one row owning 1 250 tiles or a few rows far longer than a chunk's share, rows of one or two edges - a tile
        #  never touches a third row - : the model then runs k_mpnn_edge16, and so does this test)
        assert plan is not None or case != "mpnn_like"
        assert plan is None or case not in ("no_edges", "one_row_owns_everything", "every_row_one_edge")
        if plan is None:
            ops.mpnn_edge16(A.cuda() * scale, Bm.cuda() * scale, csr, w16, b, agg, partials, l2e=l2e)
        else:
            assert plan.zero_fill == bool(((csr.rowptr[1:] - csr.rowptr[:-1]) == 0).any())
            nt = plan.ntiles.cpu()
            assert int(nt.sum()) == plan.n_tiles and int(nt.max()) <= plan.tpc
            ops.mpnn_edge_stream(A.cuda() * scale, Bm.cuda() * scale, csr, plan, w32, b, agg, l2e=l2e)
    elif kernel == "edgews":                  # generation 10: wave-specialised (gather / GEMM / epilogue waves through LDS)
        w, b = pack_edge32t_fp16x2(lin.weight.cuda(), lin.bias.cuda())
        dev_ops.mpnn_edge_ws(A.cuda() * scale, Bm.cuda() * scale, csr, w, b, agg, partials, l2e=l2e)
    else:
        w, b = pack_edge32t_fp16x2(lin.weight.cuda(), lin.bias.cuda())
        dev_ops.mpnn_edge32t(A.cuda() * scale, Bm.cuda() * scale, csr, w, b, agg, partials, l2e=l2e, paired=kernel == "edge32p")
    assert not torch.isnan(agg).any()
    if E:
        # a 40 001-term fp32 running sum (one row owning every edge) carries ~1e-6 of rounding by itself
        assert rel_l2(agg, want) < (3e-6 if case == "one_row_owns_everything" else 1e-6), rel_l2(agg, want)
    else:
        assert float(agg.abs().max()) == 0.0


@pytest.mark.parametrize("pattern", ["all33", "all32", "30to33", "all64", "all65", "17to40", "mesh_with_holes", "long_and_short", "tiny"])
@pytest.mark.parametrize("l2e", [False, True])
def test_mpnn_edge_stream_degree_patterns(pattern, l2e):
    """k_mpnn_edge_stream32 (generation 14): 32-edge tiles cut from the edge stream, two rows per tile, the open row's sum carried from tile
    to tile.  Degree patterns that move the split through every position (33), never split (32, 64), split rows over three tiles (65),
    force cuts in front of a third row (17 .. 40), leave rows empty, and a graph smaller than one workgroup's chunks - against a host
    evaluation in fp64; the plan itself is checked edge by edge (every edge in exactly one tile, at most two rows per tile, chunks
    aligned to rows)."""
    from magnet_amd import ops
    from magnet_amd.packing import pack_edge16_fp16x2, pack_edge32t_fp16x2
    from oracle.pyg_semantics import scatter_mean
    g = torch.Generator().manual_seed(23)
    H = 128
    n = {"tiny": 7, "long_and_short": 900}.get(pattern, 3000)
    if pattern == "all33":
        deg = torch.full((n,), 33)
    elif pattern == "all32":
        deg = torch.full((n,), 32)
    elif pattern == "30to33":
        deg = torch.randint(30, 34, (n,), generator=g)
    elif pattern == "all64":
        deg = torch.full((n,), 64)
    elif pattern == "all65":
        deg = torch.full((n,), 65)
    elif pattern == "17to40":
        deg = torch.randint(17, 41, (n,), generator=g)
    elif pattern == "mesh_with_holes":
        deg = torch.where(torch.rand(n, generator=g) < 0.9, torch.full((n,), 33), torch.randint(20, 33, (n,), generator=g))
        deg[torch.randint(0, n, (25,), generator=g)] = 0
        deg[0] = 0
        deg[-1] = 0
    elif pattern == "long_and_short":
        deg = torch.where(torch.rand(n, generator=g) < 0.5, torch.randint(90, 130, (n,), generator=g), torch.randint(8, 20, (n,), generator=g))
    else:
        deg = torch.tensor([33, 5, 0, 31, 33, 2, 40])
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = deg.cumsum(0)
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    E = tgt.numel()
    src = torch.randint(0, n, (E,), generator=g)
    A, Bm = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    lin = torch.nn.Linear(H, H)
    sw = lambda x: x * torch.sigmoid(x)  # noqa: E731
    with torch.no_grad():
        m = sw(lin.double()(sw((A[tgt.long()] + Bm[src.long()]).double())))
    want = scatter_mean(m, tgt.long(), n)
    csr = ops.CSR(rowptr.int().cuda(), src.int().cuda(), tgt.int().cuda(), n, E)
    plan = ops.stream_tiles(csr)
    assert plan is not None
    # ---- the plan, edge by edge
    tiles, nt = plan.tiles.cpu(), plan.ntiles.cpu()
    covered = torch.zeros(E, dtype=torch.int32)
    rp = rowptr.long()
    next_row_start = 0
    for c in range(plan.n_chunks):
        for k in range(int(nt[c])):
            e0, r, z, w = (int(v) for v in tiles[c, k])
            cnt, split, ends0, ends1 = z & 0xFF, (z >> 8) & 0xFF, bool(z & (1 << 16)), bool(z & (1 << 17))
            assert 1 <= cnt <= 32 and 1 <= split <= cnt
            covered[e0:e0 + cnt] += 1
            assert rp[r] <= e0 < rp[r + 1] and split == min(int(rp[r + 1]) - e0, cnt)
            assert (w & 0xFFFF) == int(deg[r])
            assert ends0 == (int(rp[r + 1]) <= e0 + cnt)
            if split < cnt:                              # a second row: it is row r + 1 and nothing beyond it is touched
                assert int(deg[r + 1]) > 0 and e0 + cnt <= int(rp[r + 2]) and (w >> 16) == int(deg[r + 1])
                assert ends1 == (int(rp[r + 2]) == e0 + cnt)
            else:
                assert not ends1
            if k == 0:                                   # a chunk begins where a row begins ...
                assert e0 == int(rp[r]) and e0 == next_row_start
            if k == int(nt[c]) - 1:                      # ... and ends where a row ends
                assert ends1 if split < cnt else ends0
                next_row_start = e0 + cnt
    assert bool((covered == 1).all()) and next_row_start == E
    assert plan.zero_fill == bool((deg == 0).any())
    # ---- the kernel
    w32, _ = pack_edge32t_fp16x2(lin.weight.cuda(), lin.bias.cuda())
    _, b = pack_edge16_fp16x2(lin.weight.cuda(), lin.bias.cuda())
    scale = ops.LOG2E if l2e else 1.0
    agg = torch.full((n, H), float("nan"), device="cuda")
    ops.mpnn_edge_stream(A.cuda() * scale, Bm.cuda() * scale, csr, plan, w32, b, agg, l2e=l2e)
    assert not torch.isnan(agg).any()
    assert rel_l2(agg, want) < 1e-6, rel_l2(agg, want)
    from magnet_amd import _lib
    # repeatable; two / three gather sets in flight (development library: four as well): the same bits
    for deep in (None, 0, 1) + ((2,) if _lib.DEV else ()):
        again = torch.full((n, H), float("nan"), device="cuda")
        ops.mpnn_edge_stream(A.cuda() * scale, Bm.cuda() * scale, csr, plan, w32, b, again, l2e=l2e, deep=deep)
        assert torch.equal(agg, again), deep


@pytest.mark.parametrize("seed", range(12))
def test_mpnn_edge_stream_random_degree_sequences(seed):
    """k_mpnn_edge_stream32 on seeded random degree sequences (a mixture per seed of a constant, a uniform range, a two-point
    distribution and empty rows, 200 .. 4 000 rows): whenever the plan exists, the kernel agrees with a host evaluation in fp64 and
    with k_mpnn_edge16 on the same inputs; whenever it does not, the reason is one of the documented ones."""
    from magnet_amd import ops
    from magnet_amd.packing import pack_edge16_fp16x2, pack_edge32t_fp16x2
    from oracle.pyg_semantics import scatter_mean
    g = torch.Generator().manual_seed(1000 + seed)
    H = 128
    n = int(torch.randint(200, 4000, (1,), generator=g))
    kind = seed % 4
    if kind == 0:
        deg = torch.full((n,), int(torch.randint(17, 70, (1,), generator=g)))
    elif kind == 1:
        lo = int(torch.randint(17, 40, (1,), generator=g))
        deg = torch.randint(lo, lo + int(torch.randint(1, 30, (1,), generator=g)), (n,), generator=g)
    elif kind == 2:
        a, b = int(torch.randint(18, 34, (1,), generator=g)), int(torch.randint(34, 100, (1,), generator=g))
        deg = torch.where(torch.rand(n, generator=g) < 0.5, torch.full((n,), a), torch.full((n,), b))
    else:
        deg = torch.randint(20, 36, (n,), generator=g)
        deg[torch.rand(n, generator=g) < 0.05] = 0
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = deg.cumsum(0)
    tgt = torch.repeat_interleave(torch.arange(n), deg)
    E = tgt.numel()
    src = torch.randint(0, n, (E,), generator=g)
    A, Bm = torch.randn(n, H, generator=g), torch.randn(n, H, generator=g)
    lin = torch.nn.Linear(H, H)
    sw = lambda x: x * torch.sigmoid(x)  # noqa: E731
    with torch.no_grad():
        m = sw(lin.double()(sw((A[tgt.long()] + Bm[src.long()]).double())))
    want = scatter_mean(m, tgt.long(), n)
    csr = ops.CSR(rowptr.int().cuda(), src.int().cuda(), tgt.int().cuda(), n, E)
    plan = ops.stream_tiles(csr)
    assert plan is not None, (seed, n, E)              # rows of 17 edges and more, none longer than a chunk's slack: always planned
    nt = plan.ntiles.cpu()
    assert int(nt.sum()) == plan.n_tiles and 32 * plan.n_tiles >= E and plan.zero_fill == bool((deg == 0).any())
    w32, _ = pack_edge32t_fp16x2(lin.weight.cuda(), lin.bias.cuda())
    w16, b = pack_edge16_fp16x2(lin.weight.cuda(), lin.bias.cuda())
    agg = torch.full((n, H), float("nan"), device="cuda")
    ops.mpnn_edge_stream(A.cuda(), Bm.cuda(), csr, plan, w32, b, agg)
    assert not torch.isnan(agg).any() and rel_l2(agg, want) < 1e-6, rel_l2(agg, want)
    ref = torch.full((n, H), float("nan"), device="cuda")
    ops.mpnn_edge16(A.cuda(), Bm.cuda(), csr, w16, b, ref, torch.zeros(ops.edge_scratch_floats(E), device="cuda"))
    assert rel_l2(agg, ref) < 5e-7


@pytest.mark.parametrize("case", ["no_edges", "under_one_tile", "one_row_owns_everything", "empty_rows_at_both_ends"])
def test_mpnn_edge_range_partials_corner_cases(case):
    """k_mpnn_edge7 carries a row's running sum across the tiles of a wave and hands only range-crossing rows to
    k_edge_range_fixup: a graph without edges, fewer edges than one tile, one row spanning every wave's range, and
    empty rows before / after the populated ones (all three arithmetic modes against a host evaluation)."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    from oracle.pyg_semantics import scatter_mean
    torch.manual_seed(7)
    H = 128
    if case == "no_edges":
        n, deg = 5, [0, 0, 0, 0, 0]
    elif case == "under_one_tile":
        n, deg = 6, [3, 0, 7, 1, 0, 2]
    elif case == "one_row_owns_everything":
        n, deg = 3, [0, 40000, 0]                      # 1250 tiles > the 1024 waves of a full grid: every range is crossed
    else:
        n, deg = 400, [0] * 30 + [int(d) for d in torch.randint(1, 70, (340,))] + [0] * 30
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.tensor(deg).cumsum(0)
    E = int(rowptr[-1])
    tgt = torch.repeat_interleave(torch.arange(n), torch.tensor(deg)).int()
    src = torch.randint(0, n, (E,)).int()
    A, Bm = torch.randn(n, H), torch.randn(n, H)
    lin = torch.nn.Linear(H, H)
    sw = lambda x: x * torch.sigmoid(x)  # noqa: E731
    with torch.no_grad():
        m = sw(lin.double()(sw((A[tgt.long()] + Bm[src.long()]).double())))
    want = scatter_mean(m, tgt.long(), n) if E else torch.zeros(n, H, dtype=torch.float64)
    csr = ops.CSR(rowptr.cuda(), src.cuda(), tgt.cuda(), n, E)
    lin = lin.float().cuda()
    for mode in ("fp16x3", "bf16x6", "fp32"):
        w, b = pack(lin.weight, lin.bias, mode)
        agg = torch.full((n, H), float("nan"), device="cuda")
        partials = torch.zeros(max((E + 31) // 32, 1) * 2 * H, device="cuda")
        ops.mpnn_edge(A.cuda(), Bm.cuda(), csr, w, b, agg, partials)
        assert not torch.isnan(agg).any(), mode
        if E:
            assert rel_l2(agg, want) < 1e-6, mode
        else:
            assert float(agg.abs().max()) == 0.0, mode


@pytest.mark.parametrize("dim,loop", [(1, False), (2, False), (2, True)])
def test_radius_search_ragged_graphs_duplicates_and_isolated_nodes(dim, loop):
    """graphs of different sizes (one of them empty, one a single node), duplicate coordinates, isolated nodes and a
    cluster beyond the 32-neighbour cap - against the oracle's torch_cluster restatement."""
    from magnet_amd import ops
    from oracle.pyg_semantics import radius_graph
    sizes = [70, 0, 1, 130, 65, 3]
    g = torch.Generator().manual_seed(17 + dim + loop)
    xs, batch = [], []
    for b, n in enumerate(sizes):
        x = torch.rand(n, dim, generator=g)
        if n >= 60:
            x[:45] = x[:45] * 0.05 + 0.3
            x[50], x[52] = x[48].clone(), x[48].clone()         # duplicates
            x[-1] = 7.0 + b                                      # isolated
        xs.append(x)
        batch.append(torch.full((n,), b, dtype=torch.long))
    x, batch = torch.cat(xs), torch.cat(batch)
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device="cuda")
    # the oracle wants consecutive graph ids: relabel (empty graph 1 disappears)
    want = radius_graph(x, 0.2, torch.unique_consecutive(batch, return_inverse=True)[1], loop=loop)
    for brute in (False, True):
        found, query, qptr, E = ops.radius_pairs(x.cuda(), 0.2, gptr, loop=loop, brute_force=brute, ragged=True)
        assert E == want.shape[1]
        assert torch.equal(torch.stack([found.long(), query.long()]).cpu(), want)
    assert int(torch.bincount(want[1]).max()) == (32 if loop else 33)


@pytest.mark.parametrize("mode,n_chan,k,res_layers,shape", [("2d", 128, 3, 3, (2, 10, 13, 9)), ("1d", 128, 3, 2, (3, 25, 31)),
                                                             ("1d", 64, 5, 1, (1, 7, 40)), ("2d", 64, 3, 1, (1, 4, 8, 8)),
                                                             ("2d", 128, 3, 1, (4, 10, 64, 64))])   # 16384 rows: unsplit launch
def test_edsr_gathered_tap_convolutions_match_torch(mode, n_chan, k, res_layers, shape):
    """the HIP EDSR (convolutions as gathered-tap GEMMs, zero row for the padding) vs the oracle's torch convolutions"""
    from magnet_amd.models.backbones.edsr import EDSR
    from oracle.models import EDSR as OracleEDSR
    torch.manual_seed(8)
    ref = OracleEDSR(shape[1], n_chan, res_layers, k, mode)
    net = EDSR(shape[1], n_chan, res_layers, k, 1, mode)
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    x = torch.randn(*shape)
    with torch.no_grad():
        want = ref(x)
    got = net(x.cuda())
    assert got.shape == want.shape
    assert rel_l2(got, want) < 2e-6
    rows = net.rows(x.cuda())                                    # channels-last rows, what the models consume
    assert rel_l2(rows, want.reshape(shape[0], n_chan, -1).permute(0, 2, 1).reshape(-1, n_chan)) < 2e-6


@pytest.mark.parametrize("d,shape,out", [(2, (6, 40, 40), (16, 16)), (2, (3, 256, 256), (64, 64)), (2, (2, 17, 23), (30, 9)),
                                         (1, (5, 40), (16,)), (1, (4, 33), (70,))])
def test_resize_linear_matches_interpolate(d, shape, out):
    from magnet_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(9)
    x = torch.randn(*shape, device="cuda")
    want = F.interpolate(x[:, None], size=out if d == 2 else out[0], mode="bilinear" if d == 2 else "linear",
                         align_corners=False)[:, 0]
    got = ops.resize_linear(x, out, d)
    assert got.shape == want.shape
    assert rel_l2(got, want) < 1e-6


@pytest.mark.parametrize("mode", ["fp16x3", "fp16x3s"])
@pytest.mark.parametrize("kin,n_graphs,rows,with_norm", [(32, 4, 4096, True), (64, 2, 2144, True), (32, 1, 1013, False),
                                                          (32, 8, 32, True), (32, 4, 16384, True)])
def test_node_halves_resident_kernel_matches_norm_plus_generic_mlp(kin, n_graphs, rows, with_norm, mode):
    """magnet_mpnn_node_halves_f32 (InstanceNorm apply on load + both per-node halves + normalised h, weights resident in LDS;
    mode "fp16x3s": magnet_mpnn_node_halves2_f32, the one-accumulator packs on two waves per SIMD)
    == magnet_instance_norm_f32 followed by the generic one-Linear launch, and both against fp64 on the host."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    g = torch.Generator().manual_seed(3)
    H, n = 128, n_graphs * rows
    wa, wb = torch.randn(H, H + kin, generator=g) * 0.1, torch.randn(H, H + kin, generator=g) * 0.1
    ba = torch.randn(H, generator=g)
    h_raw = (torch.randn(n, H, generator=g) * 2 + 1).cuda()
    node_in = torch.randn(n, kin, generator=g).cuda()
    pa, pb = pack(wa.cuda(), ba.cuda(), mode, k_pad=H + kin), pack(wb.cuda(), None, mode, k_pad=H + kin)
    A, Bm, hn = (torch.full((n, H), float("nan"), device="cuda") for _ in range(3))
    assert ops.node_halves_supported(kin, n_graphs, rows) or not with_norm
    if with_norm:
        stats = ops.instance_norm_stats(h_raw, n_graphs, rows)
        ops.mpnn_node_halves(h_raw, node_in, pa, pb, A, Bm, stats=stats, rows_per_graph=rows)       # without the h_norm output
        A1 = A.clone()
        ops.mpnn_node_halves(h_raw, node_in, pa, pb, A, Bm, stats=stats, rows_per_graph=rows, h_norm=hn)
        assert torch.equal(A, A1)
        h_ref = ops.instance_norm(h_raw, n_graphs, rows)
        assert rel_l2(hn, h_ref) < 2e-7
        x64 = h_raw.double().view(n_graphs, rows, H)
        h64 = ((x64 - x64.mean(1, keepdim=True)) / (x64.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()).view(n, H)
        assert rel_l2(hn, h64) < 1e-6
    else:
        ops.mpnn_node_halves(h_raw, node_in, pa, pb, A, Bm)
        h_ref, h64 = h_raw, h_raw.double()
    AB = torch.empty(2, n, H, device="cuda")
    T = ops.Term
    ga, gb = (pa, pb) if mode == "fp16x3" else (pack(wa.cuda(), ba.cuda(), "fp16x3", k_pad=H + kin), pack(wb.cuda(), None, "fp16x3", k_pad=H + kin))
    ops.mlp(n, ops.ktiles(T(h_ref, H), T(node_in, kin)), [ga, gb], nl=1, ht=4, ot=4, act=ops.ACT_NONE, y=AB, ldy=H, colgroups=2,
            groups_in_block=True, y_group_stride=n * H)
    assert rel_l2(A, AB[0]) < 5e-7 and rel_l2(Bm, AB[1]) < 5e-7
    x64 = torch.cat([h64, node_in.double()], 1)
    assert rel_l2(A, x64 @ wa.double().cuda().t() + ba.double().cuda()) < 2e-6
    assert rel_l2(Bm, x64 @ wb.double().cuda().t()) < 2e-6


def test_mpnn_2d_rollout_with_fused_node_halves_equals_the_unfused_path():
    import types
    import bench
    from magnet_amd.models.factory import FACTORY
    outs = []
    for fused in (True, False):
        torch.manual_seed(0)
        m = FACTORY["mpnn_2d"](types.SimpleNamespace(**dict(bench.HP, fused_node_halves=fused, hip_graph=False))).cuda().eval()
        batch = bench.synthetic_batch(64, 2, 1234, "cuda")
        outs.append(m.validation_step(batch, 0))
    assert torch.isfinite(outs[0]).all()
    assert rel_l2(outs[0], outs[1]) < 1e-6


@pytest.mark.parametrize("single", [False, True])
@pytest.mark.parametrize("kin,n", [(32, 4 * 4096), (32, 1000 + 13), (64, 2 * 640 + 5), (32, 70), (32, 300 * 128 + 77)])
def test_node_update_persistent_kernel_matches_generic_chain(kin, n, single):
    """magnet_mpnn_node_update_f32 (persistent, W2 resident, W1 streamed, h k tiles last) == the generic two-Linear launch with
    the residual epilogue, and both against fp64 on the host.  single: the one-accumulator packs -> magnet_mpnn_node_update2_f32
    (k_node_update2: two workgroups per CU, both Linears streamed, residual rows read again; round 4) - 300 tiles + a ragged one gives
    every workgroup of its 512-block grid... more than one tile and the last one a partial tile."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    if single and kin != 32:
        pytest.skip("k_node_update2 is instantiated for kin = 32")
    g = torch.Generator().manual_seed(5)
    H = 128
    w1, w2 = torch.randn(H, 2 * H + kin, generator=g) * 0.08, torch.randn(H, H, generator=g) * 0.1
    b1, b2 = torch.randn(H, generator=g) * 0.3, torch.randn(H, generator=g) * 0.3
    h, agg = torch.randn(n, H, generator=g).cuda(), torch.randn(n, H, generator=g).cuda()
    node_in = torch.randn(n, kin, generator=g).cuda()
    u1, u2 = pack(w1.cuda(), b1.cuda(), "fp16x3", k_pad=2 * H + kin), pack(w2.cuda(), b2.cuda(), "fp16x3")
    g1, g2 = u1, u2                                                    # the generic chain always takes the two-accumulator packs
    if single:
        u1, u2 = pack(w1.cuda(), b1.cuda(), "fp16x3s", k_pad=2 * H + kin), pack(w2.cuda(), b2.cuda(), "fp16x3s")
    y = torch.full((n, H), float("nan"), device="cuda")
    ops.mpnn_node_update(h, agg, node_in, u1, u2, y)
    ref = torch.empty(n, H, device="cuda")
    T = ops.Term
    ops.mlp(n, ops.ktiles(T(h, H), T(agg, H), T(node_in, kin)), [g1, g2], nl=2, ht=4, ot=4, act=ops.ACT_SWISH, last_act=True,
            epi=ops.EPI_STORE_RES, res=h, y=ref)
    assert rel_l2(y, ref) < 5e-7
    sw = lambda v: v * torch.sigmoid(v)  # noqa: E731

    def host(hh):
        x64 = torch.cat([hh, agg.double().cpu(), node_in.double().cpu()], 1)
        return hh + sw(sw(x64 @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double())
    assert rel_l2(y, host(h.double().cpu())) < 2e-6
    if n % 512 == 0:          # the statistics of the output from the kernel's own partials == a separate pass over y
        mean, var = ops.mpnn_node_update(h, agg, node_in, u1, u2, y, rows_per_graph=n // 4, stats_out_graphs=4)
        m2, v2 = ops.instance_norm_stats(y, 4, n // 4)
        y64 = y.double().cpu().view(4, n // 4, H)
        assert rel_l2(mean, y64.mean(1)) < 1e-6 and rel_l2(var, y64.var(1, unbiased=False)) < 1e-6
        assert rel_l2(mean, m2) < 1e-6 and rel_l2(var, v2) < 1e-6
        assert rel_l2(y, ref) < 5e-7
    if n % 4 == 0:            # InstanceNorm of the h input applied on load (4 graphs): == normalising first
        raw = h * 2 + 1
        stats = ops.instance_norm_stats(raw, 4, n // 4)
        ops.mpnn_node_update(raw, agg, node_in, u1, u2, y, stats=stats, rows_per_graph=n // 4)
        ops.mpnn_node_update(ops.instance_norm(raw, 4, n // 4), agg, node_in, u1, u2, ref)
        assert rel_l2(y, ref) < 5e-7
        x = raw.double().cpu().view(4, n // 4, H)
        assert rel_l2(y, host(((x - x.mean(1, keepdim=True)) / (x.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()).view(n, H))) < 2e-6


@pytest.mark.parametrize("hidden,out,k_in", [(64, 64, 40), (128, 128, 45), (64, 32, 96)])
def test_chain_kernel_is_bitwise_deterministic_over_launches(hidden, out, k_in):
    """The LDS weight ring is fed by LDS-DMA and hand-placed waits: a missing wait shows up as an intermittent error in single
    32-row wave tiles (an experiment of round 2 did exactly that, profiles/r2/chain_sq_counters.txt).  60 launches, all bitwise
    equal to the first."""
    from magnet_amd import ops
    from magnet_amd.models.backbones.mlp import MLP
    from magnet_amd.models.magnet_common import pack_mlp_ln, run_mlp
    from magnet_amd.packing import pad_to
    torch.manual_seed(0)
    seq = torch.nn.Sequential(MLP(k_in, [hidden] * 4, out), torch.nn.LayerNorm(out)).cuda()
    M, kp = 1013, pad_to(k_in)
    x = torch.zeros(M, kp, device="cuda")
    x[:, :k_in] = torch.randn(M, k_in, device="cuda")
    pk = pack_mlp_ln(seq, k_pad=kp, mode="fp16x3")
    kts = ops.ktiles(ops.Term(x, kp))
    first = run_mlp(M, kts, pk, torch.empty(M, out, device="cuda")).clone()
    for _ in range(60):
        assert torch.equal(run_mlp(M, kts, pk, torch.empty(M, out, device="cuda")), first)


def test_node_kernels_are_bitwise_deterministic_over_launches():
    import types
    import bench
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    torch.manual_seed(0)
    m = FACTORY["mpnn_2d"](types.SimpleNamespace(**bench.HP)).cuda().eval()
    lyr = m._packed().layers[0]
    N, H, kin = 4 * 2048, 128, m.kin
    h, agg = torch.randn(N, H, device="cuda") * 2 + 1, torch.randn(N, H, device="cuda")
    node_in = torch.randn(N, kin, device="cuda")
    st = ops.instance_norm_stats(h, 4, N // 4)
    A, B, g = (torch.empty(N, H, device="cuda") for _ in range(3))
    ops.mpnn_node_halves(h, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, A, B, stats=st, rows_per_graph=N // 4)
    ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, g, stats=st, rows_per_graph=N // 4)
    ref = (A.clone(), B.clone(), g.clone())
    for _ in range(40):
        ops.mpnn_node_halves(h, node_in, lyr.pre_a_l2e, lyr.pre_b_l2e, A, B, stats=st, rows_per_graph=N // 4)
        ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, g, stats=st, rows_per_graph=N // 4)
        assert torch.equal(A, ref[0]) and torch.equal(B, ref[1]) and torch.equal(g, ref[2])


@pytest.mark.parametrize("tw,dim,n,n_graphs", [(10, 2, 4 * 4096, 4), (10, 1, 8192 + 37, 0), (25, 1, 2 * 4160, 2), (20, 1, 8200, 0),
                                               (16, 1, 8192, 1)])
def test_readout_line_kernel_with_instance_norm_on_load(tw, dim, n, n_graphs):
    """magnet_mpnn_readout_norm_f32 / the large-n path of magnet_mpnn_readout_f32 (rows through LDS as whole lines, optional InstanceNorm
    on load) vs the two Conv1d of models/mpnn.py:140-163,197-201 in fp64 on the host, and vs the wave-per-node kernel on a small slice."""
    from magnet_amd import ops
    from magnet_amd.models.mpnn import _READOUT
    torch.manual_seed(7)
    k1, s1, k2 = _READOUT[tw]
    use_swish = not (dim == 1 and tw == 10)
    c1, c2 = torch.nn.Conv1d(1, 8, k1, stride=s1).double(), torch.nn.Conv1d(8, 1, k2).double()
    h_raw = torch.randn(n, 128) * 1.7 + 0.4
    u = torch.randn(n, tw)
    dt = torch.tensor([0.37])
    hx = h_raw.double()
    if n_graphs:
        x = hx.view(n_graphs, n // n_graphs, 128)
        hx = ((x - x.mean(1, keepdim=True)) / (x.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()).view(n, 128)
    with torch.no_grad():
        z = c1(hx[:, None])
        if use_swish:
            z = z * torch.sigmoid(z)
        want = u[:, -1:].double() + 0.37 * torch.arange(1, tw + 1, dtype=torch.float64) * c2(z).squeeze(1)
    f32 = lambda t: t.detach().float().cuda().contiguous()  # noqa: E731
    args = (f32(c1.weight).reshape(-1), f32(c1.bias), k1, s1, f32(c2.weight).reshape(-1), f32(c2.bias), k2, use_swish, dt.cuda())
    hc, uc = h_raw.cuda(), u.cuda()
    if n_graphs:
        stats = ops.instance_norm_stats(hc, n_graphs, n // n_graphs)
        got = ops.mpnn_readout(hc, uc, tw, *args, stats=stats, rows_per_graph=n // n_graphs)
        assert got is not None
    else:
        got = ops.mpnn_readout(hc, uc, tw, *args)
    assert rel_l2(got, want) < 2e-6
    small = ops.mpnn_readout(hx.float().cuda()[:100].contiguous(), uc[:100].contiguous(), tw, *args)     # n < 8192: wave-per-node kernel
    assert rel_l2(small, want[:100]) < 2e-6


def test_resident_convolution_equals_the_streamed_launch_and_is_deterministic(monkeypatch):
    """magnet_conv_resident_f32 (3 x 3 EDSR convolution on a small map, the column group's planes resident in LDS) against the generic
    gathered-tap launch on the same packs (MAGNET_RESIDENT_CONV=0) and against itself over 30 launches, ragged pixel count included"""
    from magnet_amd.models.backbones.edsr import EDSR
    torch.manual_seed(3)
    x = torch.randn(2, 5, 37, 29).cuda()                          # 2 x 1 073 pixels: not a multiple of 64
    outs = []
    for knob in ("1", "0"):
        monkeypatch.setenv("MAGNET_RESIDENT_CONV", knob)
        torch.manual_seed(4)
        net = EDSR(5, 128, 3, 3, 1, "2d").cuda()
        assert net.resident_conv == (knob == "1")
        outs.append(net.rows(x).clone())
        if knob == "1":
            for _ in range(30):
                assert torch.equal(net.rows(x), outs[0])
    assert rel_l2(outs[0], outs[1]) < 5e-7


@pytest.mark.parametrize("mode", ["fp16x3", "bf16x6", "bf16", "fp32"])
def test_device_weight_packing_equals_the_torch_layouts(mode):
    """magnet_pack_linear_f32 (one launch per Linear) against magnet_amd.packing's torch index shuffles, bit for bit: odd sizes
    (zero padding), a column placement map, a transposed view (strides), an explicit k_pad, subnormal / large values; and the
    transposed backward packs of autograd._tpack against the padded-copy construction they replace."""
    from magnet_amd import packing as P
    torch.manual_seed(5)
    ref = {"fp16x3": P.pack_linear_fp16x2, "bf16x6": P.pack_linear_bf16x3, "bf16": P.pack_linear_bf16x1, "fp32": P.pack_linear}[mode]

    def same(a, b):
        return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.view(torch.uint8), b.view(torch.uint8))

    for (n, k, kw) in [(128, 128, {}), (70, 45, {}), (128, 13, dict(k_pad=32)), (33, 257, {}), (128, 161, dict(k_pad=192)),
                       (64, 96, dict(n_pad=128))]:
        w = torch.randn(n, k) * torch.logspace(-6, 2, k)[None, :]          # magnitudes from 1e-6 to 1e2 (fp16 subnormal low pieces)
        b = torch.randn(n)
        want_w, want_b = ref(w, b, **kw)
        got_w, got_b = P.pack(w.cuda(), b.cuda(), mode, **kw)
        assert same(got_w.cpu(), want_w) and torch.equal(got_b.cpu(), want_b), (mode, n, k)
        # column placement (the var column of update_net_1, the INR head's split columns)
        cmap = torch.randperm(pad_to_32(k) + 32)[:k]
        want_w, _ = ref(w, None, col_map=cmap, **{kk: vv for kk, vv in kw.items() if kk == "n_pad"})
        got_w, got_b = P.pack(w.cuda(), None, mode, col_map=cmap, **{kk: vv for kk, vv in kw.items() if kk == "n_pad"})
        assert same(got_w.cpu(), want_w) and float(got_b.abs().max()) == 0.0
        # a transposed view is read through its strides
        wt = torch.randn(k, n)
        want_w, _ = ref(wt.t().contiguous(), None)
        got_w, _ = P.pack(wt.cuda().t(), None, mode)
        assert same(got_w.cpu(), want_w)
    P.raise_if_bad_weights("cuda")                                         # nothing above left the fp16 range


def pad_to_32(n):
    return (n + 31) // 32 * 32


def test_device_packing_flags_weights_outside_the_fp16_range():
    from magnet_amd import packing as P
    w = torch.randn(64, 64)
    w[3, 5] = 7.0e4
    P.pack(w.cuda(), None, "fp16x3")
    with pytest.raises(ValueError, match="65504"):
        P.raise_if_bad_weights("cuda")
    P.raise_if_bad_weights("cuda")                                         # the word was cleared
    w[3, 5] = float("nan")
    P.pack(w.cuda(), None, "fp16x3")
    with pytest.raises(ValueError):
        P.raise_if_bad_weights("cuda")
    P.pack(w.cuda(), None, "bf16x6")                                       # the full-range split does not care
    P.raise_if_bad_weights("cuda")
    # flag_if_abs_reaches (the pre-scaled planes' limit, one status-check launch): below / at the limit, a strided view, a NaN
    v = torch.randn(64, 96).cuda()
    v[5, 70] = 31.0
    P.flag_if_abs_reaches(v, 31.98)
    assert not P.bad_weights("cuda")
    P.flag_if_abs_reaches(v[:, 64:], 30.0)                                 # a column block (not contiguous) that holds the 31.0
    assert P.bad_weights("cuda") and not P.bad_weights("cuda")
    P.flag_if_abs_reaches(v[:, :64], 30.0)                                 # ... and one that does not
    assert not P.bad_weights("cuda")
    v[0, 0] = float("nan")
    P.flag_if_abs_reaches(v, 1e9)
    assert P.bad_weights("cuda")


def test_transposed_backward_packs_equal_the_padded_copy_construction():
    import types
    from magnet_amd import autograd as AG
    from magnet_amd import packing as P
    torch.manual_seed(8)
    n, k, Np, Kp = 64, 45, 64, 96
    W = torch.randn(n, k).cuda()
    cmap = torch.tensor(sorted(torch.randperm(Kp)[:k].tolist()))
    for col_map, first_cols in ((None, None), (cmap, None), (cmap[:30], torch.randperm(k)[:30])):
        m = types.SimpleNamespace(col_map=col_map, first_cols=first_cols)
        Wp = torch.zeros(Np, Kp, device="cuda")
        if first_cols is not None:
            Wp[:n, col_map] = W[:, first_cols]
        elif col_map is not None:
            Wp[:n, col_map] = W
        else:
            Wp[:n, :k] = W
        pk = types.SimpleNamespace()
        for (c0, w) in AG._tiles_of(Kp):
            want = P.pack_linear_bf16x3(Wp[:, c0:c0 + w].t().contiguous().cpu(), None)
            got = AG._tpack(m, pk, 0, c0, w, W, Np, Kp)
            assert torch.equal(got[0].cpu(), want[0]) and float(got[1].abs().max()) == 0.0, (c0, w)


@pytest.mark.parametrize("M,n_out,kc,ntile", [(4096, 128, 128, 6), (1000, 64, 64, 3), (70, 128, 32, 2), (50000, 32, 128, 1)])
def test_weight_gradient_of_several_column_tiles_in_one_launch(M, n_out, kc, ntile):
    """magnet_wgrad_tiles_f32 (gridDim.y = column tiles) == one magnet_wgrad_n_f32 launch per tile (bitwise: same kernel, same
    partials, same reduction order) and == fp64 on the host"""
    import ctypes as C
    from magnet_amd import _lib as L, ops
    g = torch.Generator().manual_seed(3)
    dY = (torch.randn(M, n_out, generator=g) * 1e-6).cuda()
    X = torch.randn(M, 32 + ntile * kc, generator=g).cuda()
    nblk = min(768, (M + 31) // 32)
    ws = torch.empty(ntile * nblk * n_out * kc + nblk * n_out + ntile * n_out * kc + nblk * (128 * 128 + 128) + 128 * 128, device="cuda")
    dW = torch.full((n_out, 16 + ntile * kc), 7.0, device="cuda")
    db = torch.full((n_out,), 7.0, device="cuda")
    L.check(L.lib().magnet_wgrad_tiles_f32(ops._p(dY), n_out, n_out, ops._p(X), X.shape[1], 32, kc, ntile, C.c_long(M), ops._p(dW),
                                           dW.shape[1], 16, ops._p(db), ops._p(ws), 0, ops._stream()), "wgrad_tiles")
    ref, rb = torch.full_like(dW, 7.0), torch.full_like(db, 7.0)
    for t in range(ntile):
        L.check(L.lib().magnet_wgrad_n_f32(ops._p(dY), n_out, n_out, ops._p(X), X.shape[1], 32 + t * kc, kc, C.c_long(M), ops._p(ref),
                                           ref.shape[1], 16 + t * kc, ops._p(rb if t == 0 else None), ops._p(ws), 0, ops._stream()), "wgrad_n")
    assert torch.equal(dW, ref) and torch.equal(db, rb)
    assert torch.equal(dW[:, :16], torch.full((n_out, 16), 7.0, device="cuda"))
    want = dY.double().t() @ X.double()[:, 32:]
    assert rel_l2(dW[:, 16:], want) < 2e-6 and rel_l2(db, dY.double().sum(0)) < 2e-6
    acc = dW.clone()                                                     # accumulate = 1 adds to what is there
    L.check(L.lib().magnet_wgrad_tiles_f32(ops._p(dY), n_out, n_out, ops._p(X), X.shape[1], 32, kc, ntile, C.c_long(M), ops._p(acc),
                                           acc.shape[1], 16, None, ops._p(ws), 1, ops._stream()), "wgrad_tiles")
    assert rel_l2(acc[:, 16:], 2 * want) < 2e-6


@pytest.mark.parametrize("n_runs,width,rows,src_rows", [(9, 128, 4097, 4096), (2, 32, 300, 1000), (5, 256, 77, 500), (1, 64, 10, 64)])
def test_segment_sum_of_several_gathers_in_one_launch(n_runs, width, rows, src_rows):
    """magnet_segment_sum_cols_runs_f32 == one magnet_segment_sum_cols_f32 launch per run (same sums, run order) and == index_add on the host"""
    from magnet_amd import autograd as AG, ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(src_rows, 32 + n_runs * width, generator=g).cuda()
    out0 = torch.randn(rows, width + 8, generator=g).cuda()
    runs, want = [], out0.double().clone()
    for u in range(n_runs):
        idx = torch.randint(0, rows, (src_rows,), generator=g, dtype=torch.int32).cuda()      # row k of x was gathered from row idx[k]
        if u == 0:
            idx[: src_rows // 2] = 3                                                           # one heavy row: many entries
        rowptr, perm = AG._inverse_csr(idx, rows)
        runs.append((rowptr, perm, 32 + u * width))
        want[:, 4:4 + width].index_add_(0, idx.long(), 0.5 * x.double()[:, 32 + u * width:32 + (u + 1) * width])
    got, ref = out0.clone(), out0.clone()
    AG._segment_sum_cols_runs(x, runs, 0, rows, width, 0.5, got, 4)
    for rp, pm, xc in runs:
        AG._segment_sum_cols(x, xc, rp, pm, rows, width, 0.5, ref, 4)
    assert rel_l2(got, want) < 1e-6 and rel_l2(got, ref) < 1e-6
    assert torch.equal(got[:, :4], out0[:, :4]) and torch.equal(got[:, 4 + width:], out0[:, 4 + width:])


@pytest.mark.parametrize("Wd", [128, 64])
@pytest.mark.parametrize("M,mask,chained", [(128, True, False), (5000, True, True), (70, False, False), (33000, True, True), (4096, False, True)])
def test_linear_backward_of_a_hidden_layer_in_one_kernel(M, mask, chained, Wd):
    """magnet_linear_bwd128_f32: dx = (x > 0) * (dz W), dW = dz^T x, db = colsum(dz) against fp64 on the host; gradients ~1e-7 with an
    outlier row (it sets the launch-wide scale of the staged dz); ``chained``: the block maxima come from a previous call's amax_out"""
    from magnet_amd import _lib as L, ops
    from magnet_amd.packing import pack
    g = torch.Generator().manual_seed(21)
    W = (torch.randn(Wd, Wd, generator=g) * 0.1).cuda()
    x = torch.relu(torch.randn(M, Wd, generator=g)).cuda()
    dz = (torch.randn(M, Wd, generator=g) * 1e-7).cuda()
    dz[M // 2] *= 300
    dz[min(3, M - 1)] = 0
    wt = pack(W.t().contiguous(), None, "fp16x3")[0]
    ws = torch.empty(L.lib().magnet_linear_bwd128_workspace(), device="cuda")
    amax = None
    if chained:                                                         # a first call whose dx is this test's dz: its maxima describe dz
        dz0 = (torch.randn(M, Wd, generator=g) * 1e-7).cuda()
        eye = pack(torch.eye(Wd).cuda(), None, "fp16x3")[0]
        dz = torch.empty(M, Wd, device="cuda")
        amax = ops.linear_bwd128(dz0, x, eye, dz, torch.empty(Wd, Wd, device="cuda"), torch.empty(Wd, device="cuda"), ws, mask=False)
        assert rel_l2(dz, dz0) < 1e-6
        assert abs(float(amax[0][:amax[1]].max()) - float(dz.abs().max())) <= 1e-6 * float(dz.abs().max())
    dx = torch.full((M, Wd), float("nan"), device="cuda")
    dW, db = torch.full((Wd, Wd), 7.0, device="cuda"), torch.full((Wd,), 7.0, device="cuda")
    out = ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, mask=mask, amax_in=amax)
    want = dz.double() @ W.double()
    if mask:
        want = want * (x > 0)
    assert rel_l2(dx, want) < 2e-6
    assert rel_l2(dW, dz.double().t() @ x.double()) < 2e-6 and rel_l2(db, dz.double().sum(0)) < 2e-6
    assert abs(float(out[0][:out[1]].max()) - float(dx.abs().max())) <= 1e-6 * float(dx.abs().max())
    dx2, dW2, db2 = torch.empty_like(dx), torch.empty_like(dW), torch.empty_like(db)
    ops.linear_bwd128(dz, x, wt, dx2, dW2, db2, ws, mask=mask, amax_in=amax)
    assert torch.equal(dx, dx2) and torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("D,rows,mean", [(64, 5003, False), (64, 5003, True), (128, 777, True), (32, 4096, False), (48, 300, False), (128, 3, False)])
def test_layernorm_backward_with_block_maxima_and_fused_mean_gather(D, rows, mean):
    """magnet_layernorm_bwd_amax_f32 (16-byte-access kernel for D in {32, 64, 128}, one row per wave otherwise) against fp64 autograd of
    LayerNorm; ``mean``: the rows are CSR-ordered edges and dy is the gradient of their per-target mean (gathered and divided in the load)"""
    import ctypes as C
    from magnet_amd import _lib as L, ops
    g = torch.Generator().manual_seed(4)
    z = torch.randn(rows, D, generator=g).cuda()
    gamma = (torch.rand(D, generator=g) + 0.5).cuda()
    if mean:
        n_nodes = max(rows // 7, 1)
        tgt = torch.sort(torch.randint(0, n_nodes, (rows,), generator=g)).values.to(torch.int32).cuda()
        rowptr = torch.zeros(n_nodes + 1, dtype=torch.int32, device="cuda")
        rowptr[1:] = torch.cumsum(torch.bincount(tgt.long(), minlength=n_nodes), 0).to(torch.int32)
        dy_nodes = (torch.randn(n_nodes, D, generator=g) * 1e-6).cuda()
        deg = (rowptr[1:] - rowptr[:-1]).clamp(min=1).double()
        dy = dy_nodes.double()[tgt.long()] / deg[tgt.long()][:, None]
        src, tp, rp = dy_nodes, tgt, rowptr
    else:
        dy = (torch.randn(rows, D, generator=g) * 1e-6).cuda().double()
        src, tp, rp = dy.float(), None, None
        dy = src.double()
    z64 = z.double().requires_grad_()
    g64 = gamma.double().requires_grad_()
    b64 = torch.zeros(D, dtype=torch.float64, device="cuda", requires_grad=True)
    torch.nn.functional.layer_norm(z64, (D,), g64, b64, 1e-5).backward(dy)
    nblk = min(2048, (rows + 3) // 4)
    dz = torch.full((rows, D), float("nan"), device="cuda")
    part = torch.empty(nblk, 2, D, device="cuda")
    amax = torch.empty(nblk, device="cuda")
    L.check(L.lib().magnet_layernorm_bwd_amax_f32(ops._p(z), D, ops._p(src), D, ops._p(gamma), C.c_long(rows), D, C.c_float(1e-5), ops._p(dz), D,
                                                  ops._p(part), ops._p(amax), ops._p(tp), ops._p(rp), ops._stream()), "layernorm_bwd")
    assert rel_l2(dz, z64.grad) < 2e-6
    assert rel_l2(part[:, 0].double().sum(0), g64.grad) < 2e-6 and rel_l2(part[:, 1].double().sum(0), b64.grad) < 2e-6
    assert abs(float(amax.max()) - float(dz.abs().max())) <= 1e-6 * float(dz.abs().max())


@pytest.mark.parametrize("n", [5, 1000, 262144 * 128 + 3])
def test_swish_backward_leaves_the_block_maxima_of_its_output(n):
    from magnet_amd import ops
    g = torch.Generator().manual_seed(8)
    dr, z = (torch.randn(n, generator=g) * 1e-6).cuda(), torch.randn(n, generator=g).cuda()
    amax = torch.zeros(256, device="cuda")
    out = ops.swish_bwd(dr, z, amax=amax)
    assert torch.equal(out, ops.swish_bwd(dr, z))
    assert float(amax.max()) == float(out.abs().max())


def test_one_accumulator_chain_is_what_the_64_wide_chains_run_and_keeps_small_activations():
    """packing.chain_gemm: a 5-Linear 64-wide chain is packed as "fp16x3s" (high plane and bias times 2^11) and runs
    magnet_mlp_fp16x3s_f32; the low piece of an activation is an UNSCALED fp16 there (a subnormal for small values), so the case
    that has to be shown is small inputs: 1e-3-sized rows through the chain against fp64, and against the two-accumulator form."""
    import os
    from magnet_amd import ops
    from magnet_amd.models.backbones.mlp import MLP
    from magnet_amd.models.magnet_common import pack_mlp_ln, run_mlp
    from magnet_amd.packing import SinglePlanes
    torch.manual_seed(3)
    import copy
    m, norm = MLP(96, [64] * 4, 32), torch.nn.LayerNorm(32)
    seq = torch.nn.Sequential(copy.deepcopy(m), copy.deepcopy(norm)).cuda()
    m, norm = m.double(), norm.double()
    pk = pack_mlp_ln(seq)
    assert all(isinstance(w, SinglePlanes) for w, _ in pk.layers)
    old = os.environ.get("MAGNET_CHAIN_GEMM")
    os.environ["MAGNET_CHAIN_GEMM"] = "fp16x3"
    try:
        pk2 = pack_mlp_ln(seq)
    finally:
        os.environ.pop("MAGNET_CHAIN_GEMM") if old is None else os.environ.__setitem__("MAGNET_CHAIN_GEMM", old)
    assert not any(isinstance(w, SinglePlanes) for w, _ in pk2.layers)
    # same planes up to the scaling of the high one
    a, b = pk.layers[1][0].view(-1, 2, 512).float(), pk2.layers[1][0].view(-1, 2, 512).float()
    assert torch.equal(a[:, 1], b[:, 1]) and torch.equal(a[:, 0], b[:, 0] * 2048.0)
    assert torch.equal(pk.layers[1][1], pk2.layers[1][1] * 2048.0)
    M = 4099
    for scale in (1.0, 1e-3, 30.0):
        x = torch.randn(M, 96) * scale
        y, y2 = torch.empty(M, 32, device="cuda"), torch.empty(M, 32, device="cuda")
        run_mlp(M, ops.ktiles(ops.Term(x.cuda(), 96)), pk, y)
        run_mlp(M, ops.ktiles(ops.Term(x.cuda(), 96)), pk2, y2)
        with torch.no_grad():
            ref = norm(_torch_mlp(m, x.double()))
        # (a LayerNorm of 1e-3-sized rows amplifies by 1 / std: the bar is the two-accumulator form's own error on the same input)
        e1, e2 = rel_l2(y, ref), rel_l2(y2, ref)
        assert e1 < max(2e-6, 3 * e2), (scale, e1, e2)


def test_one_accumulator_pack_flags_weights_out_of_its_range():
    """|w| >= 31.98: 2^11 w does not fit an fp16 - the pack sets the device status word, raise_if_bad_weights reports it"""
    from magnet_amd import packing
    w = torch.randn(64, 64, device="cuda")
    packing.pack(w, None, "fp16x3s")
    packing.raise_if_bad_weights(w.device)
    w[3, 5] = 32.0
    packing.pack(w, None, "fp16x3s")
    with pytest.raises(ValueError):
        packing.raise_if_bad_weights(w.device)
    packing.pack(w, None, "fp16x3")            # the two-accumulator form takes it
    packing.raise_if_bad_weights(w.device)


def test_the_training_note_for_weights_over_31_98_is_left_by_square_hidden_linears_only():
    """a weight in [31.98, 65504) fits the forward's two-accumulator planes; what cannot hold it are the 2^11-scaled planes of
    linear_bwd128 / 64 and of the edge backward, which take square 128- / 64-wide Linears only.  The pack kernel's note (read by
    bad_weights(training=True)) comes from those shapes - a first layer or a 32-wide output Linear with such a weight does not widen
    the process"""
    from magnet_amd import packing
    dev = torch.device("cuda", torch.cuda.current_device())
    packing.bad_weights(dev, training=True)                         # (clear what earlier tests left)
    for shape, noted in (((128, 160), False), ((32, 128), False), ((128, 128), True), ((64, 64), True), ((64, 128), False)):
        w = torch.randn(*shape, device="cuda") * 0.1
        w[1, 2] = 40.0
        packing.pack(w, None, "fp16x3")
        assert not packing.bad_weights(dev) or False, shape         # inference: 40 is fine (and the read clears the word)
        packing.pack(w, None, "fp16x3")
        assert packing.bad_weights(dev, training=True) == noted, shape


@pytest.mark.parametrize("n_graphs,rows", [(2, 50), (1, 128), (3, 37), (2, 32), (1, 7), (4, 32)])
def test_small_update_with_fused_instance_norm_equals_the_two_launches(n_graphs, rows):
    """magnet_mpnn_node_update_norm_small_f32 (the update of a hop and its InstanceNorm in one launch when the whole batch fits one
    workgroup - BASELINE configs[0]) against magnet_mpnn_node_update_f32 followed by magnet_instance_norm_f32, and against fp64."""
    import types
    import bench
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    torch.manual_seed(4)
    hp = dict(bench.HP)
    m = FACTORY["mpnn_2d"](types.SimpleNamespace(**hp)).cuda().eval()
    lyr = m._packed().layers[0]
    N, H, kin = n_graphs * rows, 128, m.kin
    assert kin == 32 and ops.mpnn_node_update_norm_small_supported(kin, N)
    h, agg = torch.randn(N, H, device="cuda") * 2 + 1, torch.randn(N, H, device="cuda")
    node_in = torch.randn(N, kin, device="cuda")
    y = torch.empty(N, H, device="cuda")
    ops.mpnn_node_update(h, agg, node_in, lyr.upd1, lyr.upd2, y)
    want = ops.instance_norm(y, n_graphs, rows)
    got = torch.full((N, H), float("nan"), device="cuda")
    ops.mpnn_node_update_norm_small(h, agg, node_in, lyr.upd1, lyr.upd2, got, rows)
    assert rel_l2(got, want) < 2e-6
    y64 = y.double().view(n_graphs, rows, H)
    ref = ((y64 - y64.mean(1, keepdim=True)) / torch.sqrt(y64.var(1, unbiased=False, keepdim=True) + 1e-5)).view(N, H)
    assert rel_l2(got, ref) < 2e-6
    again = torch.empty_like(got)
    for _ in range(20):
        ops.mpnn_node_update_norm_small(h, agg, node_in, lyr.upd1, lyr.upd2, again, rows)
        assert torch.equal(again, got)
    assert not ops.mpnn_node_update_norm_small_supported(kin, 129) and not ops.mpnn_node_update_norm_small_supported(64, 100)


def test_mpnn_1d_rollout_with_the_fused_small_norm_equals_the_two_launch_path(monkeypatch):
    """C1-sized mpnn (2 x 50 nodes): the rollout with the update kernel applying the InstanceNorm equals the path with a separate
    InstanceNorm launch to rounding (MAGNET_SMALL_NORM=0)"""
    from magnet_amd import ops
    from magnet_amd.models.factory import FACTORY
    from tests.helpers import load_case
    name, hp, salt, batch, d = load_case("mpnn_1d")
    n_rows = batch["u"].shape[0] * batch["u"].shape[1]
    if not ops.mpnn_node_update_norm_small_supported(32, n_rows):
        pytest.skip(f"the fixture's batch has {n_rows} rows: more than one workgroup")
    from oracle.weightgen import fill_state_dict
    from oracle import models as O
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    cb = {k: v.cuda() for k, v in batch.items()}
    outs = []
    for knob in ("1", "0"):
        monkeypatch.setenv("MAGNET_SMALL_NORM", knob)
        model = FACTORY[name](hp)
        model.load_state_dict(oracle.state_dict())
        model = model.cuda().eval()
        with torch.no_grad():
            o = model.validation_step(cb, 0)
        outs.append(o[0] if isinstance(o, (tuple, list)) else o)
    assert torch.is_tensor(outs[0]) and rel_l2(outs[0], outs[1]) < 1e-6 and not torch.equal(outs[0], outs[1])     # (two different paths ran)
