"""GPU parity of the MPNN training step: the HIP backward kernels vs torch autograd.

Op level: each building block against the same op differentiated by torch (fp32 on the GPU or fp64 on the CPU).
Model level: loss and every parameter gradient of ``training_step`` against torch autograd through the oracle
restatement (whose forward is pinned to the reference's golden vectors; the reference itself obtains these
gradients from torch autograd through the same operations, models/mpnn.py:254-297).
"""
import types

import pytest
import torch

from tests.helpers import load_case, load_grads, oracle_grads, rel_l2

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-4      # relative L2 per parameter gradient vs fp64 autograd through the oracle
GOLD_TOL = 1e-3      # vs the reference's own fp32 autograd gradients (both sides carry fp32 rounding)


def test_swish_and_axpy_kernels():
    from magnet_amd import ops
    torch.manual_seed(0)
    for n in (4096, 1001):
        z = (torch.randn(n, device="cuda") * 3).requires_grad_()
        dr = torch.randn(n, device="cuda")
        r = z * torch.sigmoid(z)
        r.backward(dr)
        assert rel_l2(ops.swish_fwd(z.detach()), r.detach()) < 1e-6
        assert rel_l2(ops.swish_bwd(dr, z.detach()), z.grad) < 1e-6
        a, b = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
        assert torch.equal(ops.axpy(a, b, -0.5), a + (-0.5) * b)


@pytest.mark.parametrize("M,kc,xcol", [(1000, 128, 0), (64, 32, 128), (70001, 64, 32), (5, 96, 0)])
def test_wgrad_matches_matmul(M, kc, xcol):
    from magnet_amd import ops
    torch.manual_seed(1)
    dY = torch.randn(M, 128, device="cuda")
    X = torch.randn(M, xcol + kc + 32, device="cuda")
    dW = torch.randn(128, 40 + kc, device="cuda")
    db = torch.randn(128, device="cuda")
    dW0, db0 = dW.clone(), db.clone()
    ws = ops.wgrad_workspace(M, "cuda")
    ops.wgrad(dY, X, xcol, kc, dW, 8, db, ws, accumulate=True)
    want = dW0.double()
    want[:, 8:8 + kc] += dY.double().t() @ X[:, xcol:xcol + kc].double()
    assert rel_l2(dW, want) < 1e-6
    assert rel_l2(db, db0.double() + dY.double().sum(0)) < 1e-6
    # deterministic, and accumulate=False overwrites
    dW2, db2 = torch.full_like(dW, 7.0), torch.full_like(db, 7.0)
    ops.wgrad(dY, X, xcol, kc, dW2, 8, db2, ws, accumulate=False)
    dW3, db3 = torch.full_like(dW, 7.0), torch.full_like(db, 7.0)
    ops.wgrad(dY, X, xcol, kc, dW3, 8, db3, ws, accumulate=False)
    assert torch.equal(dW2, dW3) and torch.equal(db2, db3)
    assert rel_l2(dW2[:, 8:8 + kc], dY.double().t() @ X[:, xcol:xcol + kc].double()) < 1e-6
    assert torch.all(dW2[:, :8] == 7.0)


@pytest.mark.parametrize("M", [1, 33, 4097, 70001])
@pytest.mark.parametrize("with_bias", [True, False])
def test_wgrad_pair_matches_two_launches(M, with_bias):
    """magnet_wgrad_pair_f32 (dY^T h and dY^T node_in in one pass over dY): the two blocks land where two magnet_wgrad_f32 launches put
    them, bit for bit (the same slabs, pieces and accumulation order per block), and match fp64."""
    from magnet_amd import ops
    torch.manual_seed(2)
    dY = torch.randn(M, 128, device="cuda") * 1e-7                  # (gradients of a mean loss)
    X, X2 = torch.randn(M, 128, device="cuda"), torch.randn(M, 32, device="cuda")
    dW = torch.randn(128, 288, device="cuda") * 1e-7
    db = torch.randn(128, device="cuda") * 1e-7 if with_bias else None
    ws = ops.wgrad_workspace(M, "cuda")
    a, ab = dW.clone(), (db.clone() if with_bias else None)
    ops.wgrad_pair(dY, X, 0, X2, 0, a, 0, 256, ab, ws)
    b, bb = dW.clone(), (db.clone() if with_bias else None)
    ops.wgrad(dY, X, 0, 128, b, 0, bb, ws)
    ops.wgrad(dY, X2, 0, 32, b, 256, None, ws)
    assert torch.equal(a, b) and (not with_bias or torch.equal(ab, bb))
    want = dW.double()
    want[:, :128] += dY.double().t() @ X.double()
    want[:, 256:] += dY.double().t() @ X2.double()
    assert rel_l2(a, want) < 1e-6 and torch.equal(a[:, 128:256], dW[:, 128:256])
    c = torch.full_like(dW, 7.0)
    ops.wgrad_pair(dY, X, 0, X2, 0, c, 0, 256, None, ws, accumulate=False)
    assert rel_l2(c[:, :128], dY.double().t() @ X.double()) < 1e-6 and torch.all(c[:, 128:256] == 7.0)


def test_instance_norm_backward():
    from magnet_amd import ops
    from oracle import pyg_semantics as S
    torch.manual_seed(2)
    B, Ns, H = 3, 257, 128
    x = torch.randn(B * Ns, H, device="cuda") * 2 + 1
    dy = torch.randn(B * Ns, H, device="cuda")
    y, var = ops.instance_norm(x, B, Ns, return_var=True)
    dx = ops.instance_norm_bwd(dy, y, var, B, Ns)
    xc = x.double().cpu().requires_grad_()
    batch = torch.arange(B).repeat_interleave(Ns)
    S.instance_norm(xc, batch).backward(dy.double().cpu())
    assert rel_l2(dx, xc.grad) < 1e-5


def test_edge_gather_scatter_kernels():
    from magnet_amd import ops
    torch.manual_seed(3)
    N, E = 300, 4000
    src = torch.randint(0, N, (E,), dtype=torch.int32, device="cuda")
    tgt = torch.randint(0, N - 5, (E,), dtype=torch.int32, device="cuda")      # a few nodes without edges
    csr = ops.csr_by_target(src, tgt, N)
    A, Bm = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
    s, t = csr.src.long(), csr.tgt.long()
    assert torch.equal(ops.edge_pre(A, Bm, csr), A[t] + Bm[s])
    dagg = torch.randn(N, 128, device="cuda")
    deg = torch.bincount(t, minlength=N).float()
    assert rel_l2(ops.edge_mean_bwd(dagg, csr), dagg[t] / deg[t, None]) < 1e-6
    dp = torch.randn(E, 128, device="cuda")
    want_a = torch.zeros(N, 128, dtype=torch.float64, device="cuda").index_add_(0, t, dp.double())
    assert rel_l2(ops.segment_sum128(dp, csr.rowptr, N), want_a) < 1e-6
    bs = ops.csr_by_target(csr.tgt, csr.src, N)
    want_b = torch.zeros(N, 128, dtype=torch.float64, device="cuda").index_add_(0, s, dp.double())
    assert rel_l2(ops.segment_sum128(dp, bs.rowptr, N, perm=bs.perm), want_b) < 1e-6


@pytest.mark.parametrize("tw,dim", [(10, 1), (10, 2), (25, 1), (50, 1)])
def test_readout_backward(tw, dim):
    """reference: torch autograd through the two Conv1d of the read-out, evaluated on the HOST in fp64 (a Conv1d backward on the
    device goes through MIOpen, which is not part of this framework - and which aborted the whole pytest process when this test ran
    late in the full -m gpu session)"""
    from magnet_amd import ops
    from magnet_amd.models.mpnn import _READOUT
    torch.manual_seed(4)
    k1, s1, k2 = _READOUT[tw]
    use_swish = not (dim == 1 and tw == 10)
    n = 1037
    h = torch.randn(n, 128, device="cuda")
    dout = torch.randn(n, tw, device="cuda")
    c1 = torch.nn.Conv1d(1, 8, k1, stride=s1)
    c2 = torch.nn.Conv1d(8, 1, k2)
    dt = torch.tensor([0.37], device="cuda")
    u_last = torch.randn(n, dtype=torch.float64, requires_grad=True)
    hh = h.cpu().double().requires_grad_()
    c1d, c2d = c1.double(), c2.double()
    z = c1d(hh[:, None])
    if use_swish:
        z = z * torch.sigmoid(z)
    out = u_last[:, None] + dt.cpu().double() * torch.arange(1, tw + 1, dtype=torch.float64) * c2d(z).squeeze(1)
    out.backward(dout.cpu().double())
    c1, c2 = c1d, c2d
    f32 = lambda t: t.detach().float().cuda()  # noqa: E731
    dh, du, gp = ops.mpnn_readout_bwd(h, dout, tw, f32(c1.weight).reshape(-1).contiguous(), f32(c1.bias), k1, s1,
                                      f32(c2.weight).reshape(-1).contiguous(), k2, use_swish, dt)
    g = torch.zeros(gp.shape[1], device="cuda")
    ops.reduce_partials(gp, gp.shape[0], gp.shape[1], g)
    assert rel_l2(dh, hh.grad) < 1e-5
    assert rel_l2(du, u_last.grad) < 1e-6
    assert rel_l2(g[:8 * k1], c1.weight.grad.reshape(-1)) < 1e-5
    assert rel_l2(g[8 * k1:8 * k1 + 8], c1.bias.grad) < 1e-5
    assert rel_l2(g[8 * k1 + 8:8 * k1 + 8 + 8 * k2], c2.weight.grad.reshape(-1)) < 1e-5
    assert rel_l2(g[-1:], c2.bias.grad) < 1e-5


def _pair(tag, **extra):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict

    name, hp, salt, batch, d = load_case(tag)
    for k, v in extra.items():
        setattr(hp, k, v)
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp)
    model.load_state_dict(oracle.state_dict())
    return hp, batch, oracle, model.cuda()


def test_fused_edge_backward_matches_autograd():
    """magnet_mpnn_edge_bwd_f32 against torch autograd (fp64) of the same edge block, gradient magnitudes ~1e-7."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    torch.manual_seed(5)
    N, E = 500, 7001
    src = torch.randint(0, N, (E,), dtype=torch.int32, device="cuda")
    tgt = torch.randint(0, N - 3, (E,), dtype=torch.int32, device="cuda")
    csr = ops.csr_by_target(src, tgt, N)
    A, Bm = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
    W2 = torch.randn(128, 128, device="cuda") / 11
    b2 = torch.randn(128, device="cuda") * 0.1
    dagg = torch.randn(N, 128, device="cuda") * 1e-7
    dagg[7] = 0                                                        # a target whose incoming gradient is exactly zero
    s, t = csr.src.long(), csr.tgt.long()
    A64, B64 = A.double().requires_grad_(), Bm.double().requires_grad_()
    W64, b64 = W2.double().requires_grad_(), b2.double().requires_grad_()
    sw = lambda v: v * torch.sigmoid(v)  # noqa: E731
    msg = sw(sw(A64[t] + B64[s]) @ W64.t() + b64)
    deg = torch.bincount(t, minlength=N).clamp(min=1).double()
    agg = torch.zeros(N, 128, dtype=torch.float64, device="cuda").index_add_(0, t, msg) / deg[:, None]
    agg.backward(dagg.double())
    w2p, w2tp = pack(W2, b2, "fp16x3"), pack(W2.t(), None, "fp16x3")
    s1, dq, dp = (torch.empty(E, 128, device="cuda") for _ in range(3))
    dA = torch.full((N, 128), 7.0, device="cuda")
    partials = torch.empty(((E + 31) // 32) * 2 * 128 + 1, device="cuda")
    ops.mpnn_edge_bwd(A, Bm, dagg, csr, w2p[0], w2tp[0], w2p[1], s1, dq, dp, dA, partials)
    assert rel_l2(dA, A64.grad) < 2e-6
    bs = ops.csr_by_target(csr.tgt, csr.src, N)
    assert rel_l2(ops.segment_sum128(dp, bs.rowptr, N, perm=bs.perm), B64.grad) < 2e-6
    gW, gb = torch.zeros(128, 128, device="cuda"), torch.zeros(128, device="cuda")
    ops.wgrad(dq, s1, 0, 128, gW, 0, gb, ops.wgrad_workspace(E, "cuda"))
    assert rel_l2(gW, W64.grad) < 2e-6 and rel_l2(gb, b64.grad) < 2e-6
    assert rel_l2(s1, sw(A64[t] + B64[s]).detach()) < 1e-6


@pytest.mark.parametrize("generation", [2, 1])
@pytest.mark.parametrize("case", ["random", "one_tile", "many_tiles_per_wave", "tiny_gradients", "accumulate", "rows_over_many_tiles", "empty_rows",
                                  "log2e_halves"])
def test_fused_edge_backward_with_weight_gradient_matches_autograd(case, generation):
    """magnet_mpnn_edge_bwd_w_f32 (dW2 / db2 formed inside the kernel, s1 / dq never written) against torch autograd (fp64) of the
    same edge block and against the kernel that writes s1 / dq (bit-identical dp and dA: the data path is the same code)."""
    from magnet_amd import ops
    from magnet_amd.packing import pack
    torch.manual_seed(11)
    # (generation 2 = k_mpnn_edge_bwd_co, the default; 1 = k_mpnn_edge_bwd_wp.  rows_over_many_tiles: rows of 100 - 300 edges - whole
    #  32-edge tiles inside one row; empty_rows: two thirds of the rows have no edge at all)
    from magnet_amd import _lib
    N, E = {"one_tile": (9, 20), "many_tiles_per_wave": (3000, 150001), "rows_over_many_tiles": (60, 12001)}.get(case, (500, 7001))
    scale = 1e-12 if case == "tiny_gradients" else 1e-7
    src = torch.randint(0, N, (E,), dtype=torch.int32, device="cuda")
    tgt = torch.randint(0, max(N - 3, 1), (E,), dtype=torch.int32, device="cuda")
    if case == "empty_rows":
        tgt = tgt - tgt % 3
    was = _lib.lib().magnet_mpnn_edge_bwd_w_generation(generation)
    try:
        if case == "log2e_halves" and generation != 2:       # (the training forward's scaled halves: generation 3 of the kernel only)
            assert not ops.mpnn_edge_bwd_w_takes_l2e(N)
            z = torch.zeros(1, device="cuda")
            with pytest.raises(RuntimeError):
                ops.mpnn_edge_bwd_w(z, z, z, ops.csr_by_target(src, tgt, N), z, z, z, z, z, z, torch.zeros(128, 128, device="cuda"),
                                    torch.zeros(128, device="cuda"), l2e=True)
            return
        _edge_backward_case(case, N, E, scale, src, tgt)
    finally:
        _lib.lib().magnet_mpnn_edge_bwd_w_generation(was)


def _edge_backward_case(case, N, E, scale, src, tgt):
    from magnet_amd import ops
    from magnet_amd.packing import pack
    csr = ops.csr_by_target(src, tgt, N)
    A, Bm = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
    W2 = torch.randn(128, 128, device="cuda") / 11
    b2 = torch.randn(128, device="cuda") * 0.1
    dagg = torch.randn(N, 128, device="cuda") * scale
    dagg[min(7, N - 1)] = 0
    dagg[0, 0] = 40 * scale                                            # an outlier sets the launch-wide scale of the fp16 split
    s, t = csr.src.long(), csr.tgt.long()
    A64, B64 = A.double().requires_grad_(), Bm.double().requires_grad_()
    W64, b64 = W2.double().requires_grad_(), b2.double().requires_grad_()
    sw = lambda v: v * torch.sigmoid(v)  # noqa: E731
    msg = sw(sw(A64[t] + B64[s]) @ W64.t() + b64)
    deg = torch.bincount(t, minlength=N).clamp(min=1).double()
    agg = torch.zeros(N, 128, dtype=torch.float64, device="cuda").index_add_(0, t, msg) / deg[:, None]
    agg.backward(dagg.double())
    w2p, w2tp = pack(W2, b2, "fp16x3"), pack(W2.t(), None, "fp16x3")
    partials = torch.empty(((E + 31) // 32) * 2 * 128 + 1, device="cuda")
    dp, dA = torch.empty(E, 128, device="cuda"), torch.full((N, 128), 7.0, device="cuda")
    gW0, gb0 = (torch.randn(128, 128, device="cuda") * scale, torch.randn(128, device="cuda") * scale) if case == "accumulate" else \
        (torch.zeros(128, 128, device="cuda"), torch.zeros(128, device="cuda"))
    gW, gb = gW0.clone(), gb0.clone()
    if case == "log2e_halves":       # A' = log2(e) A, Bm' = log2(e) Bm in, the gradients of the unscaled block out
        LOG2E = 1.4426950408889634
        ops.mpnn_edge_bwd_w(A * LOG2E, Bm * LOG2E, dagg, csr, w2p[0], w2tp[0], w2p[1], dp, dA, partials, gW, gb, l2e=True)
    else:
        ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2p[0], w2tp[0], w2p[1], dp, dA, partials, gW, gb)
    assert rel_l2(gW - gW0, W64.grad) < 2e-6 and rel_l2(gb - gb0, b64.grad) < 2e-6
    assert rel_l2(dA, A64.grad) < 2e-6
    s1, dq, dp_ref = (torch.empty(E, 128, device="cuda") for _ in range(3))
    dA_ref = torch.full((N, 128), 7.0, device="cuda")
    ops.mpnn_edge_bwd(A, Bm, dagg, csr, w2p[0], w2tp[0], w2p[1], s1, dq, dp_ref, dA_ref, partials)
    assert rel_l2(dp, dp_ref) < 1e-6 and rel_l2(dA, dA_ref) < 1e-6     # (Swish'(p) comes from another evaluation order: not bitwise)
    gW2, gb2 = gW0.clone(), gb0.clone()                                # twice the same launch: bit-identical
    if case == "log2e_halves":
        ops.mpnn_edge_bwd_w(A * LOG2E, Bm * LOG2E, dagg, csr, w2p[0], w2tp[0], w2p[1], dp_ref, dA_ref, partials, gW2, gb2, l2e=True)
    else:
        ops.mpnn_edge_bwd_w(A, Bm, dagg, csr, w2p[0], w2tp[0], w2p[1], dp_ref, dA_ref, partials, gW2, gb2)
    assert torch.equal(gW, gW2) and torch.equal(gb, gb2) and torch.equal(dp, dp_ref) and torch.equal(dA, dA_ref)


@pytest.mark.parametrize("edge_bwd", ["fused_w", "fused", "unfused"])
@pytest.mark.parametrize("teacher_forcing", [False, True])
@pytest.mark.parametrize("tag", ["mpnn_1d", "mpnn_2d"])
def test_training_step_gradients_match_autograd(tag, teacher_forcing, edge_bwd):
    hp, batch, oracle, model = _pair(tag, teacher_forcing=teacher_forcing, edge_bwd=edge_bwd)
    want_loss, want = oracle_grads(oracle, batch, teacher_forcing, hp.loss)
    cb = {k: v.cuda() for k, v in batch.items()}
    model.train()
    model.zero_grad()
    loss = model.training_step(cb, 0)
    assert loss.requires_grad
    loss.backward()
    assert abs(float(loss.detach()) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    worst = {}
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        worst[n] = rel_l2(p.grad, want[n])
    bad = {n: e for n, e in worst.items() if not e < GRAD_TOL}
    assert not bad, bad
    # and against the gradients the reference itself produced (tests/golden/*_grad.npz: norm + strided sample)
    gold_loss, gold = load_grads(tag, teacher_forcing)
    assert abs(float(loss.detach()) - gold_loss) < 2e-5 * max(1.0, abs(gold_loss))
    for n, p in model.named_parameters():
        norm, sample, stride = gold[n]
        assert abs(float(p.grad.norm()) - norm) < GOLD_TOL * norm, n
        assert rel_l2(p.grad.flatten()[::stride], sample) < GOLD_TOL, n


def test_fused_w_backward_leaves_weights_its_scaled_planes_cannot_hold_to_the_full_range_path():
    """k_mpnn_edge_bwd_w keeps 2^11 fp16(W2) in LDS: |w| >= 31.98 must not overflow silently.  One host read per re-pack notices; the
    backward (and everything after it) then runs in the full-range split - materialising bf16x6 edge block, per-Linear node block - and
    the gradients are the oracle's (round 4: this used to raise ValueError)."""
    from magnet_amd import packing
    from tests.helpers import oracle_grads
    packing.narrow()
    try:
        hp, batch, oracle, model = _pair("mpnn_1d", edge_bwd="fused_w")
        cb = {k: v.cuda() for k, v in batch.items()}
        model.train()
        with torch.no_grad():
            model.gnn_layers[1].message_net_2[0].weight[3, 5] = 40.0
            oracle.gnn_layers[1].message_net_2[0].weight[3, 5] = 40.0
        want_loss, want = oracle_grads(oracle, batch, hp.teacher_forcing, hp.loss)
        with pytest.warns(RuntimeWarning, match="31.98"):
            loss = model.training_step(cb, 0)
            loss.backward()
        assert packing.wide()
        assert abs(float(loss.detach()) - want_loss) < 1e-4 * abs(want_loss)
        for n, p in model.named_parameters():
            assert rel_l2(p.grad, want[n].float()) < 2e-3, (n, rel_l2(p.grad, want[n].float()))
    finally:
        packing.narrow()
    hp2, _, _, ok = _pair("mpnn_1d", edge_bwd="fused")                   # the kernel that writes s1 / dq has no such limit
    with torch.no_grad():
        ok.gnn_layers[1].message_net_2[0].weight[3, 5] = 40.0
    ok.train()
    ok.training_step(cb, 0).backward()
    assert not packing.wide()


def test_training_step_is_deterministic_and_optimizer_steps():
    hp, batch, oracle, model = _pair("mpnn_2d")
    cb = {k: v.cuda() for k, v in batch.items()}
    model.train()
    grads = []
    for _ in range(2):
        model.zero_grad()
        model.training_step(cb, 0).backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))
    opt = model.configure_optimizers()["optimizer"]
    l0 = float(model.training_step(cb, 0).detach())
    for _ in range(5):
        opt.zero_grad()
        loss = model.training_step(cb, 0)
        loss.backward()
        opt.step()
    assert float(model.training_step(cb, 0).detach()) < l0          # a few Adam steps on one batch reduce its loss


@pytest.mark.parametrize("tag", ["mpnn_2d", "magnet_gnn_tiny", "magnet_cnn_2d"])
def test_fused_adam_zero_grad_keeps_one_flat_gradient_buffer(tag, monkeypatch):
    """FusedAdam.zero_grad leaves every p.grad as a zeroed view of ONE flat buffer (no ~230 copy_ launches to gather the gradients for
    the Adam kernel / the all-reduce; autograd and the accumulating backward kernels add in place).  The gradients are the same bits
    as with torch's zero_grad (MAGNET_FLAT_GRADS=0), two backward calls accumulate, and step() reads the flat buffer itself."""
    from magnet_amd.optim import flat_gradient
    grads = {}
    for flat in ("1", "0"):
        monkeypatch.setenv("MAGNET_FLAT_GRADS", flat)
        hp, batch, oracle, model = _pair(tag)
        cb = {k: v.cuda() for k, v in batch.items()}
        model.train()
        opt = model.configure_optimizers()["optimizer"]
        params = [p for p in model.parameters() if p.requires_grad]
        opt.zero_grad()
        if flat == "1":
            fg = flat_gradient(params)
            assert fg is not None and fg.numel() == sum(p.numel() for p in params) and float(fg.abs().max()) == 0.0
            assert all(p.grad is not None and p.grad.shape == p.shape for p in params)
        else:
            assert all(p.grad is None for p in params)
        torch.manual_seed(5)
        model.training_step(cb, 0).backward()
        grads[flat] = [p.grad.clone() for p in params]
        if flat == "1":
            fg = flat_gradient(params)
            assert fg is not None and opt._gradient().data_ptr() == fg.data_ptr()     # step() consumes it without a gather
            torch.manual_seed(5)
            model.training_step(cb, 0).backward()                                     # a second backward accumulates
            # (g + g up to the rounding of the in-kernel additions: a kernel that adds its partial sums onto what is there rounds differently)
            assert all(rel_l2(p.grad, 2 * g) < 1e-6 for p, g in zip(params, grads[flat]) if float(g.abs().max()) > 0)
            opt.zero_grad()
            assert float(flat_gradient(params).abs().max()) == 0.0
    assert all(torch.equal(a, b) for a, b in zip(grads["1"], grads["0"]))


@pytest.mark.parametrize("case", ["gnn_512", "gnn_64"])
def test_the_tape_without_the_assembled_first_layer_input_gives_the_same_gradients(case, monkeypatch):
    """MAgNet's edge chains take [x_i | x_j | e] (gathered + dense) or the factored P / Q products + e: every k tile of their first Linear
    is walked back at source-row granularity or is a scaled copy of a source tensor, so the backward reads the sources and the tape
    keeps no x0 (autograd._x0_not_needed; C4b 177 -> 142 GB, C5 211 -> 147 GB at B = 1).  The same sums (the scales are powers of two)
    in a slightly different blocking where two sources share a launch group ([agg | x] of the node chains: one weight-gradient launch
    per source instead of one for both, the row blocks are dealt out per launch) - equal to the last few bits.  By default only for
    x0 of 256 MB and more (there the chains have ONE dense source: same bits) - forced on here."""
    import bench
    from magnet_amd.models.factory import FACTORY
    name, hp, mk = bench.ref_shape_cases()[case]
    batch, _ = mk(True)
    batch = {k: v.cuda() for k, v in batch.items()}
    grads = {}
    for free in ("1", "0"):
        monkeypatch.setenv("MAGNET_X0_FREE", free)
        monkeypatch.setenv("MAGNET_X0_FREE_MIN_MB", "0")
        torch.manual_seed(0)
        model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
        torch.manual_seed(5)
        model.training_step(batch, 0).backward()
        grads[free] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert grads["1"].keys() == grads["0"].keys() and len(grads["1"]) > 50
    assert all(rel_l2(grads["1"][n], grads["0"][n]) < 1e-6 for n in grads["1"] if float(grads["0"][n].abs().max()) > 0)
    assert sum(torch.equal(grads["1"][n], grads["0"][n]) for n in grads["1"]) > len(grads["1"]) // 2       # (most of them to the bit)


def test_fused_adam_matches_torch_adam():
    """magnet_adam_f32 on the flat buffer vs torch.optim.Adam (+ StepLR) on the same gradients, several steps."""
    from magnet_amd.optim import FusedAdam
    torch.manual_seed(6)
    shapes = [(128, 37), (128,), (1, 8, 10), (3,)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdam(pa, lr=1e-2, weight_decay=1e-3)
    ob = torch.optim.Adam(pb, lr=1e-2, weight_decay=1e-3)
    sa = torch.optim.lr_scheduler.StepLR(oa, step_size=3, gamma=0.3)
    sb = torch.optim.lr_scheduler.StepLR(ob, step_size=3, gamma=0.3)
    for it in range(8):
        for x, y in zip(pa, pb):
            g = torch.randn_like(x) * (10.0 ** -(it % 4))
            x.grad, y.grad = g.clone(), g.clone()
        oa.step(), ob.step()
        sa.step(), sb.step()
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
    for x, y in zip(pa, pb):
        assert rel_l2(x, y) < 1e-6


def test_training_with_fused_adam_tracks_torch_adam():
    """five optimizer steps of the HIP training step: FusedAdam (flat gradients, one kernel) vs torch.optim.Adam"""
    losses = {}
    for kind in ("fused", "torch"):
        hp, batch, oracle, model = _pair("mpnn_2d", optimizer=kind)
        cb = {k: v.cuda() for k, v in batch.items()}
        model.train()
        cfg = model.configure_optimizers()
        opt, sch = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
        assert type(opt).__name__ == ("FusedAdam" if kind == "fused" else "Adam")
        out = []
        for _ in range(5):
            opt.zero_grad()
            loss = model.training_step(cb, 0)
            loss.backward()
            opt.step()
            sch.step()
            out.append(float(loss.detach()))
        losses[kind] = out
        if kind == "fused":
            from magnet_amd.optim import flat_gradient
            assert flat_gradient(list(model.parameters())) is not None      # the backward left one flat buffer
    assert losses["fused"][-1] < losses["fused"][0]
    for a, b in zip(losses["fused"], losses["torch"]):
        assert abs(a - b) < 2e-4 * abs(b)


def test_two_ranks_gradient_allreduce_on_gpu_tensors():
    """scripts/ddp_check.py under torch.distributed.run with 2 ranks sharing this GPU (gloo on GPU tensors; RCCL needs
    one device per rank): all-reduced shard gradients == full-batch gradients, parameters identical after Adam."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MAGNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "scripts", "ddp_check.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("DDP_OK") == 2


@pytest.mark.parametrize("hidden,D,epi", [(128, 128, "segmean"), (64, 32, "segmean"), (128, 128, "res"), (64, 10, "store"),
                                          (128, 1, "store"), (64, 5, "store")])
def test_differentiable_fused_mlp_matches_torch_autograd(hidden, D, epi):
    """magnet_amd.autograd.mlp (fused forward, HIP backward) vs the same chain written in torch (fp64 autograd):
    gathered / differenced / scaled / dense sources, 5 Linears + ReLU, LayerNorm (when D is a hidden-like width),
    store | +residual | CSR-mean epilogues."""
    from magnet_amd import autograd as AG, ops
    from magnet_amd.models.magnet_common import _mlp_ln, pack_mlp, pack_mlp_ln
    from magnet_amd.models.backbones.mlp import MLP
    torch.manual_seed(10)
    N, E, Dn = (396 if D == 5 else 90), 700, 32 if hidden == 64 else 128
    src = torch.randint(0, N, (E,), dtype=torch.int32, device="cuda")
    tgt = torch.randint(0, N - 4, (E,), dtype=torch.int32, device="cuda")
    csr = ops.csr_by_target(src, tgt, N)
    s, t = csr.src.long(), csr.tgt.long()
    x = torch.randn(N, Dn, device="cuda")
    e0 = torch.randn(E, Dn, device="cuda")
    nf = torch.randn(N, 32, device="cuda")
    use_ln = D >= 32
    if epi == "segmean":
        net = _mlp_ln(3 * Dn, hidden, 4, D).cuda()
        pk = pack_mlp_ln(net)
        groups = [ops.Term(x, Dn, idx=csr.tgt), ops.Term(x, Dn, idx=csr.src), ops.Term(e0, Dn, scale=4.0)]
        rows = E
    elif epi == "res":
        net = _mlp_ln(Dn + 32, hidden, 4, D).cuda()
        pk = pack_mlp_ln(net)
        groups = [ops.Term(x, Dn), (ops.Term(nf, 32, idx=csr.src[:N].contiguous(), scale=1.0), ops.Term(nf, 32, idx=csr.tgt[:N].contiguous(), scale=-1.0))]
        rows = N
    else:
        net = MLP(Dn, [hidden] * 4, D).cuda()
        pk = pack_mlp(net)
        groups = [ops.Term(x, Dn)]
        rows = N
    res = torch.randn(N, D, device="cuda") if epi == "res" else None
    leaves = [v.requires_grad_() for v in (x, e0, nf)] + ([res.requires_grad_()] if res is not None else [])
    y = AG.mlp(rows, groups, pk, pk.params, epi={"segmean": ops.EPI_SEGMEAN, "res": ops.EPI_STORE_RES, "store": ops.EPI_STORE}[epi],
               res=res, csr=csr if epi == "segmean" else None, out_rows=N if epi == "segmean" else rows)
    gy = torch.randn_like(y) * 1e-6                                  # gradient magnitudes of a mean loss
    y.backward(gy)
    got = {"x": x.grad, "e0": e0.grad, "nf": nf.grad, "res": res.grad if res is not None else None}
    gpar = [p.grad.clone() for p in pk.params]
    # ---- torch fp64 reference of the same computation
    net64 = __import__("copy").deepcopy(net).double()
    x64, e64, nf64 = (v.detach().double().requires_grad_() for v in (x, e0, nf))
    mlp64 = net64[0] if isinstance(net64, torch.nn.Sequential) else net64
    lins = mlp64.linears()
    if epi == "segmean":
        inp = torch.cat([x64[t], x64[s], 4.0 * e64], 1)
    elif epi == "res":
        a_, b_ = csr.src[:N].long(), csr.tgt[:N].long()
        inp = torch.cat([x64, nf64[a_] - nf64[b_]], 1)
    else:
        inp = x64
    h = inp
    for i, lin in enumerate(lins):
        h = lin(h)
        if i < len(lins) - 1:
            h = torch.relu(h)
    if isinstance(net64, torch.nn.Sequential):
        h = net64[1](h)
    if epi == "segmean":
        deg = torch.bincount(t, minlength=N).clamp(min=1).double()
        out = torch.zeros(N, D, dtype=torch.float64, device="cuda").index_add_(0, t, h) / deg[:, None]
    elif epi == "res":
        r64 = res.detach().double().requires_grad_()
        out = h + r64
    else:
        out = h
    assert rel_l2(y, out.detach()) < 2e-6
    out.backward(gy.double())
    tol = 5e-5
    assert rel_l2(got["x"], x64.grad) < tol
    if epi == "segmean":
        assert rel_l2(got["e0"], e64.grad) < tol
    if epi == "res":
        assert rel_l2(got["nf"], nf64.grad) < tol and rel_l2(got["res"], r64.grad) < 1e-6
    ref_par = [p.grad for lin in lins for p in (lin.weight, lin.bias)] + ([net64[1].weight.grad, net64[1].bias.grad] if isinstance(net64, torch.nn.Sequential) else [])
    for i, (a_, b_) in enumerate(zip(gpar, ref_par)):
        assert rel_l2(a_, b_) < tol, i


@pytest.mark.parametrize("pair", [True, False])
@pytest.mark.parametrize("hidden,D,epi", [(128, 128, "segmean"), (128, 128, "res"), (128, 1, "store"), (64, 32, "segmean"), (64, 10, "store")])
def test_differentiable_fused_mlp_with_the_one_kernel_layer_backward(hidden, D, epi, pair, monkeypatch):
    """the same comparison with the hidden 128 -> 128 Linears walked back by csrc/linear_bwd.hip (ops.linear_bwd128_pair: two layers per
    launch; ops.linear_bwd128: one, ``pair`` off or an odd layer left over; the models take that path from 4096 rows on): forced on at
    these row counts, and checked that it really ran"""
    from magnet_amd import autograd as AG, ops
    calls = []
    orig, orig_pair = ops.linear_bwd128, ops.linear_bwd128_pair
    monkeypatch.setattr(AG, "_FUSED_LAYER_ROWS", 1)
    monkeypatch.setattr(AG, "_LINEAR_BWD_PAIR", pair)
    monkeypatch.setattr(ops, "linear_bwd128", lambda *a, **k: (calls.append(a[0].shape[0]), orig(*a, **k))[1])
    monkeypatch.setattr(ops, "linear_bwd128_pair", lambda *a, **k: (calls.extend([a[0].shape[0]] * 2), orig_pair(*a, **k))[1])
    fact = []
    orig_f = AG._first_layer_at_source_rows
    monkeypatch.setattr(AG, "_first_layer_at_source_rows", lambda *a, **k: (fact.append(sum(a[6])), orig_f(*a, **k))[1])
    test_differentiable_fused_mlp_matches_torch_autograd(hidden, D, epi)
    assert len(calls) == (4 if D == hidden else 3), calls            # layers 4..1 (the last Linear only when it is square); a pair counts twice
    # the gathered node terms of the edge chain ([x_i | x_j | e]: 2 of 3 blocks) took the source-granularity first layer
    assert fact == ([2 * (128 if hidden == 128 else 32) // 32] if epi == "segmean" else []), fact


@pytest.mark.parametrize("hidden,D,epi", [(128, 128, "segmean"), (64, 32, "segmean"), (128, 1, "store")])
def test_gradients_do_not_depend_on_whether_the_tape_was_kept_or_recomputed(hidden, D, epi, monkeypatch):
    """the forward launch writing the backward's tape (MAGNET_KEEP_TAPE=1) and the backward recomputing it (=0) give bitwise the same
    gradients (the same kernel code produces the same arrays), with and without the one-kernel layer backward"""
    from magnet_amd import autograd as AG
    grads = {}
    for rows in (1 << 30, 1):
        monkeypatch.setattr(AG, "_FUSED_LAYER_ROWS", rows)
        for keep in ("1", "0"):
            monkeypatch.setenv("MAGNET_KEEP_TAPE", keep)
            seen = []
            orig = AG._Mlp.backward

            def spy(ctx, dy, _orig=orig, _seen=seen):
                out = _orig(ctx, dy)
                _seen.append([g.clone() for g in out if torch.is_tensor(g)])
                return out
            monkeypatch.setattr(AG._Mlp, "backward", staticmethod(spy))
            test_differentiable_fused_mlp_matches_torch_autograd(hidden, D, epi)
            monkeypatch.setattr(AG._Mlp, "backward", staticmethod(orig))
            grads[(rows, keep)] = seen[0]
        a, b = grads[(rows, "1")], grads[(rows, "0")]
        assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("teacher_forcing", [False, True])
def test_magnet_gnn_training_step_gradients(teacher_forcing):
    """MAgNet[GNN] training_step (BASELINE.json configs[4]): loss.backward() through the HIP autograd Functions.

    This net is piecewise linear: 3.2e7 ReLUs per training step, ~150 of them within 1e-6 of the kink (make_golden.py),
    so a forward that differs from torch's CPU kernels in the seventh digit takes the other branch on a few units and the
    gradient - discontinuous there - moves by far more than rounding (with the L1 loss one unit moved the graph-2 gradients
    by 0.6 - 4 %).  The exact check of the backward is therefore made where it is well posed:
      (a) every differentiable op against fp64 autograd (the tests above), and
      (b) here, the decoder chain inside the real training step: its parameter gradients against fp64 autograd of the same
          chain fed with the inputs / output gradients the HIP run itself produced, on the HIP run's own ReLU branches;
    against the oracle and the reference's own gradients (tests/golden/magnet_gnn_tiny_grad.npz) the test asks for what
    a kink-limited comparison can give: equal loss, the half of the network upstream of the flipped units (LR-mesh graph
    net, INR decoder) to rounding, everything within a few percent and in the same direction."""
    import copy
    from magnet_amd import autograd as AG, ops
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    from tests.helpers import oracle_magnet_gnn_grads
    name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
    hp.teacher_forcing = teacher_forcing
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp)
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    want_loss, want = oracle_magnet_gnn_grads(oracle, batch, teacher_forcing, hp.loss)
    cb = {k: v.cuda() for k, v in batch.items()}
    calls, orig = [], AG.mlp

    def spy(M, groups, pk, params, **kw):
        y = orig(M, groups, pk, params, **kw)
        if pk is model._packed().decoder:
            rec = {"x": groups[0].t.detach().clone()}
            y.register_hook(lambda g, rec=rec: rec.__setitem__("dy", g.detach().clone()))
            calls.append(rec)
        return y

    AG.mlp = spy
    try:
        model.zero_grad()
        loss = model.training_step(cb, 0)
        loss.backward()
    finally:
        AG.mlp = orig
    assert abs(float(loss.detach()) - want_loss) < 2e-5 * max(1.0, abs(want_loss))
    # (b) the decoder chain on the run's own inputs and output gradients
    net = copy.deepcopy(model._decoder.node_fn).double()
    lins = net.linears()
    # ReLU branches of the reference chain = the HIP run's own (the hidden activations its backward recomputes, layer by
    # layer): with ~1e6 hidden units per call a few pre-activations sit within fp32 rounding of zero, and an fp64 forward
    # would put them on the other side
    pk = model._packed().decoder
    for rec in calls:
        x32 = rec["x"]
        xh = AG._pad_cols(x32, (x32.shape[1] + 31) // 32 * 32)
        M, nl = x32.shape[0], len(lins)
        X = [torch.empty_like(xh)] + [torch.empty(M, pk.ht * 32, device=xh.device) for _ in range(nl - 1)]
        AG._recompute_tape(M, ops.ktiles(ops.Term(xh, xh.shape[1])), pk, nl, X[0], X, None, xh.shape[1])   # the backward's own tape launch
        h = x32.double()
        for i, lin in enumerate(lins):
            h = lin(h)
            if i < len(lins) - 1:
                h = h * (X[i + 1][:, :h.shape[1]] > 0).double()
        h.backward(rec["dy"].double())
    assert len(calls) == 2
    for i, lin in enumerate(lins):
        got_w, got_b = model._packed().decoder.params[2 * i].grad, model._packed().decoder.params[2 * i + 1].grad
        assert rel_l2(got_w, lin.weight.grad) < 5e-6 and rel_l2(got_b, lin.bias.grad) < 5e-6, i
    # kink-limited comparison with the oracle / the reference
    gold_loss, gold = load_grads("magnet_gnn_tiny", teacher_forcing)
    assert abs(float(loss.detach()) - gold_loss) < 2e-5 * max(1.0, abs(gold_loss))
    errs = {}
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        errs[n] = rel_l2(p.grad, want[n])
        cos = float((p.grad.double().cpu().flatten() @ want[n].double().flatten()) / (p.grad.double().norm().cpu() * want[n].double().norm()).clamp_min(1e-300))
        # (kink-limited: ONE of the step's 3.15e7 units takes the other branch in the fp64 run - tests/test_mask_sync_gpu.py counts them - and the
        #  L1 sign of a few entries; the well-posed comparison, 3e-6 on every parameter, is made there.  Measured worst here: 1.3e-3)
        assert errs[n] < 1e-2 and cos > 0.9999, (n, errs[n], cos)
        norm, sample, stride = gold[n]
        assert abs(float(p.grad.norm()) - norm) < 1e-2 * norm, n
        if sample.numel() >= 16:                                     # a handful of samples of a 64-entry bias is noise
            assert rel_l2(p.grad.flatten()[::stride], sample) < 1e-2, n
    upstream = [e for n, e in errs.items() if n.startswith(("encoder.", "processor.", "proj_head", "projector."))]
    assert len(upstream) > 90 and max(upstream) < 2e-5, max(upstream)


@pytest.mark.parametrize("tag", ["magnet_gnn_tiny", "magnet_cnn_2d"])
def test_magnet_optimizer_steps_reduce_the_loss(tag):
    """a few FusedAdam steps of a MAgNet model on one batch: packed-weight caches (incl. EDSR's own) follow the updates"""
    from magnet_amd.models.factory import FACTORY
    name, hp, salt, batch, d = load_case(tag)
    torch.manual_seed(0)
    model = FACTORY[name](hp).cuda().train()
    cb = {k: v.cuda() for k, v in batch.items()}
    opt = model.configure_optimizers()["optimizer"]
    assert type(opt).__name__ == "FusedAdam"
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = model.training_step(cb, 0)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    # the inference path sees the updated weights too: its rollout loss equals a fresh model loaded with the trained weights
    pred = model.validation_step(cb, 0)
    fresh = FACTORY[name](hp).cuda().eval()
    fresh.load_state_dict(model.state_dict())
    assert torch.isfinite(pred).all() and rel_l2(pred, fresh.validation_step(cb, 0)) < 1e-6


def test_differentiable_edsr_matches_torch_autograd():
    """EDSR.rows_grad: every convolution a differentiable gathered-tap Linear (HIP forward + backward) vs torch conv autograd"""
    from magnet_amd.models.backbones.edsr import EDSR
    from oracle.models import EDSR as OracleEDSR
    torch.manual_seed(12)
    for mode, shape in (("2d", (2, 5, 9, 7)), ("1d", (2, 5, 21))):
        ref = OracleEDSR(shape[1], 64, 2, 3, mode).double()
        net = EDSR(shape[1], 64, 2, 3, 1, mode)
        net.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        net = net.cuda()
        x = torch.randn(*shape)
        xg = x.cuda().requires_grad_()
        rows = net.rows_grad(xg)
        x64 = x.double().requires_grad_()
        want = ref(x64)
        want_rows = want.reshape(shape[0], 64, -1).permute(0, 2, 1).reshape(-1, 64)
        assert rel_l2(rows, want_rows.detach()) < 2e-6
        g = torch.randn_like(rows) * 1e-5
        rows.backward(g)
        want_rows.backward(g.double().cpu())
        assert rel_l2(xg.grad, x64.grad) < 5e-5
        refp = dict(ref.named_parameters())
        for n, p in net.named_parameters():
            assert rel_l2(p.grad, refp[n].grad) < 5e-5, (mode, n)


@pytest.mark.parametrize("teacher_forcing", [False, True])
@pytest.mark.parametrize("tag", ["magnet_cnn_1d", "magnet_cnn_2d"])
def test_magnet_cnn_training_step_gradients(tag, teacher_forcing):
    """MAgNet[CNN] training_step through the HIP autograd Functions (EDSR, grid INR, joint graph, decoder, Euler): equal loss;
    kink-limited agreement (see test_magnet_gnn_training_step_gradients) with fp64 autograd through the oracle and with the
    reference's own gradients (tests/golden/magnet_cnn_*_grad.npz)."""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    from tests.helpers import oracle_magnet_cnn_grads
    name, hp, salt, batch, d = load_case(tag)
    hp.teacher_forcing = teacher_forcing
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp)
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    want_loss, want = oracle_magnet_cnn_grads(oracle, batch, teacher_forcing, hp.loss, dtype=torch.float32)   # the oracle's grid_sample path is fp32
    cb = {k: v.cuda() for k, v in batch.items()}
    model.zero_grad()
    loss = model.training_step(cb, 0)
    loss.backward()
    assert abs(float(loss.detach()) - want_loss) < 2e-5 * max(1.0, abs(want_loss))
    gold_loss, gold = load_grads(tag, teacher_forcing)
    assert abs(float(loss.detach()) - gold_loss) < 2e-5 * max(1.0, abs(gold_loss))
    errs = {}
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        errs[n] = rel_l2(p.grad, want[n])
        cos = float((p.grad.double().cpu().flatten() @ want[n].double().flatten()) / (p.grad.double().norm().cpu() * want[n].double().norm()).clamp_min(1e-300))
        assert errs[n] < 1e-2 and cos > 0.9999, (n, errs[n], cos)         # (kink-limited; tests/test_mask_sync_gpu.py has the well-posed 3e-6)
        norm, sample, stride = gold[n]
        assert abs(float(p.grad.norm()) - norm) < 1e-2 * norm, n
    assert sorted(errs.values())[len(errs) // 2] < 1e-5 and max(errs.values()) < 1e-2       # measured: 6e-7 ... 2e-6, 1.3e-3


def test_fused_adam_checkpoint_resume():
    """optimizer.state_dict() / load_state_dict() carry the flat moments and the step count: a resumed run continues bitwise"""
    from magnet_amd.optim import FusedAdam
    torch.manual_seed(13)
    def make():
        torch.manual_seed(13)
        return [torch.nn.Parameter(torch.randn(64, 33, device="cuda")), torch.nn.Parameter(torch.randn(7, device="cuda"))]
    grads = [[torch.randn(64, 33, device="cuda"), torch.randn(7, device="cuda")] for _ in range(6)]
    a = make()
    oa = FusedAdam(a, lr=1e-2, weight_decay=1e-3)
    for g in grads[:3]:
        for p, gg in zip(a, g):
            p.grad = gg.clone()
        oa.step()
    ckpt_params, ckpt_opt = [p.detach().clone() for p in a], oa.state_dict()
    for g in grads[3:]:
        for p, gg in zip(a, g):
            p.grad = gg.clone()
        oa.step()
    b = make()
    ob = FusedAdam(b, lr=1e-2, weight_decay=1e-3)
    with torch.no_grad():
        for p, c in zip(b, ckpt_params):
            p.copy_(c)
    ob.load_state_dict(ckpt_opt)
    for g in grads[3:]:
        for p, gg in zip(b, g):
            p.grad = gg.clone()
        ob.step()
    assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("tw,dim", [(50, 1), (20, 1), (16, 1)])
def test_mpnn_training_gradients_other_time_windows(tw, dim):
    """the other read-out variants of mpnn.py:140-163 (tw = 50 pads node_in to 64 columns: two k tiles in every node-level
    backward GEMM): training_step gradients vs fp64 autograd through the oracle (smooth net: tight tolerance)."""
    import types
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    hp = types.SimpleNamespace(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l2", hidden_features=128, hidden_layer=5,
                               time_window=tw, teacher_forcing=False, neighbors=3)
    g = torch.Generator().manual_seed(tw)
    B, N, nt = 2, 40, 3 * tw
    batch = {"u": torch.randn(B, N, nt, generator=g), "x": torch.linspace(0, 16, N)[None, :, None].repeat(B, 1, 1),
             "t": torch.linspace(0, 4, nt)[None].repeat(B, 1)}
    oracle = O.build("mpnn", hp)
    fill_state_dict(oracle, 40 + tw)
    model = FACTORY["mpnn"](hp)
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    want_loss, want = oracle_grads(oracle, batch, False, hp.loss)
    loss = model.training_step({k: v.cuda() for k, v in batch.items()}, 0)
    loss.backward()
    assert abs(float(loss.detach()) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    bad = {n: rel_l2(p.grad, want[n]) for n, p in model.named_parameters() if not rel_l2(p.grad, want[n]) < GRAD_TOL}
    assert not bad, bad


@pytest.mark.parametrize("teacher_forcing", [False])          # (teacher forcing only drops the carry between the forward calls: small fixtures)
def test_mpnn_2d_training_gradients_on_a_mesh_of_the_benchmark_kind(teacher_forcing):
    """The golden mpnn_2d fixture is a 12 x 12 mesh: small-graph kernels.  Here the paths the benchmark's training step takes - the
    stream-tiled forward edge kernel (rows of 33 edges), the fused backward of the edge block (k_mpnn_edge_bwd_co: rows over two and
    three tiles, the by-source sum of dp), the one-kernel Linear backward of the node update (>= 4096 rows) - on a 72 x 72 mesh, B = 2:
    every parameter gradient of training_step against fp64 autograd through the oracle (models/mpnn_2d.py:254-297)."""
    import types
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    hp = types.SimpleNamespace(lr=1e-3, weight_decay=0.0, factor=0.3, step_size=50, loss="l2", hidden_features=128, hidden_layer=5,
                               time_window=10, teacher_forcing=teacher_forcing, neighbors=4)
    W, B, nt = 72, 2, 30
    g = torch.Generator().manual_seed(72)
    lin = torch.linspace(0, 1, W)
    xy = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)
    # smooth fields (a rollout on noise saturates nothing but gives gradients of 1e-9: keep them at the size a real step has)
    k = torch.randn(B, 1, 4, generator=g)
    u = torch.sin(6.28 * (xy[None, :, :1] * k[..., :1] + xy[None, :, 1:] * k[..., 1:2]) + k[..., 2:3]) + 0.1 * torch.randn(B, W * W, 1, generator=g)
    u = u * torch.linspace(1.0, 0.7, nt)[None, None, :] + 0.05 * torch.randn(B, W * W, nt, generator=g)
    batch = {"u": u.float(), "x": xy[None].repeat(B, 1, 1), "t": torch.linspace(0, 1, nt)[None].repeat(B, 1)}
    oracle = O.build("mpnn_2d", hp)
    fill_state_dict(oracle, 72)
    model = FACTORY["mpnn_2d"](hp)
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    want_loss, want = oracle_grads(oracle, batch, teacher_forcing, hp.loss)
    loss = model.training_step({k_: v.cuda() for k_, v in batch.items()}, 0)
    loss.backward()
    assert abs(float(loss.detach()) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    errs = {n: rel_l2(p.grad, want[n]) for n, p in model.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"[mpnn_2d 72 x 72, tf={teacher_forcing}] worst parameter gradient {worst}: {errs[worst]:.2e}")
    bad = {n: e for n, e in errs.items() if not e < GRAD_TOL}
    assert not bad, bad
    assert model._topo.csr.E > 30 * B * W * W and model._topo.csr.N == B * W * W      # (rows of ~33 edges: the stream / co-operative kernels ran)


def test_differentiable_mlp_shared_rows_source():
    """a dense source shared by ``rdiv`` consecutive rows (row r reads t[r // rdiv]): forward and the fixed-size segment sum
    of its backward, vs torch autograd"""
    from magnet_amd import autograd as AG, ops
    from magnet_amd.models.magnet_common import pack_mlp
    from magnet_amd.models.backbones.mlp import MLP
    torch.manual_seed(14)
    rows, rdiv = 250, 5
    net = MLP(64, [64] * 4, 32).cuda()
    pk = pack_mlp(net)
    a = torch.randn(rows, 32, device="cuda", requires_grad=True)
    b = torch.randn(rows // rdiv, 32, device="cuda", requires_grad=True)
    y = AG.mlp(rows, [ops.Term(a, 32), ops.Term(b, 32, rdiv=rdiv)], pk, pk.params)
    gy = torch.randn_like(y)
    y.backward(gy)
    net64 = __import__("copy").deepcopy(net).double()
    a64, b64 = a.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    h = torch.cat([a64, b64.repeat_interleave(rdiv, 0)], 1)
    lins = net64.linears()
    for i, lin in enumerate(lins):
        h = lin(h)
        if i < len(lins) - 1:
            h = torch.relu(h)
    assert rel_l2(y, h.detach()) < 2e-6
    h.backward(gy.double())
    assert rel_l2(a.grad, a64.grad) < 5e-5 and rel_l2(b.grad, b64.grad) < 5e-5


def test_rccl_process_group_runs_the_training_collectives_on_one_rank():
    """scripts/rccl_check.py in a fresh process: backend "nccl" (= RCCL), world size 1 on cuda:0 - the flat-gradient all-reduce after
    a real HIP backward, the scalar MAX / SUM reductions, all-gather, barriers.  Two RCCL ranks cannot share this box's single GPU
    (the 2-rank test above runs over gloo); this is the part of the N > 1 path that CAN execute on RCCL here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "rccl_check.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    if "RCCL_INIT_FAILED" in r.stdout:
        pytest.skip("no RCCL communicator on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tag", ["mpnn_2d", "magnet_gnn_tiny", "magnet_cnn_2d"])
def test_training_step_replayed_from_one_hip_graph_equals_the_eager_steps(tag):
    """graphs.TrainStepGraph: training_step + backward captured once, replayed - including the in-place re-pack of
    the weights the optimizer moved; the zero fill of the gradients, the range words, the Adam launch stay outside.  Six steps: the same
    losses and the same final parameters as six eager steps, bit for bit; every step after the first capture is a replay; the host
    generator ends where the eager run leaves it (MAgNet[GNN] draws its input noise there)."""
    from magnet_amd.graphs import TrainStepGraph
    runs = {}
    for mode in ("eager", "graph"):
        hp, batch, oracle, model = _pair(tag, lr=1e-4)
        cb = {k: v.cuda() for k, v in batch.items()}
        model.train()
        opt = model.configure_optimizers()["optimizer"]
        step = TrainStepGraph(model, opt) if mode == "graph" else None
        torch.manual_seed(3)
        losses = []
        for _ in range(6):
            if step is not None:
                losses.append(float(step(cb).detach()))
            else:
                opt.zero_grad()
                loss = model.training_step(cb, 0)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        runs[mode] = (losses, [p.detach().clone() for p in model.parameters()], torch.get_rng_state().clone())
        if step is not None:
            assert step.replays == 6 and step.eager_steps == 0, (step.replays, step.eager_steps)
    assert runs["eager"][0] == runs["graph"][0], (runs["eager"][0], runs["graph"][0])
    assert all(torch.equal(a, b) for a, b in zip(runs["eager"][1], runs["graph"][1]))
    assert torch.equal(runs["eager"][2], runs["graph"][2])
    assert runs["eager"][0][-1] < runs["eager"][0][0]


def test_ddp_wrapper_sees_every_parameter_gradient():
    """the backward kernels' in-place parameter gradients (autograd.accumulate_param_grads, the MPNN rollout's flat
    buffer) bypass AccumulateGrad - and with it DistributedDataParallel's reducer hooks.  Inside a DDP forward the shortcut must be
    off: scripts/ddp_wrapper_check.py wraps training_step of an MPNN and a MAgNet model in torch DDP (one rank, RCCL), runs two steps,
    counts the post-accumulate hooks and compares the gradients with the un-wrapped model's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29548")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "ddp_wrapper_check.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    if "RCCL_INIT_FAILED" in r.stdout:
        pytest.skip("no RCCL communicator on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0 and "DDP_WRAPPER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("tag", ["mpnn_2d", "magnet_gnn_tiny", "magnet_cnn_2d"])
def test_parameter_hooks_fire_after_fused_adam_zero_grad(tag):
    """A parameter that carries a hook is never given its gradient behind autograd's back (autograd.takes_inplace_grad): after
    ``FusedAdam.zero_grad`` (flat views installed - the in-place path's precondition) ``Tensor.register_hook`` and
    ``register_post_accumulate_grad_hook`` fire once per parameter and the gradients are the hook-free model's bits."""
    hp, batch, oracle, model = _pair(tag)
    cb = {k: v.cuda() for k, v in batch.items()}
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt.zero_grad()
    torch.manual_seed(5)
    model.training_step(cb, 0).backward()
    want = {n: p.grad.clone() for n, p in params}
    seen_grad, seen_post = {}, {}
    handles = []
    for n, p in params:
        handles.append(p.register_hook(lambda g, n=n: seen_grad.__setitem__(n, seen_grad.get(n, 0) + 1)))
        handles.append(p.register_post_accumulate_grad_hook(lambda p_, n=n: seen_post.__setitem__(n, seen_post.get(n, 0) + 1)))
    opt.zero_grad()
    torch.manual_seed(5)
    model.training_step(cb, 0).backward()
    assert all(seen_post.get(n, 0) == 1 for n, _ in params), [n for n, _ in params if seen_post.get(n, 0) != 1][:5]
    assert all(seen_grad.get(n, 0) >= 1 for n, _ in params), [n for n, _ in params if not seen_grad.get(n)][:5]
    assert all(rel_l2(p.grad, want[n]) < 1e-6 for n, p in params if float(want[n].abs().max()) > 0)
    for h in handles:
        h.remove()


@pytest.mark.parametrize("W", [128, 64])
def test_linear_bwd_generations_agree_with_fp64(W):
    """magnet_linear_bwd128_f32 / 64 (csrc/linear_bwd.hip): k_linear_bwd_r in its shapes (generation 1 = the size rule, 2 / 3 = the two
    128-wide shapes forced) and k_linear_bwd<NT> (generation 0) against fp64 on ragged row counts, strided x, both mask settings; block
    maxima exact; repeat launches bit-identical"""
    from magnet_amd import _lib as L, ops
    from magnet_amd.packing import pack
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    Wt = torch.randn(W, W, device="cuda", generator=g) * 0.1
    wt = pack(Wt.t().contiguous(), None, "fp16x3")[0]
    ws = torch.empty(lib.magnet_linear_bwd128_workspace(), device="cuda")
    prev = lib.magnet_linear_bwd128_generation(-1)
    try:
        for M in (1, 33, 4097, 70001):
            for mask in (True, False):
                dz = torch.randn(M, W, device="cuda", generator=g) * 1e-7
                xf = torch.relu(torch.randn(M, W + 32, device="cuda", generator=g))
                x = xf[:, 16:16 + W]
                want_dx = (dz.double() @ Wt.double()) * ((x > 0).double() if mask else 1.0)
                want_dW, want_db = dz.double().t() @ x.double(), dz.double().sum(0)
                for gen in ((0, 1, 2, 3) if W == 128 else (0, 1)):
                    lib.magnet_linear_bwd128_generation(gen)
                    L._QUERIES.clear()                                   # (the grid query follows the generation)
                    out = []
                    for _ in range(2):
                        dx, dW, db = torch.full((M, W), 7.0, device="cuda"), torch.empty(W, W, device="cuda"), torch.empty(W, device="cuda")
                        am, n = ops.linear_bwd128(dz, x, wt, dx, dW, db, ws, mask=mask)
                        out.append((dx, dW, db))
                    dx, dW, db = out[0]
                    assert rel_l2(dx, want_dx) < 1e-6 and rel_l2(dW, want_dW) < 2e-6 and rel_l2(db, want_db) < 1e-6, (M, mask, gen)
                    assert float(am[:n].max()) == float(dx.abs().max()), (M, mask, gen)
                    assert all(torch.equal(a, b) for a, b in zip(out[0], out[1])), (M, mask, gen)
    finally:
        lib.magnet_linear_bwd128_generation(prev)
        L._QUERIES.clear()


@pytest.mark.parametrize("W", [128, 64])
def test_linear_bwd_pair_agrees_with_fp64_and_with_two_single_layer_launches(W):
    """magnet_linear_bwd128_pair_f32 (k_linear_bwd_r2: two consecutive hidden Linears, the gradient between them stays on the chip) against
    fp64 and against two magnet_linear_bwd128_f32 launches, ragged row counts, strided x, accumulate, amax handed in / computed, and a
    second layer whose gradient is 2^20 times smaller than the bound its scale comes from; block maxima exact; repeats bit-identical"""
    from magnet_amd import _lib as L, ops
    from magnet_amd.packing import pack
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(6)
    ws = torch.empty(lib.magnet_linear_bwd128_workspace(), device="cuda")
    for case in ("plain", "small_wa"):
        WA = torch.randn(W, W, device="cuda", generator=g) * (0.1 if case == "plain" else 1e-3)
        if case == "small_wa":
            WA[3, 5] = 20.0                                   # one large weight: the column-sum bound is far above what dzB reaches
        WB = torch.randn(W, W, device="cuda", generator=g) * 0.1
        wtA, wtB = pack(WA.t().contiguous(), None, "fp16x3")[0], pack(WB.t().contiguous(), None, "fp16x3")[0]
        for M in (1, 33, 4097, 70001):
            dz = torch.randn(M, W, device="cuda", generator=g) * 1e-7
            x1f = torch.relu(torch.randn(M, W + 32, device="cuda", generator=g))
            x1 = x1f[:, 16:16 + W]
            x2 = torch.relu(torch.randn(M, W, device="cuda", generator=g))
            dzB = (dz.double() @ WA.double()) * (x1 > 0).double()
            want_dx = (dzB @ WB.double()) * (x2 > 0).double()
            want = (want_dx, dz.double().t() @ x1.double(), dz.double().sum(0), dzB.t() @ x2.double(), dzB.sum(0))
            # (sums of terms of both signs: an error is measured against the sum of the terms' magnitudes - 'small_wa' makes one column of dzB
            #  2e4 times larger than the rest and its column sum a thousandth of that)
            scale = (want_dx.abs(), dz.double().abs().t() @ x1.double(), dz.double().abs().sum(0), dzB.abs().t() @ x2.double(), dzB.abs().sum(0))
            out = []
            for rep in range(2):
                dx = torch.full((M, W), 7.0, device="cuda")
                dWA, dbA, dWB, dbB = (torch.empty(W, W, device="cuda"), torch.empty(W, device="cuda"),
                                      torch.empty(W, W, device="cuda"), torch.empty(W, device="cuda"))
                am, n = ops.linear_bwd128_pair(dz, x1, x2, wtA, wtB, dx, dWA, dbA, dWB, dbB, ws)
                out.append((dx, dWA, dbA, dWB, dbB))
            for got, w_, sc_, tol in zip(out[0], want, scale, (1e-6, 2e-6, 1e-6, 3e-6, 2e-6)):
                if case == "plain":
                    assert rel_l2(got, w_) < tol, (case, M, rel_l2(got, w_))
                err = float((got.double() - w_).norm() / sc_.norm().clamp_min(1e-300))
                assert err < 3e-7, (case, M, err)
            assert float(am[:n].max()) == float(out[0][0].abs().max()), (case, M)
            assert all(torch.equal(a, b) for a, b in zip(out[0], out[1])), (case, M)
            # against the two single-layer launches, with the maxima handed in and accumulation into what is there
            mid, dW1, db1 = torch.empty(M, W, device="cuda"), torch.ones(W, W, device="cuda"), torch.ones(W, device="cuda")
            dx2, dW2, db2 = torch.empty(M, W, device="cuda"), torch.ones(W, W, device="cuda"), torch.ones(W, device="cuda")
            am1 = ops.linear_bwd128(dz, x1, wtA, mid, dW1, db1, ws, mask=True, accumulate=True)
            ops.linear_bwd128(mid, x2, wtB, dx2, dW2, db2, ws, mask=True, amax_in=am1, accumulate=True)
            dx, a1, a2, a3, a4 = (torch.empty(M, W, device="cuda"), torch.ones(W, W, device="cuda"), torch.ones(W, device="cuda"),
                                  torch.ones(W, W, device="cuda"), torch.ones(W, device="cuda"))
            amx = torch.zeros(256, device="cuda")
            amx[0] = dz.abs().max()
            ops.linear_bwd128_pair(dz, x1, x2, wtA, wtB, dx, a1, a2, a3, a4, ws, amax_in=(amx, 256), want_amax=False, accumulate=True)
            for got, ref, sc_ in zip((dx, a1 - 1, a2 - 1, a3 - 1, a4 - 1), (dx2, dW1 - 1, db1 - 1, dW2 - 1, db2 - 1), scale):
                # (the `+ 1 - 1` of the accumulate check costs 6e-8 absolute: gradients of 1e-7 drown in it - compared on that scale)
                assert float((got - ref).abs().max()) < 5e-7 + 1e-6 * float(sc_.max()), (case, M)


def test_dense_source_gradients_written_by_the_data_gemm_equal_the_copy_and_add_path():
    """autograd._Mlp.backward writes the gradient of a dense full-width source from the first Linear's data GEMM itself (residual dy or
    the running sum of the processor's e0 gradient chain added in the epilogue); MAGNET_DIRECT_SRC_GRAD=0 is the old path (dX0, column
    copies, torch adds).  Same training step, both ways: every parameter gradient equal to rounding, and the new path is the one taken"""
    from magnet_amd import autograd as AG
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    cb = {k: v.cuda() for k, v in batch.items()}
    grads, taken = {}, {True: 0, False: 0}
    orig = AG._dense_source_of_block

    def spy(*a):
        r = orig(*a)
        taken[AG.DIRECT_SRC_GRAD] += r is not None
        return r

    prev = AG.DIRECT_SRC_GRAD
    AG._dense_source_of_block = spy
    try:
        for direct in (True, False):
            AG.DIRECT_SRC_GRAD = direct
            model = FACTORY[name](hp)
            model.load_state_dict(oracle.state_dict())
            model = model.cuda().train()
            loss = model.training_step(cb, 0)
            loss.backward()
            grads[direct] = (float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters()})
    finally:
        AG.DIRECT_SRC_GRAD, AG._dense_source_of_block = prev, orig
    assert taken[True] > 0 and taken[False] == 0
    assert grads[True][0] == grads[False][0]
    worst = max(rel_l2(grads[True][1][n], grads[False][1][n]) for n in grads[True][1])
    assert worst < 2e-6, worst


@pytest.mark.parametrize("tag", ["magnet_gnn_tiny", "magnet_cnn_2d"])
def test_parameter_gradients_accumulated_by_the_kernel_equal_autograds_accumulation(tag, monkeypatch):
    """autograd.accumulate_param_grads (entered by the models' training rollouts): the one-kernel Linear backward adds its dW / db to
    p.grad itself instead of returning them for torch's AccumulateGrad.  Same training step with the switch off: the same loss and
    gradients to rounding; with it on, fewer gradients travel through autograd.  (_FUSED_LAYER_ROWS lowered: the golden cases are small.)"""
    from magnet_amd import autograd as AG
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    monkeypatch.setattr(AG, "_FUSED_LAYER_ROWS", 1)
    name, hp, salt, batch, d = load_case(tag)
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    cb = {k: v.cuda() for k, v in batch.items()}
    calls = {True: 0, False: 0}
    orig = AG.ops.linear_bwd128

    def spy(*a, **kw):
        calls[bool(kw.get("accumulate", False))] += 1
        return orig(*a, **kw)

    monkeypatch.setattr(AG.ops, "linear_bwd128", spy)
    grads = {}
    for on in (True, False):
        monkeypatch.setattr(AG, "ACCUM_PARAM_GRADS", on)
        calls[True] = calls[False] = 0
        model = FACTORY[name](hp)
        model.load_state_dict(oracle.state_dict())
        model = model.cuda().train()
        loss = model.training_step(cb, 0)
        loss.backward()
        assert all(p.grad is not None and p.grad.shape == p.shape for p in model.parameters())
        grads[on] = (float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters()}, dict(calls))
    assert grads[True][2][True] > 0 and grads[False][2][True] == 0, (grads[True][2], grads[False][2])
    assert grads[True][0] == grads[False][0]
    worst = max(rel_l2(grads[True][1][n], grads[False][1][n]) for n in grads[True][1])
    assert worst < 2e-6, worst


def test_a_frozen_hidden_linear_gets_no_gradient_from_the_accumulating_backward(monkeypatch):
    """requires_grad=False parameters are leaves too.  With autograd.accumulate_param_grads the one-kernel Linear
    backward must not assign / add to the .grad of a frozen Linear (torch.optim.Adam(self.parameters()) would then update it);
    every other parameter keeps the gradient it has without the freeze."""
    from magnet_amd import autograd as AG
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    monkeypatch.setattr(AG, "_FUSED_LAYER_ROWS", 1)
    name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    cb = {k: v.cuda() for k, v in batch.items()}
    grads = {}
    for frozen in (False, True):
        model = FACTORY[name](hp)
        model.load_state_dict(oracle.state_dict())
        model = model.cuda().train()
        hidden = model._processor.gnn_stacks[0].edge_fn[0].layers[2]          # a hidden Linear of a 5-Linear chain
        assert hidden.weight.shape[0] == hidden.weight.shape[1]
        if frozen:
            hidden.weight.requires_grad_(False)
            hidden.bias.requires_grad_(False)
        before = hidden.weight.detach().clone()
        loss = model.training_step(cb, 0)
        loss.backward()
        grads[frozen] = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
        if frozen:
            assert hidden.weight.grad is None and hidden.bias.grad is None
            opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-2)
            opt.step()
            assert torch.equal(hidden.weight.detach(), before)
    frozen_names = {n for n, g in grads[True].items() if g is None}
    assert len(frozen_names) == 2, frozen_names
    worst = max(rel_l2(grads[True][n], grads[False][n]) for n in grads[True] if n not in frozen_names)
    assert worst < 2e-6, worst


@pytest.mark.parametrize("unequal", [False, True])
def test_forward_is_differentiable_like_the_reference(unequal):
    """models/mpnn.py:175-203 is an ordinary differentiable torch function: a custom loop may call model(data, L, tmax, dt) in train()
    mode and backpropagate a loss of its own.  Here: parameter gradients and the gradient of data.x of ONE forward call against torch
    autograd through the oracle, on the golden 1-D graph and on a foreign Data of three graphs of UNEQUAL size (PyG's
    InstanceNorm(x, batch) takes any batch vector; training on such a batch raised NotImplementedError before round 5)."""
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.pyg_semantics import radius_graph
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case("mpnn_1d")
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp)
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    tw = hp.time_window
    g = torch.Generator().manual_seed(4)
    sizes = [37, 64, 21] if unequal else [40, 40]
    xs = [torch.linspace(0, 16, n) for n in sizes]
    pos_x = torch.cat(xs)
    bvec = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)])
    u = torch.cat([torch.sin(0.5 * x[:, None] + torch.linspace(0, 1, tw)[None, :] * 3 + i) for i, x in enumerate(xs)])
    u = u + 0.01 * torch.randn(u.shape, generator=g)
    pos = torch.cat([torch.full((pos_x.numel(), 1), 0.25), pos_x[:, None]], 1)
    ei = radius_graph(pos_x[:, None], r=3 * float(xs[0][1] - xs[0][0]) + 1e-4, batch=bvec, loop=False)
    L_, tmax, dt = torch.tensor(16.0), torch.tensor(4.0), torch.tensor(0.016)
    wgt = torch.randn(u.shape[0], tw, generator=g)                       # a loss that weighs every output differently
    # oracle: fp64 autograd
    o64 = O.build(name, hp).double()
    o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
    ux = u.double().requires_grad_(True)
    data = types.SimpleNamespace(x=ux, pos=pos.double(), batch=bvec, edge_index=ei)
    want = o64(data, L_.double(), tmax.double(), dt.double())
    (want * wgt.double()).sum().backward()
    want_g = {n: p.grad.float() for n, p in o64.named_parameters()}
    # HIP
    uc = u.cuda().requires_grad_(True)
    cdata = types.SimpleNamespace(x=uc, pos=pos.cuda(), batch=bvec.cuda(), edge_index=ei.cuda())
    got = model(cdata, L_.cuda(), tmax.cuda(), dt.cuda())
    assert got.grad_fn is not None and rel_l2(got.detach(), want.detach().float()) < 1e-5
    (got * wgt.cuda()).sum().backward()
    worst = max(rel_l2(p.grad, want_g[n]) for n, p in model.named_parameters())
    assert worst < 2e-4, worst
    assert rel_l2(uc.grad, ux.grad.float()) < 2e-4
    # eval mode: the inference path, no graph
    model.eval()
    assert model(cdata, L_.cuda(), tmax.cuda(), dt.cuda()).grad_fn is None


@pytest.mark.parametrize("B,Ns", [(1, 100), (3, 257), (4, 4096)])
def test_instance_norm_bwd_with_the_swish_backward_in_its_apply_pass_equals_the_two_calls(B, Ns):
    """magnet_instance_norm_bwd_swish_f32: dx bit-identical to instance_norm_bwd, dz = dx * Swish'(z) equal to swish_bwd of it up to the
    compiler's choice of fused multiply-adds (a few ulp), the block maxima of |dz| exact"""
    from magnet_amd import ops
    torch.manual_seed(7)
    N, H = B * Ns, 128
    dy, z = torch.randn(N, H, device="cuda") * 1e-7, torch.randn(N, H, device="cuda")
    x = torch.randn(N, H, device="cuda")
    yhat, var = ops.instance_norm(x, B, Ns, out=torch.empty_like(x), return_var=True)
    dx0 = ops.instance_norm_bwd(dy, yhat, var, B, Ns)
    am0 = torch.zeros(256, device="cuda")
    dz0 = ops.swish_bwd(dx0, z, torch.empty_like(z), amax=am0)
    am1 = torch.zeros(256, device="cuda")
    dx1, dz1 = ops.instance_norm_bwd_swish(dy, yhat, var, B, Ns, z, torch.empty_like(z), amax=am1)
    assert torch.equal(dx0, dx1) and rel_l2(dz1, dz0) < 2e-7 and float((dz1 - dz0).abs().max()) <= 4e-7 * float(dz0.abs().max())
    assert float(am1.max()) == float(dz1.abs().max())
    dx2, dz2 = ops.instance_norm_bwd_swish(dy, yhat, var, B, Ns, z, torch.empty_like(z))
    assert torch.equal(dx1, dx2) and torch.equal(dz1, dz2)                 # (with and without the maxima: the same pass)
    want = dx0.double() * (torch.sigmoid(z.double()) * (1 + z.double() * (1 - torch.sigmoid(z.double()))))
    assert rel_l2(dz1, want) < 3e-7
