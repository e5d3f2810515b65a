"""GPU parity of the MPNN training step: the HIP backward kernels vs torch autograd.

Op level: each building block against the same op differentiated by torch (fp32 on the GPU or fp64 on the CPU).
Model level: loss and every parameter gradient of ``training_step`` against torch autograd through the oracle
restatement (whose forward is pinned to the reference's golden vectors; the reference itself obtains these
gradients from torch autograd through the same operations, models/mpnn.py:254-297).
"""
import types

import pytest
import torch

from tests.helpers import load_case, rel_l2

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-4      # relative L2 per parameter gradient (fp32 accumulation order differs; values ~1e-6)


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
    from magnet_amd import ops
    from magnet_amd.models.mpnn import _READOUT
    torch.manual_seed(4)
    k1, s1, k2 = _READOUT[tw]
    use_swish = not (dim == 1 and tw == 10)
    n = 1037
    h = torch.randn(n, 128, device="cuda")
    dout = torch.randn(n, tw, device="cuda")
    c1 = torch.nn.Conv1d(1, 8, k1, stride=s1).cuda()
    c2 = torch.nn.Conv1d(8, 1, k2).cuda()
    dt = torch.tensor([0.37], device="cuda")
    u_last = torch.randn(n, device="cuda", requires_grad=True)
    hh = h.clone().requires_grad_()
    z = c1(hh[:, None])
    if use_swish:
        z = z * torch.sigmoid(z)
    out = u_last[:, None] + dt * torch.arange(1, tw + 1, device="cuda") * c2(z).squeeze(1)
    out.backward(dout)
    dh, du, gp = ops.mpnn_readout_bwd(h, dout, tw, c1.weight.detach().reshape(-1).contiguous(), c1.bias.detach(), k1, s1,
                                      c2.weight.detach().reshape(-1).contiguous(), k2, use_swish, dt)
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


def _oracle_grads(oracle, batch, teacher_forcing, loss_name):
    """loss + gradients by torch autograd through the oracle's rollout (fp64 on the CPU for a clean target)."""
    from oracle.models import _criterion
    o = oracle.double()
    f32 = {k: v.float() for k, v in batch.items()}
    u = f32["u"].permute(0, 2, 1)
    x = f32["x"] if o.dim == 2 else f32["x"].squeeze(-1)
    t = f32["t"]
    B, _, N = u.shape
    tw = o.tw
    dt = (t[0][1] - t[0][0]).double()

    def graph_of(window, step):
        # topology / positions from the fp32 inputs (what the reference sees), features in fp64 with their graph
        g = o._build_graph(window.detach().float(), t, x, [step] * B)
        g.x = window.double().permute(0, 2, 1).reshape(B * N, tw)
        g.pos = g.pos.double()
        return g

    graph = graph_of(u[:, :tw], o._step_index(0))
    target = u[:, tw:].double()
    outs = []
    for i in range(target.shape[1] // tw):
        y = o.forward(graph, x[0, -1].double(), t[0, -1].double(), dt).reshape(B, N, -1).permute(0, 2, 1)
        outs.append(y)
        nxt = u[:, (i + 1) * tw:(i + 2) * tw] if teacher_forcing else y
        graph = graph_of(nxt, o._step_index(i + 1))
    loss = _criterion(loss_name)(torch.cat(outs, 1), target)
    o.zero_grad()
    loss.backward()
    return float(loss.detach()), {n: p.grad.clone() for n, p in o.named_parameters()}


@pytest.mark.parametrize("teacher_forcing", [False, True])
@pytest.mark.parametrize("tag", ["mpnn_1d", "mpnn_2d"])
def test_training_step_gradients_match_autograd(tag, teacher_forcing):
    hp, batch, oracle, model = _pair(tag, teacher_forcing=teacher_forcing)
    want_loss, want = _oracle_grads(oracle, batch, teacher_forcing, hp.loss)
    cb = {k: v.cuda() for k, v in batch.items()}
    model.train()
    model.zero_grad()
    loss = model.training_step(cb, 0)
    assert loss.requires_grad
    loss.backward()
    assert abs(float(loss) - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    worst = {}
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        worst[n] = rel_l2(p.grad, want[n])
    bad = {n: e for n, e in worst.items() if not e < GRAD_TOL}
    assert not bad, bad


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
    l0 = float(model.training_step(cb, 0))
    for _ in range(5):
        opt.zero_grad()
        loss = model.training_step(cb, 0)
        loss.backward()
        opt.step()
    assert float(model.training_step(cb, 0)) < l0          # a few Adam steps on one batch reduce its loss
