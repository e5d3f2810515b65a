"""Whole-model, mask-synchronised gradient parity of the three MAgNet models.

The MAgNet nets are piecewise linear (3e7 ReLUs per training step of the fixtures, an L1 loss): an fp32 forward that differs from the
reference's in the seventh digit takes the other branch of a few units, the gradient - discontinuous there - moves by percents, and a
comparison of whole-model gradients can only be loose (tests/test_train_gpu.py: 1e-2).  Here the comparison is made WELL POSED instead:
the HIP run's own branch decisions - the ReLU mask of every hidden layer of every chain call (from the activations its backward
recomputes), of every EDSR convolution, and the sign tensor of every L1 term - are recorded and FORCED into an fp64 autograd run of the
oracle (``h * mask`` instead of ``relu(h)``, ``(x - y) * sign`` instead of ``|x - y|``), whose graphs get their edges in the HIP run's
CSR order so that the rows of every chain line up.  Both sides then differentiate the SAME smooth function at the same point, and every
parameter gradient of the whole ``training_step`` must agree to rounding (<= 1e-4; measured ~1e-6): a bug that moves any gradient by a
fraction of a percent fails, which the kink-limited comparison could not see.  Reference: models/magnet_gnn.py:388-440,
models/magnet_cnn.py:393-437, models/magnet_cnn_2d.py:399-443, models/backbones/mlp.py:9-28, models/backbones/edsr.py:32-60.
"""
import collections

import pytest
import torch

from tests.helpers import load_case, oracle_magnet_cnn_grads, oracle_magnet_gnn_grads, rel_l2

pytestmark = pytest.mark.gpu


def _kts_of(groups):
    """the k tiles of an autograd.mlp call in the form its backward hands to ops.mlp (autograd._Mlp.backward)"""
    from magnet_amd import ops
    out = []
    for g in groups:
        terms = g if isinstance(g, (tuple, list)) else (g,)
        for k in range(terms[0].width // 32):
            out.append([(ops.Term(ops._f32(t.t, "source"), 32, idx=t.idx, scale=t.scale, rdiv=t.rdiv), t.col + 32 * k) for t in terms])
    return out


def _hip_run(model, cb, break_backward=None):
    """one training_step + backward on the HIP model; returns (loss, recorded branch decisions)"""
    from magnet_amd import autograd as AG
    import torch.nn.functional as F
    pname = {id(p): n for n, p in model.named_parameters()}
    rec = types_ns(masks=collections.defaultdict(list), outs=collections.defaultdict(list), graphs=[], signs=[], edsr=[])
    orig_mlp, orig_conv, orig_l1, orig_net = AG.mlp, AG.conv, F.l1_loss, type(model)._graph_net_grad
    orig_bwd = AG._Mlp.backward

    def spy_mlp(M, groups, pk, params, **kw):
        y = orig_mlp(M, groups, pk, params, **kw)
        nl = len(pk.layers)
        if nl > 1 and M > 0:
            assert not kw.get("final_relu")
            kts = _kts_of(groups)
            K0 = 32 * len(kts)
            dev = y.device
            X = [torch.empty(M, K0, device=dev)] + [torch.empty(M, pk.ht * 32, device=dev) for _ in range(nl - 1)]
            zL = torch.empty(M, pk.ot * 32, device=dev) if pk.ln is not None else None
            with torch.no_grad():
                AG._recompute_tape(M, kts, pk, nl, X[0], X, zL, K0)      # the hidden activations the backward itself masks with
            path = pname[id(params[0])].split(".layers.")[0]
            hid = params[0].shape[0]
            rec.masks[path].append([(X[l + 1][:, :hid] > 0).cpu() for l in range(nl - 1)])
            rec.outs[path].append(y.detach().double().cpu())
        return y

    def spy_conv(src, taps, taps_rev, pk, W2, bias, relu=False, res=None):
        y = orig_conv(src, taps, taps_rev, pk, W2, bias, relu=relu, res=res)
        if relu:
            rec.edsr.append((y.detach() > 0).cpu())
        return y

    def spy_l1(input, target, *a, **kw):
        rec.signs.append(torch.sign(input.detach() - target.detach()).double().cpu())
        return orig_l1(input, target, *a, **kw)

    def spy_net(self, pk, nf, topo, D):
        rec.graphs.append((topo.csr.tgt.long().cpu(), topo.csr.src.long().cpu()))
        return orig_net(self, pk, nf, topo, D)

    def broken_bwd(ctx, dy):                                              # a deliberately wrong backward: one LayerNorm gradient off by 2 %
        out = list(orig_bwd(ctx, dy))
        for i in range(len(out) - 1, -1, -1):
            if torch.is_tensor(out[i]) and out[i].dim() == 1 and out[i].numel() == break_backward:
                out[i] = out[i] * 1.02
                break
        return tuple(out)

    AG.mlp, AG.conv, F.l1_loss = spy_mlp, spy_conv, spy_l1
    type(model)._graph_net_grad = spy_net
    if break_backward:
        AG._Mlp.backward = staticmethod(broken_bwd)
    try:
        model.zero_grad()
        loss = model.training_step(cb, 0)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        AG.mlp, AG.conv, F.l1_loss = orig_mlp, orig_conv, orig_l1
        type(model)._graph_net_grad = orig_net
        AG._Mlp.backward = orig_bwd
    return float(loss.detach()), rec


class types_ns:                                                           # (a SimpleNamespace)
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _oracle_run(oracle, batch, hp, rec, kind):
    """fp64 autograd through the oracle with the recorded branch decisions forced; returns (loss, {name: grad}, worst chain-output error)"""
    import torch.nn.functional as F
    from oracle import models as O
    from oracle import pyg_semantics as S
    masks = {k: collections.deque(v) for k, v in rec.masks.items()}
    outs = {k: collections.deque(v) for k, v in rec.outs.items()}
    graphs, signs, edsr = collections.deque(rec.graphs), collections.deque(rec.signs), collections.deque(rec.edsr)
    state = types_ns(cur=None, layer=0, call=None, out=None, worst=0.0, flips=0, units=0)
    pending = collections.defaultdict(collections.deque)
    orig_relu, orig_l1, orig_radius = torch.relu, F.l1_loss, S.radius_graph

    def relu(x):
        if state.cur == "edsr":
            mk = edsr.popleft()                                           # [P, C] channels-last rows of the HIP run -> [B, C, *spatial]
            B = x.shape[0]
            mk = mk.view(B, *x.shape[2:], x.shape[1]).movedim(-1, 1)
        else:
            mk = state.call[state.layer]
            state.layer += 1
            if mk.shape != x.shape and mk.numel() == x.numel():           # rows (query, time step) of the HIP chain = the leading dimensions
                mk = mk.view(x.shape)                                      # of the oracle's tensor (the chain-output check below confirms it)
        assert mk.shape == x.shape, (state.cur, mk.shape, x.shape)
        state.flips += int(((x.detach() > 0) != mk).sum())
        state.units += mk.numel()
        return x * mk.to(x.dtype)

    def l1(input, target, *a, **kw):
        s = signs.popleft()
        assert s.shape == input.shape
        return ((input - target) * s).mean()

    def radius_graph(x, r, batch=None, loop=False, **kw):
        want = orig_radius(x, r, batch, loop=loop, **kw)
        tgt, src = graphs.popleft()
        got = torch.stack([tgt, src])
        # the same edge SET as the restated torch_cluster call (the HIP run's CSR order instead of query-major order)
        key = lambda e: (e[0] * (x.shape[0] + 1) + e[1]).sort().values  # noqa: E731
        assert got.shape == want.shape and torch.equal(key(got), key(want))
        return got

    hooks = []
    for path, mod in oracle.named_modules():
        if isinstance(mod, O.MLP):
            def pre(m, inp, path=path):
                state.cur, state.layer = path, 0
                if path not in masks:
                    state.call = None
                    return
                x = inp[0]
                rows = x.numel() // x.shape[-1]
                if not pending[path]:
                    call, out = masks[path].popleft(), outs[path].popleft()
                    M = call[0].shape[0]
                    if M == rows:
                        pending[path].append((call, out))
                    else:
                        # MAgNet[CNN]'s grid decoder: the HIP chain runs the 2^d neighbour cells of a query in ONE call, rows
                        # ((b, q), cell s, time step i); the reference calls proj_head once per cell on [(b, q), i] (magnet_cnn_2d.py:329-390)
                        assert x.dim() == 3 and M % rows == 0, (path, M, x.shape)
                        S_, T_ = M // rows, x.shape[1]
                        for s_ in range(S_):
                            pending[path].append(([mk.view(rows // T_, S_, T_, -1)[:, s_].reshape(rows, -1) for mk in call],
                                                  out.view(rows // T_, S_, T_, -1)[:, s_].reshape(rows, -1)))
                state.call, state.out = pending[path].popleft()
            def post(m, inp, out, path=path):
                if path in masks and not any(isinstance(c, torch.nn.LayerNorm) for c in m_parent(oracle, path).children()):
                    state.worst = max(state.worst, rel_l2(out.reshape(-1), state.out.reshape(-1)))
                state.cur = None
            hooks += [mod.register_forward_pre_hook(pre), mod.register_forward_hook(post)]
        elif isinstance(mod, O.EDSR):
            hooks += [mod.register_forward_pre_hook(lambda m, i: setattr(state, "cur", "edsr")),
                      mod.register_forward_hook(lambda m, i, o: setattr(state, "cur", None))]
    orig_gs = F.grid_sample
    F.grid_sample = lambda inp, grid, **kw: orig_gs(inp, grid.to(inp.dtype), **kw)      # (the fp64 run: make_coord's grids are fp32)
    torch.relu, F.l1_loss, S.radius_graph = relu, l1, radius_graph
    try:
        if kind == "gnn":
            loss, grads = oracle_magnet_gnn_grads(oracle, batch, hp.teacher_forcing, hp.loss)
        else:
            loss, grads = oracle_magnet_cnn_grads(oracle, batch, hp.teacher_forcing, hp.loss)
    finally:
        torch.relu, F.l1_loss, S.radius_graph = orig_relu, orig_l1, orig_radius
        F.grid_sample = orig_gs
        for h in hooks:
            h.remove()
    assert not any(len(q) for q in masks.values()) and not edsr and not graphs, "the oracle made fewer calls than the HIP run recorded"
    return loss, grads, state


def m_parent(root, path):
    parent = path.rsplit(".", 1)[0] if "." in path else ""
    return root.get_submodule(parent) if parent else root


def _pair(tag, teacher_forcing):
    from magnet_amd.models.factory import FACTORY
    from oracle import models as O
    from oracle.weightgen import fill_state_dict
    name, hp, salt, batch, d = load_case(tag)
    hp.teacher_forcing = teacher_forcing
    oracle = O.build(name, hp)
    fill_state_dict(oracle, salt)
    model = FACTORY[name](hp)
    model.load_state_dict(oracle.state_dict())
    return name, hp, batch, oracle, model.cuda().train()


@pytest.mark.parametrize("tag,teacher_forcing", [("magnet_gnn_tiny", False), ("magnet_gnn_tiny", True), ("magnet_cnn_1d", False),
                                                 ("magnet_cnn_1d", True), ("magnet_cnn_2d", False), ("magnet_cnn_2d", True),
                                                 # the shipped hyper-parameters (128-wide chains, 5 / 10 hops, 16 residual blocks): the backward kernels
                                                 # of the production shapes - one-kernel Linear backward, first Linear at source rows, x0-free tape
                                                 ("magnet_gnn_full", False), ("magnet_gnn_c5_reduced", False), ("magnet_cnn_1d_shipped", True),
                                                 ("magnet_cnn_2d_shipped", False)])
def test_every_parameter_gradient_against_mask_synchronised_fp64_autograd(tag, teacher_forcing):
    name, hp, batch, oracle, model = _pair(tag, teacher_forcing)
    cb = {k: v.cuda() for k, v in batch.items()}
    loss, rec = _hip_run(model, cb)
    assert sum(len(v) for v in rec.masks.values()) > 10 and len(rec.graphs) >= 2
    want_loss, want, st = _oracle_run(oracle, batch, hp, rec, "gnn" if name == "magnet_gnn" else "cnn")
    # the fp64 forward on the HIP run's branches reproduces the HIP loss, and only a handful of the units sat on the other side
    assert abs(loss - want_loss) < 2e-5 * max(1.0, abs(want_loss)), (loss, want_loss)
    assert st.units > 1e6 and st.flips < 1e-4 * st.units, (st.flips, st.units)
    assert st.worst < 1e-5, st.worst                       # the rows of the un-normalised chains (projector, decoder) line up with the oracle's
    errs = {n: rel_l2(p.grad, want[n]) for n, p in model.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"[{tag} tf={teacher_forcing}] {len(errs)} parameters, worst {worst} {errs[worst]:.2e}; {st.flips} of {st.units} units forced "
          f"onto the HIP run's branch")
    assert errs[worst] < 1e-4, {n: e for n, e in errs.items() if e >= 1e-4}


@pytest.mark.parametrize("tag", ["magnet_gnn_tiny", "magnet_cnn_2d"])
def test_mask_synchronised_comparison_notices_a_two_percent_error_in_one_gradient(tag):
    """the net has to catch what the kink-limited comparison lets through: one LayerNorm weight gradient scaled by 1.02"""
    name, hp, batch, oracle, model = _pair(tag, False)
    cb = {k: v.cuda() for k, v in batch.items()}
    loss, rec = _hip_run(model, cb, break_backward=hp.latent_dim)
    _, want, _ = _oracle_run(oracle, batch, hp, rec, "gnn" if name == "magnet_gnn" else "cnn")
    errs = {n: rel_l2(p.grad, want[n]) for n, p in model.named_parameters()}
    bad = {n: e for n, e in errs.items() if e >= 1e-4}
    assert bad and max(bad.values()) > 5e-3, errs
