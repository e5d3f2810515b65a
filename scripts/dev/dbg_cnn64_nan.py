"""Dev: find the first kernel of the cnn_64 training backward whose outputs are non-finite although its inputs are finite."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from magnet_amd import autograd as AG, ops
from magnet_amd.models.factory import FACTORY
name, hp, mk = bench.ref_shape_cases()["cnn_64"]
torch.manual_seed(0)
model = FACTORY[name](types.SimpleNamespace(**hp)).cuda().train()
opt = model.configure_optimizers()["optimizer"]
batch, _ = mk(True)
batch = {k: v.cuda() for k, v in batch.items()}
fin = lambda t: t is None or bool(torch.isfinite(t).all())
state = {"on": False, "hit": False}
orig_lb = ops.linear_bwd128


def lb(dz, x, wt, dx, dW, db, ws, mask=True, amax_in=None, want_amax=True, accumulate=False):
    r = orig_lb(dz, x, wt, dx, dW, db, ws, mask=mask, amax_in=amax_in, want_amax=want_amax, accumulate=accumulate)
    if state["on"] and not state["hit"]:
        ok_in = fin(dz) and fin(x[:, :dz.shape[1]])
        ok_out = fin(dx) and fin(dW) and fin(db)
        if not ok_out:
            state["hit"] = True
            am = amax_in[0][:amax_in[1]] if amax_in else None
            print("linear_bwd: M", dz.shape, "inputs finite", ok_in, "dx/dW/db finite", fin(dx), fin(dW), fin(db),
                  "max|dz|", float(dz.abs().max()), "max|x|", float(x.abs().max()),
                  "amax_in max/min/finite", (float(am.max()), float(am.min()), fin(am)) if am is not None else None, flush=True)
    return r


ops.linear_bwd128 = lb
orig_b = AG._Mlp.backward


def wrap(ctx, dy):
    if state["on"] and ctx.n_tape and not state.get("tape"):
        tape = ctx.saved_tensors[-ctx.n_tape:]
        for k, t in enumerate(tape):
            nf = ~torch.isfinite(t)
            if bool(nf.any()):
                state["tape"] = True
                idx = nf.nonzero()
                rows = idx[:, 0].unique()
                print("TAPE", k, "of", len(tape), "shape", tuple(t.shape), "non-finite entries", int(nf.sum()), "rows", rows[:8].tolist(), "...", int(rows.numel()),
                      "cols", idx[:, 1].unique()[:16].tolist(), "values", t[nf][:4].tolist(), "M", ctx.meta.M, "layers", len(ctx.meta.pk.layers),
                      "max finite", float(t[~nf].abs().max()), flush=True)
                r0 = int(rows[0])
                for kk, tt in enumerate(tape):
                    print("   tape", kk, "row", r0, "max|.|", float(tt[r0].abs().max()), "first vals", [round(v, 3) for v in tt[r0, :6].tolist()], flush=True)
    out = orig_b(ctx, dy)
    if state["on"]:
        m = ctx.meta
        okin = fin(dy)
        okout = all(fin(g) for g in out if torch.is_tensor(g))
        if not okout and not state.get("rep"):
            state["rep"] = True
            print("_Mlp.backward: M", m.M, "layers", len(m.pk.layers), "out_dim", m.pk.out_dim, "epi", m.epi, "dy finite", okin, "max|dy|",
                  float(dy.abs().max()), "grads finite", [fin(g) for g in out if torch.is_tensor(g)], flush=True)
    return out


AG._Mlp.backward = staticmethod(wrap)
orig_mlp = AG.mlp
cap = {}


def mlp_spy(M, groups, pk, params, **kw):
    y = orig_mlp(M, groups, pk, params, **kw)
    if state["on"] and M == 327680 and len(pk.layers) == 5 and "x0" not in cap:
        cols = []
        for g in groups:
            terms = g if isinstance(g, (tuple, list)) else (g,)
            acc = 0
            for tm in terms:
                src = tm.t.detach().reshape(-1, tm.t.shape[-1])
                rows_ = src[tm.idx.long()] if tm.idx is not None else src
                acc = acc + rows_[:, tm.col:tm.col + tm.width] * tm.scale
            cols.append(acc)
        x0 = torch.cat(cols, 1)
        W0, b0 = params[0].detach(), params[1].detach()
        cm = getattr(pk, "col_map", None)
        print("x0", tuple(x0.shape), "max|x0|", float(x0.abs().max()), "finite", fin(x0), "W0", tuple(W0.shape), "col_map", None if cm is None else (len(cm), list(cm[:4]), list(cm[-4:])), flush=True)
        xin = x0 if cm is None else x0[:, torch.as_tensor(cm, device=x0.device).long()]
        h1 = torch.relu(xin[:, :W0.shape[1]] @ W0.t() + b0)
        cap["x0"], cap["h1"] = x0, h1
        print("reference hidden 1: max", float(h1.abs().max()), "row 13000 first", [round(v, 3) for v in h1[13000, :6].tolist()], flush=True)
        big = (x0.abs() > 100).nonzero()
        print("entries of x0 above 100:", int(big.shape[0]), big[:5].tolist(), flush=True)
    return y


AG.mlp = mlp_spy
for i in range(9):
    opt.zero_grad(set_to_none=True)
    state["on"] = i >= 6
    loss = model.training_step(batch, 0)
    loss.backward()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not fin(p.grad)]
    print(i, round(float(loss.detach()), 4), bad[:4], flush=True)
    if bad:
        break
    opt.step()
