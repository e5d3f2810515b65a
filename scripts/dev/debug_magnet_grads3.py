import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, copy
from tests.helpers import load_case, rel_l2
from magnet_amd.models.factory import FACTORY
from magnet_amd import autograd as AG, ops
from oracle import models as O
from oracle.weightgen import fill_state_dict
name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
hp.teacher_forcing = True
oracle = O.build(name, hp); fill_state_dict(oracle, salt)
model = FACTORY[name](hp); model.load_state_dict(oracle.state_dict()); model = model.cuda().train()
cb = {k: v.cuda() for k, v in batch.items()}
# capture the inputs / output-gradients of every decoder application
calls = []
orig = AG.mlp
def spy(M, groups, pk, params, **kw):
    y = orig(M, groups, pk, params, **kw)
    if pk is model._packed().decoder:
        rec = {"x": groups[0].t.detach().clone()}
        y.register_hook(lambda g, rec=rec: rec.__setitem__("dy", g.detach().clone()))
        calls.append(rec)
    return y
AG.mlp = spy
loss = model.training_step(cb, 0); loss.backward()
AG.mlp = orig
got = [p.grad.clone() for p in model._packed().decoder.params]
net = copy.deepcopy(model._decoder.node_fn).double()
for rec in calls:
    h = rec["x"].double()
    lins = net.linears()
    for i, lin in enumerate(lins):
        h = lin(h)
        if i < 4: h = torch.relu(h)
    h.backward(rec["dy"].double())
for i, lin in enumerate(net.linears()):
    print("layer", i, rel_l2(got[2 * i], lin.weight.grad), rel_l2(got[2 * i + 1], lin.bias.grad))
print(len(calls), [float(r["dy"].abs().max()) for r in calls])
# ---- the oracle's own decoder inputs / output gradients
from tests.helpers import oracle_magnet_gnn_grads
ocalls = []
def fhook(mod, inp, out):
    rec = {"x": inp[0].detach().clone()}
    out.register_hook(lambda g, rec=rec: rec.__setitem__("dy", g.detach().clone()))
    ocalls.append(rec)
h_ = oracle._decoder.node_fn.register_forward_hook(fhook)
wl, want = oracle_magnet_gnn_grads(oracle, batch, True, hp.loss)
h_.remove()
for i, (a, b) in enumerate(zip(calls, ocalls)):
    print("call", i, "x", rel_l2(a["x"], b["x"].reshape(a["x"].shape)), "dy", rel_l2(a["dy"], b["dy"].reshape(a["dy"].shape)),
          "n sign diff", int((torch.sign(a["dy"].cpu().double()) != torch.sign(b["dy"].reshape(a["dy"].shape))).sum()), "of", a["dy"].numel())
names = [n for n, _ in model.named_parameters() if n.startswith("_decoder")]
for n in names:
    p = dict(model.named_parameters())[n]
    print(n, "mine vs oracle", rel_l2(p.grad, want[n]))
# oracle decoder grads recomputed from its own captured (x, dy)
net2 = copy.deepcopy(oracle._decoder.node_fn).double()
for p in net2.parameters(): p.grad = None
for rec in ocalls:
    net2(rec["x"]).backward(rec["dy"])
for (n, p) in net2.named_parameters():
    print("oracle self-consistency", n, rel_l2(p.grad, want["_decoder.node_fn." + n]))
print("---- cross checks")
net3 = copy.deepcopy(oracle._decoder.node_fn).double()
for rec in calls:
    net3(rec["x"].cpu().double().reshape(ocalls[0]["x"].shape)).backward(rec["dy"].cpu().double().reshape(ocalls[0]["dy"].shape))
for (n, p) in net3.named_parameters():
    print("oracle decoder on MY (x, dy)", n, rel_l2(p.grad, want["_decoder.node_fn." + n]))
for (n, p), (n2, p2) in zip(net.named_parameters(), net2.named_parameters()):
    print("weights equal", n, float((p.detach().cpu() - p2.detach()).abs().max()))
x_m, x_o = calls[0]["x"].cpu().double(), ocalls[0]["x"].reshape(calls[0]["x"].shape)
print("max row diff x", float((x_m - x_o).abs().max()), "max |x|", float(x_o.abs().max()))
dy_m, dy_o = calls[0]["dy"].cpu().double(), ocalls[0]["dy"].reshape(calls[0]["dy"].shape)
print("max diff dy", float((dy_m - dy_o).abs().max()), "max |dy|", float(dy_o.abs().max()))
print("---- near-zero pre-activations (oracle decoder, oracle x)")
h = x_o
lins2 = net2.linears() if hasattr(net2, "linears") else [m for m in net2.layers if isinstance(m, torch.nn.Linear)]
for i, lin in enumerate(lins2[:4]):
    z = lin(h)
    print("layer", i, "units", z.numel(), "|z|<1e-5:", int((z.abs() < 1e-5).sum()), "|z|<1e-3:", int((z.abs() < 1e-3).sum()), "frac positive", float((z > 0).double().mean()))
    h = torch.relu(z)
print("x rows with |x| row-norm < 1e-6:", int((x_o.norm(dim=1) < 1e-6).sum()), "of", x_o.shape[0])
print("---- per-row contribution differences (layer 0 weight grad)")
def per_row(netx, x, dy):
    out = []
    lins_ = [m for m in netx.layers if isinstance(m, torch.nn.Linear)]
    xs = x.clone().requires_grad_()
    h = xs
    acts = []
    for i, lin in enumerate(lins_):
        z = lin(h); acts.append(z)
        h = torch.relu(z) if i < 4 else z
    (h * dy).sum().backward()
    return xs.grad, acts
gx_o, acts_o = per_row(net2, x_o, dy_o)
gx_m, acts_m = per_row(net2, x_m, dy_m)
rowdiff = (gx_o - gx_m).norm(dim=1) / gx_o.norm(dim=1).clamp_min(1e-30)
print("rows with input-grad rel diff > 1e-3:", int((rowdiff > 1e-3).sum()), "max", float(rowdiff.max()))
bad = torch.nonzero(rowdiff > 1e-3).flatten()[:5]
for r in bad.tolist():
    flips = [int(((a_o[r] > 0) != (a_m[r] > 0)).sum()) for a_o, a_m in zip(acts_o[:4], acts_m[:4])]
    minabs = [float(a_o[r].abs().min()) for a_o in acts_o[:4]]
    print("row", r, "rel diff", float(rowdiff[r]), "mask flips per layer", flips, "min |z|", minabs, "|x diff|", float((x_o[r] - x_m[r]).abs().max()))
