import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, copy
from tests.helpers import load_case, rel_l2
from magnet_amd.models.factory import FACTORY
from magnet_amd import autograd as AG, ops
from oracle import models as O
from oracle.weightgen import fill_state_dict
name, hp, salt, batch, d = load_case("magnet_gnn_tiny")
oracle = O.build(name, hp); fill_state_dict(oracle, salt)
model = FACTORY[name](hp); model.load_state_dict(oracle.state_dict()); model = model.cuda().train()
pk = model._packed()
torch.manual_seed(0)
for scale_in in (1.0, 30.0):
    x = (torch.randn(396, 32, device="cuda") * scale_in).requires_grad_()
    y = AG.mlp(396, [ops.Term(x, 32)], pk.decoder, pk.decoder.params)
    gy = torch.randn_like(y) * 1e-5
    for p in pk.decoder.params: p.grad = None
    y.backward(gy)
    net = copy.deepcopy(model._decoder.node_fn).double()
    x64 = x.detach().double().requires_grad_()
    h = x64
    lins = net.linears()
    for i, lin in enumerate(lins):
        h = lin(h)
        if i < 4: h = torch.relu(h)
    print("scale", scale_in, "fwd", rel_l2(y, h.detach()), "max|h|", float(h.abs().max()))
    h.backward(gy.double())
    print("  dx", rel_l2(x.grad, x64.grad))
    for i, lin in enumerate(lins):
        print("  layer", i, rel_l2(pk.decoder.params[2 * i].grad, lin.weight.grad), rel_l2(pk.decoder.params[2 * i + 1].grad, lin.bias.grad))
