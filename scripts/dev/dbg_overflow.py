import sys; sys.path.insert(0, ".")
import torch, types
from tests.helpers import load_case
from magnet_amd.models.factory import FACTORY
from magnet_amd import ops
from oracle.weightgen import fill_state_dict
name, hp, salt, batch, d = load_case("mpnn_1d")
model = FACTORY[name](hp).eval()
fill_state_dict(model, salt)
with torch.no_grad():
    for p in model.embedding_mlp.parameters():
        p.mul_(1000.0)
model = model.cuda()
cb = {k: v.cuda() for k, v in batch.items()}
u, x, t = model._unpack(cb)
g = model._build_graph(u[:, :25], t, x, [0, 0])
tape = types.SimpleNamespace()
out = model._forward(g, x[0, -1], t[0, -1], t[0][1] - t[0][0], tape)
print("node_in max", float(tape.node_in.abs().max()))
for i, l in enumerate(tape.layers):
    print(i, "h_in max", float(l.h_in.abs().max()), "A", float(l.A.abs().max()), "finite", bool(torch.isfinite(l.A).all()), "agg", float(l.agg.abs().max()), "h_out", float(l.h_out.abs().max()))
print("out finite", bool(torch.isfinite(out).all()), float(out.abs().max()), model._guard._status)
