import sys, torch
sys.path.insert(0, "/root/repo")
from magnet_amd import ops
from magnet_amd.models.backbones.mlp import MLP
from magnet_amd.models.magnet_common import pack_mlp_ln, run_mlp
from magnet_amd.packing import pad_to
hidden, out, k_in, mode = 64, 64, 40, "fp16x3"
torch.manual_seed(0)
m = MLP(k_in, [hidden] * 4, out)
norm = torch.nn.LayerNorm(out)
with torch.no_grad():
    norm.weight.uniform_(0.5, 1.5), norm.bias.uniform_(-0.5, 0.5)
M = 1013
kp = pad_to(k_in)
x = torch.zeros(M, kp); x[:, :k_in] = torch.randn(M, k_in)
seq = torch.nn.Sequential(m, norm).cuda()
pk = pack_mlp_ln(seq, k_pad=kp, mode=mode)
acts = []
with torch.no_grad():
    r_ = x[:, :k_in].double()
    layers = list(m.double().cpu().layers)
    for layer in layers:
        r_ = layer(r_)
        if isinstance(layer, torch.nn.ReLU) or layer is layers[-1]:
            acts.append(r_.clone())
print("n reference activations", len(acts), [tuple(a.shape) for a in acts])
xc = x.cuda()
for rep in range(100):
    y = torch.full((M, out), float("nan"), device="cuda")
    hid = [torch.full((M, 64), float("nan"), device="cuda") for _ in range(5)]
    run_mlp(M, ops.ktiles(ops.Term(xc, kp)), pk, y, hid=hid)
    errs = [((h.double().cpu() - a).norm(dim=1) / a.norm(dim=1).clamp_min(1e-30)) for h, a in zip(hid, acts)]
    bad = [(l, (e > 3e-6).nonzero().flatten().tolist()) for l, e in enumerate(errs)]
    if any(b for _, b in bad):
        print(rep, [(l, sorted(set(r // 32 for r in b)), f"{float(errs[l].max()):.1e}") for l, b in bad if b])
