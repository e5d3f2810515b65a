"""Dev: repeats the (5 Linears, hidden 64, out 64, K 40, LayerNorm, 1 013 rows) launch of the chain kernel and reports the 32-row
wave tiles whose result deviates - the reproducer of the intermittent failure of the biases-from-LDS experiments
(profiles/r2/chain_sq_counters.txt).  With the shipped kernel it prints nothing."""
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
M = 1000 + 13
kp = pad_to(k_in)
x = torch.zeros(M, kp); x[:, :k_in] = torch.randn(M, k_in)
seq = torch.nn.Sequential(m, norm).cuda()
pk = pack_mlp_ln(seq, k_pad=kp, mode=mode)
with torch.no_grad():
    r_ = x[:, :k_in].double()
    for layer in m.double().cpu().layers:
        r_ = layer(r_)
    pre = r_
    ref = norm.double().cpu()(r_)
print("pre-LN row std: min", float(pre.std(1).min()), "median", float(pre.std(1).median()), " |mean|/std max", float((pre.mean(1).abs() / pre.std(1)).max()))
for rep in range(200):
    y = torch.full((M, out), float("nan"), device="cuda")
    run_mlp(M, ops.ktiles(ops.Term(x.cuda(), kp)), pk, y)
    err = (y.double().cpu() - ref)
    rows = err.norm(dim=1) / ref.norm(dim=1)
    bad = (rows > 1e-5).nonzero().flatten().tolist()
    if bad:
        e = err[bad]
        print("   col-block err norms (pre-LN not available): ", [f"{float(e[:, 8*k:8*k+8].norm()):.2e}" for k in range(8)], " first bad row err[:8]", [f"{v:.1e}" for v in err[bad[0]][:8].tolist()])
    if bad: print(rep, "bad rows", bad[0], "..", bad[-1], len(bad), "tiles", sorted(set(b // 32 for b in bad)), f"{float(rows.max()):.2e}")
    continue
    print("rel_l2", float(err.norm() / ref.norm()), "worst rows", rows.topk(5).indices.tolist(), [f"{v:.2e}" for v in rows.topk(5).values.tolist()],
          "rows > 1e-5:", int((rows > 1e-5).sum()), "median row err", float(rows.median()))
    cols = err.norm(dim=0) / ref.norm(dim=0)
    print("   worst cols", cols.topk(5).indices.tolist(), [f"{v:.2e}" for v in cols.topk(5).values.tolist()])
