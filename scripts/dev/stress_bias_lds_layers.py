"""Second stage of scripts/dev/stress_bias_lds.py: on the failing case (5 Linears, hidden 64, out 64, k_in 40, LayerNorm, two-accumulator
fp16x3) the launch also writes its tape (every layer's activated output, the last Linear's raw output); for launches that differ from the
fp64 evaluation: the FIRST layer that is wrong, the column block, and what the wrong values look like relative to that layer's bias."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MAGNET_CHAIN_GEMM"] = "fp16x3"
import torch
from magnet_amd import ops
from magnet_amd.models.backbones.mlp import MLP
from magnet_amd.models.magnet_common import pack_mlp_ln, run_mlp
from magnet_amd.packing import pad_to

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
hidden, out, k_in, M = 64, 64, 40, 1013
torch.manual_seed(0)
m, norm = MLP(k_in, [hidden] * 4, out), torch.nn.LayerNorm(out)
kp = pad_to(k_in)
x = torch.zeros(M, kp)
x[:, :k_in] = torch.randn(M, k_in)
lins = [l for l in m.layers if isinstance(l, torch.nn.Linear)]
refs = []
with torch.no_grad():
    h = x[:, :k_in].double()
    for i, lin in enumerate(lins):
        h = h @ lin.weight.double().t() + lin.bias.double()
        if i < 4:
            h = torch.relu(h)
        refs.append(h.cuda())
seq = torch.nn.Sequential(m, norm).cuda()
pk = pack_mlp_ln(seq, k_pad=kp)
xc = x.cuda()
kts = ops.ktiles(ops.Term(xc, kp))
seen = 0
for it in range(N):
    y = torch.empty(M, out, device="cuda")
    x0 = torch.empty(M, kp, device="cuda")
    hid = [torch.full((M, 64), float("nan"), device="cuda") for _ in range(5)]
    run_mlp(M, kts, pk, y, x0_out=x0, hid=hid)
    torch.cuda.synchronize()
    for l in range(5):
        d = (hid[l].double() - refs[l]).abs()
        tol = 1e-5 * float(refs[l].abs().max())
        if float(d.max()) > tol:
            rows = (d.amax(1) > tol).nonzero().flatten()
            r0 = int(rows[0])
            cols = (d[r0] > tol).nonzero().flatten()
            b = lins[l].bias.detach().double().cuda()
            got, want = hid[l][r0].double(), refs[l][r0]
            print(f"launch {it}: first wrong layer {l + 1} of 5 (Linear {l}), rows {r0}.. ({rows.numel()} rows, tiles {sorted(set((rows // 32).tolist()))[:5]}), "
                  f"columns {cols.tolist()[:6]}..{int(cols[-1])} ({cols.numel()})")
            c = cols[:4]
            print("     got  ", [f"{float(v):+.6f}" for v in got[c]])
            print("     want ", [f"{float(v):+.6f}" for v in want[c]])
            print("     bias ", [f"{float(v):+.6f}" for v in b[c]], " got-want", [f"{float(v):+.2e}" for v in (got - want)[c]])
            # is the difference the same for every wrong row of the tile (a per-column constant = a wrong bias)?
            t = r0 // 32
            blk = (hid[l].double() - refs[l])[t * 32:(t + 1) * 32][:, cols]
            live = (refs[l][t * 32:(t + 1) * 32][:, cols] > 0) if l < 4 else torch.ones_like(blk, dtype=torch.bool)
            print(f"     per-column spread of the difference over the tile's rows (live units): max std {float(torch.where(live, blk, torch.nan).nan_to_num().std(0).max()):.2e}, "
                  f"mean |diff| {float(blk.abs().mean()):.2e}")
            seen += 1
            break
    if seen >= 4:
        break
print("launches inspected", it + 1, "wrong", seen)
