"""``MLP(in_dim, hidden_list, out_dim)`` parameter container with the reference's layout
(models/backbones/mlp.py:9-28: Linear at ``layers.{0,2,..}``, an activation in between).  The
arithmetic runs in magnet_mlp_f32; ``packed()`` returns the fragment-ordered weights."""
from __future__ import annotations

from torch import nn

from ...packing import chain_gemm, default_gemm, pack, pad_to


class MLP(nn.Module):
    def __init__(self, in_dim, hidden_list, out_dim, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("the HIP MLP kernels implement the ReLU variant the hot path uses")
        dims = [in_dim] + list(hidden_list)
        mods = []
        for a, b in zip(dims[:-1], dims[1:]):
            mods += [nn.Linear(a, b), nn.ReLU()]
        mods.append(nn.Linear(dims[-1], out_dim))
        self.layers = nn.ModuleList(mods)
        self.in_dim, self.hidden, self.out_dim = in_dim, hidden_list[0], out_dim
        if len(set(hidden_list)) != 1 or self.hidden % 32:
            raise NotImplementedError("hidden widths must be equal and a multiple of 32")

    def linears(self):
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def packed(self, k_pad=None, col_map=None, mode=None):
        lins = self.linears()
        mode = chain_gemm(len(lins), self.hidden, mode or default_gemm())
        out = [pack(lins[0].weight, lins[0].bias, mode, k_pad=k_pad or pad_to(self.in_dim), col_map=col_map)]
        out += [pack(l.weight, l.bias, mode) for l in lins[1:]]
        return out
