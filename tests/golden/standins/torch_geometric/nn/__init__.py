"""PyG 2.0.3 API surface the reference touches, restated on oracle.pyg_semantics."""
import inspect

import torch
from torch import nn

from oracle import pyg_semantics as S


class MessagePassing(nn.Module):
    """propagate(): *_i <- index_select(edge_index[1]), *_j <- index_select(edge_index[0]),
    message(**) -> scatter(reduce=aggr) over targets -> update(aggregated, **caller kwargs)."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr, self.node_dim = aggr, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        i, j = edge_index[1], edge_index[0]
        n = None
        for v in kwargs.values():
            if torch.is_tensor(v) and v.dim() >= 2:
                n = v.shape[self.node_dim]
                break

        def collect(fn, skip_first):
            params = list(inspect.signature(fn).parameters)[1 if skip_first else 0:]
            out = {}
            for p in params:
                if p == "edge_index":
                    out[p] = edge_index
                elif p.endswith("_i"):
                    out[p] = kwargs[p[:-2]].index_select(self.node_dim, i)
                elif p.endswith("_j"):
                    out[p] = kwargs[p[:-2]].index_select(self.node_dim, j)
                else:
                    out[p] = kwargs[p]
            return out

        msg = self.message(**collect(self.message, False))
        assert self.aggr == "mean" and self.node_dim in (-2, 0)
        agg = S.scatter_mean(msg, i, n)
        return self.update(agg, **collect(self.update, True))


class InstanceNorm(nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=False, track_running_stats=False):
        super().__init__()
        assert not affine and not track_running_stats
        self.eps = eps

    def forward(self, x, batch=None):
        if batch is None:
            batch = torch.zeros(x.shape[0], dtype=torch.long)
        return S.instance_norm(x, batch, self.eps)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    return S.radius_graph(x, r, batch, loop, max_num_neighbors)


def knn(x, y, k, batch_x=None, batch_y=None):
    return S.knn(x, y, k, batch_x, batch_y)
