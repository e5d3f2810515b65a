"""Restatement of the un-vendored third-party arithmetic the reference calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: none of this is
covered by a test or fixture in the reference; it restates the published
algorithms of

* torch_scatter ``scatter(reduce='mean')``  — sum, count, clamp(min=1), divide;
  reached from the reference via PyG ``MessagePassing(aggr='mean')``
  (models/mpnn.py:46, models/magnet_gnn.py:54).
* torch_geometric 2.0.3 ``nn.norm.InstanceNorm`` defaults (affine=False,
  track_running_stats=False, eps=1e-5), called at models/mpnn.py:63,70.
* torch_cluster ``radius`` / ``radius_graph`` CUDA kernel (one thread per query,
  scan of the query's batch segment in ascending index, strict dist^2 < r^2,
  stop at max_num_neighbors; ``loop=False`` searches cap+1 then drops self),
  called at models/mpnn.py:245, models/magnet_gnn.py:293.
* torch_cluster ``knn`` CUDA kernel (insertion-sorted top-k, strict '>' so the
  lower index wins ties), called at models/magnet_gnn.py:247.
"""
from __future__ import annotations

import torch


# --------------------------------------------------------------------------- scatter / norm
def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """mean of ``src`` rows grouped by ``index`` (dim 0); empty groups give 0."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    cnt.clamp_(min=1)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


def instance_norm(x: torch.Tensor, batch: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """PyG InstanceNorm.forward(x, batch): per graph, per channel, biased variance."""
    nb = int(batch.max()) + 1
    cnt = torch.zeros(nb, dtype=x.dtype, device=x.device)
    cnt.index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    cnt = cnt.clamp_(min=1).view(-1, 1)
    mean = torch.zeros(nb, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, batch, x) / cnt
    xc = x - mean.index_select(0, batch)
    var = torch.zeros(nb, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, batch, xc * xc) / cnt
    return xc / (var + eps).sqrt().index_select(0, batch)


# --------------------------------------------------------------------------- neighbour search
def _segments(batch: torch.Tensor):
    """[(start, stop)] of the contiguous batch-major segments every reference caller produces."""
    nb = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=nb)
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    assert bool((batch[1:] >= batch[:-1]).all()), "batch must be sorted (batch-major)"
    return [(int(s), int(e)) for s, e in zip(starts, ends)]


def _r2_as_kernel_sees_it(r) -> torch.Tensor:
    """torch_cluster receives r as a double, squares it in double, then casts to the scalar type."""
    r64 = float(r)
    return torch.tensor(r64 * r64, dtype=torch.float64).to(torch.float32)


def radius(x: torch.Tensor, y: torch.Tensor, r, batch_x: torch.Tensor, batch_y: torch.Tensor,
           max_num_neighbors: int = 32, chunk: int = 2048) -> torch.Tensor:
    """For each y (query) the first ``max_num_neighbors`` x (ascending index, same graph) with
    dist^2 < r^2.  Returns [2, E] = (query index, found index), grouped by query ascending."""
    r2 = _r2_as_kernel_sees_it(r)
    x = x.view(x.shape[0], -1).float()
    y = y.view(y.shape[0], -1).float()
    segx, segy = _segments(batch_x), _segments(batch_y)
    rows, cols = [], []
    for (xs, xe), (ys, ye) in zip(segx, segy):
        xg = x[xs:xe]
        for q0 in range(ys, ye, chunk):
            q1 = min(ye, q0 + chunk)
            diff = y[q0:q1, None, :] - xg[None, :, :]
            # same accumulation order as the kernel: d = 0, then d += diff_k^2 for k = 0..dim-1
            d2 = torch.zeros(diff.shape[:2], dtype=torch.float32)
            for k in range(diff.shape[2]):
                d2 = d2 + diff[..., k] * diff[..., k]
            hit = d2 < r2
            keep = hit & (torch.cumsum(hit.to(torch.int32), 1) <= max_num_neighbors)
            qi, xi = keep.nonzero(as_tuple=True)          # row-major: query ascending, found ascending
            rows.append(qi + q0)
            cols.append(xi + xs)
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def radius_graph(x: torch.Tensor, r, batch: torch.Tensor | None = None, loop: bool = False,
                 max_num_neighbors: int = 32) -> torch.Tensor:
    """edge_index [2, E] with row 0 = found neighbour (PyG source j), row 1 = query (target i)."""
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long)
    qf = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    row, col = qf[1], qf[0]                                # flow='source_to_target'
    if not loop:
        m = row != col
        row, col = row[m], col[m]
    return torch.stack([row, col])


def knn(x: torch.Tensor, y: torch.Tensor, k: int, batch_x: torch.Tensor, batch_y: torch.Tensor) -> torch.Tensor:
    """[2, M*k] = (query index, k nearest x ascending by distance, lower index first on ties)."""
    x = x.view(x.shape[0], -1).float()
    y = y.view(y.shape[0], -1).float()
    rows, cols = [], []
    for (xs, xe), (ys, ye) in zip(_segments(batch_x), _segments(batch_y)):
        diff = y[ys:ye, None, :] - x[None, xs:xe, :]
        d2 = torch.zeros(diff.shape[:2], dtype=torch.float32)
        for kk in range(diff.shape[2]):
            d2 = d2 + diff[..., kk] * diff[..., kk]
        kk_eff = min(k, xe - xs)
        # stable sort = insertion with strict '>' (ties keep the earlier, i.e. lower, index first)
        order = torch.sort(d2, dim=1, stable=True).indices[:, :kk_eff]
        rows.append(torch.arange(ys, ye).repeat_interleave(kk_eff))
        cols.append((order + xs).reshape(-1))
    return torch.stack([torch.cat(rows), torch.cat(cols)])
