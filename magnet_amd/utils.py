"""Host-side helpers mirroring the reference's utils.py (only what the hot path uses)."""
import torch


def make_coord(shape, ranges=None, flatten=True, device=None):
    """Cell-centre coordinates in [-1, 1] ('ij' meshgrid, row-major); reference utils.py:19-35."""
    seqs = []
    for i, n in enumerate(shape):
        v0, v1 = (-1, 1) if ranges is None else ranges[i]
        r = (v1 - v0) / (2 * n)
        seqs.append(v0 + r + (2 * r) * torch.arange(n, device=device).float())
    ret = torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)
    return ret.view(-1, ret.shape[-1]) if flatten else ret
