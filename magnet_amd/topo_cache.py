"""Cache of device-built graph topologies keyed on the coordinate tensors they were built from.

The reference rebuilds its radius graph on every forward call (models/mpnn.py:283-287,
models/magnet_gnn.py:293) although the mesh does not move inside a rollout; here a topology is built
once and reused while its coordinates are unchanged.  "Unchanged" has to be decided without reading the
coordinates back, so an entry is recognised by identity - same storage address, shape, strides and torch
version counter as the tensor it was built from - and **the entry keeps that tensor alive**: while it is
referenced, the caching allocator cannot hand its address to another tensor, so an address match implies
the same storage, and the version counter (shared by all views of a storage) catches in-place writes.
(Keying on ``data_ptr`` alone is wrong: MAgNet training draws new query coordinates every step -
reference datamodule/dataset.py:95, dataset_2d.py:111 - the previous step's tensor is freed and the
allocator returns the same address with version 0.)

When identity misses but an entry of the same tag and shapes exists (a caller that re-creates equal
coordinates every call, e.g. ``batch["x"].float()`` from a float64 batch), the contents are compared on
the device (one ``torch.equal`` = one host sync) and the entry is re-keyed on a hit, so such callers pay a
compare instead of a graph build.
"""
from __future__ import annotations

import torch

from .graphs import hold


def _ident(t: torch.Tensor):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, t.dtype, str(t.device))


class TopologyCache:
    def __init__(self, capacity: int = 8):
        self.capacity = capacity
        self._entries = []                 # [tag, tensors (kept alive), idents, topo], most recent last

    def clear(self) -> None:
        self._entries.clear()

    def __deepcopy__(self, memo):                      # topologies belong to one model instance's tensors: a copy starts empty
        return TopologyCache(self.capacity)

    def __getstate__(self):
        return {"capacity": self.capacity, "_entries": []}

    def __len__(self) -> int:
        return len(self._entries)

    def topologies(self):
        """[(tag, topology)] of the live entries, oldest first (diagnostics / benchmarks)"""
        return [(e[0], e[3]) for e in self._entries]

    def get(self, tag, tensors, build):
        """topology for ``tensors`` (a tuple of coordinate tensors) under ``tag`` (hashable: everything else the
        topology depends on - batch size, radius, loop flag ...); ``build()`` makes it on a miss."""
        topo = self._get(tag, tuple(tensors), build)
        hold(topo)          # a hipGraph capture in progress bakes this topology's buffers in: it outlives an eviction here
        return topo

    def _get(self, tag, tensors, build):
        idents = tuple(_ident(t) for t in tensors)
        for k, e in enumerate(self._entries):
            if e[0] == tag and e[2] == idents:
                if k != len(self._entries) - 1:
                    self._entries.append(self._entries.pop(k))
                return e[3]
        for k, e in enumerate(self._entries):
            if e[0] != tag or len(e[1]) != len(tensors):
                continue
            if all(a.shape == b.shape and a.dtype == b.dtype and a.device == b.device for a, b in zip(e[1], tensors)) \
                    and all(a._version == i[3] for a, i in zip(e[1], e[2])) \
                    and all(bool(torch.equal(a, b)) for a, b in zip(e[1], tensors)):
                self._entries.pop(k)
                self._entries.append([tag, tensors, idents, e[3]])      # re-key on the tensors the caller now holds
                return e[3]
        topo = build()
        if len(self._entries) >= self.capacity:
            self._entries.pop(0)
        self._entries.append([tag, tensors, idents, topo])
        return topo
