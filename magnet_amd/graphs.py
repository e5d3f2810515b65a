"""hipGraph replay of a whole free-running rollout (SURVEY.md 8(f) rank 1 context: the launch-bound regime).

The small BASELINE configs (C1: 2 x 50 nodes, C2 / C3: tens to hundreds of nodes) are launch bound: a rollout is several
hundred kernels of a few microseconds each plus the host glue between the forward calls, and the host cannot enqueue
them as fast as the GPU retires them.  ``RolloutGraph`` captures one complete rollout - every forward call and the
feedback glue between them - into ONE hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replays it:

* the batch is copied into static input buffers, the graph is replayed, the outputs are cloned out of the graph's pool;
* a capture is keyed on the batch's shapes / dtypes and on the model's packed-weight generation (an optimizer step
  re-packs the weights into new buffers, so the old graph's pointers are stale);
* the radius graph / k-NN topology is baked into the capture (its build needs the edge count on the host).  The
  coordinate tensors of every call are therefore compared with the captured ones - by identity when the caller passes the
  same tensors again, else by content (one ``torch.equal`` = one host sync per rollout) - and a mismatch re-captures.
  "Identity" is only meaningful while the tensor it was taken from is alive (a freed block comes back from the caching
  allocator at the same address with version 0), so an entry KEEPS the caller's topology tensors, as
  ``magnet_amd.topo_cache.TopologyCache`` does;
* a capture bakes in pointers to buffers that model-side caches own (radius graphs / CSR arrays in a ``TopologyCache``,
  EDSR launch plans with their feature maps and zero-padding rows, packed weights).  Those caches evict and replace
  entries on their own schedule, so everything a capture touched is registered with ``hold()`` while it runs and the
  entry keeps it alive for as long as the graph can be replayed;
* nothing inside the captured region may synchronise: the models skip their range-guard read while capturing and
  ``RolloutGraph`` reads it after the replay.

Used by default when a rollout's graph has at most ``AUTO_MAX_ROWS`` node rows (``hip_graph`` hparam: True / False / "auto").
MPNN also takes ``hip_graph="rebuild"``: the radius graph + CSR are built afresh before EVERY replay (what the reference does per
forward call) and copied into the buffers the captured kernels read - the rollout's ~300 launches become one, the graph
construction stays in every rollout (models/mpnn.py: _replay_with_rebuild).
"""
from __future__ import annotations

import torch

# "auto": every validation shape of the reference's run scripts (64 x 64 meshes at batch 4 - 32: 16 k - 160 k node rows) is replayed;
# the 256^2 metric cases (262 k rows and up) are enqueued kernel by kernel unless the caller asks (bench.py: hip_graph=True / "rebuild")
AUTO_MAX_ROWS = 200_000
VOLATILE_AFTER = 2           # captures in a row that were thrown away by a different mesh before they were replayed twice


def capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _ident(t: torch.Tensor):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version)


_holders = []          # stack of keep-alive lists: one per capture in progress (captures do not nest in practice)


def hold(*objs) -> None:
    """called by the model-side caches (TopologyCache.get, EDSR._plan, the packed-weight caches) with the objects whose
    device buffers the kernels about to be enqueued will read or write: while a rollout is being captured they are added
    to the capture's keep-alive list, otherwise this is a no-op"""
    if _holders:
        _holders[-1].extend(objs)


class RolloutGraph:
    def __init__(self, topology_keys, max_entries: int = 4):
        """``topology_keys``: the batch entries that determine the graph topology (coordinates)."""
        self.topology_keys, self.max_entries = tuple(topology_keys), max_entries
        self._entries = {}
        self._churn = {}           # key -> captures in a row that a new mesh replaced before they paid off

    def clear(self):
        self._entries.clear()
        self._churn.clear()

    def volatile(self, key) -> bool:
        """this batch shape keeps arriving with a different mesh (irregular data: new coordinates every batch): a capture costs three
        rollouts and is thrown away at once, so such shapes are enqueued kernel by kernel"""
        return self._churn.get(key, 0) >= VOLATILE_AFTER

    def __deepcopy__(self, memo):                      # captured graphs belong to one model instance: a copy starts empty
        return RolloutGraph(self.topology_keys, self.max_entries)

    def __getstate__(self):
        return {"topology_keys": self.topology_keys, "max_entries": self.max_entries, "_entries": {}, "_churn": {}}

    def __call__(self, fn, batch: dict, generation):
        """``fn(batch: dict) -> tuple of tensors`` runs one rollout with no host synchronisation once its caches are warm"""
        key = (generation,) + tuple((k, tuple(v.shape), v.dtype, str(v.device)) for k, v in sorted(batch.items()))
        if self.volatile(key):
            return fn(batch)
        e = self._entries.get(key)
        if e is not None:
            for k in self.topology_keys:                      # same coordinates as the capture?
                seen = e["seen"].get(k)                       # (tensor kept alive, its identity when it was compared)
                if seen is not None and seen[0] is batch[k] and seen[1] == _ident(batch[k]):
                    continue
                if not bool(torch.equal(e["static"][k], batch[k].to(e["static"][k].dtype))):
                    self._churn[key] = self._churn.get(key, 0) + 1 if e["replays"] < 2 else 0
                    e = None
                    break
                e["seen"][k] = (batch[k], _ident(batch[k]))
        if e is None:
            e = self._capture(fn, batch)
            if len(self._entries) >= self.max_entries:
                self._entries.pop(next(iter(self._entries)))
            self._entries[key] = e                             # its static buffers already hold this batch
        else:
            for k, v in batch.items():
                if k not in self.topology_keys:                # topology inputs are known to be equal: leave them (and
                    e["static"][k].copy_(v)                    # their version counters) alone
        e["graph"].replay()
        e["replays"] += 1
        if e["replays"] >= 2:
            self._churn[key] = 0
        return tuple(o.clone() for o in e["out"])

    def peek(self, batch: dict, generation):
        """the capture a call with this batch would replay (shapes / dtypes / generation only: the caller vouches for the topology),
        or None"""
        key = (generation,) + tuple((k, tuple(v.shape), v.dtype, str(v.device)) for k, v in sorted(batch.items()))
        return self._entries.get(key)

    def _capture(self, fn, batch: dict):
        # floating inputs are kept as fp32 (what every model converts to first): `.float()` inside the rollout is then a view
        # of the static buffer and the topology caches recognise it by identity in the capture pass
        static = {k: (v.float().clone() if v.is_floating_point() else v.clone()) for k, v in batch.items()}
        keep = []
        _holders.append(keep)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                             # builds topology / packed weights / lazy buffers (may sync)
                    fn(static)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = fn(static)
        finally:
            _holders.pop()
        out = tuple(out) if isinstance(out, (tuple, list)) else (out,)
        return {"graph": graph, "static": static, "out": out, "keep": keep, "replays": 0,
                "seen": {k: (batch[k], _ident(batch[k])) for k in self.topology_keys}}


def wants_graph(setting, rows: int) -> bool:
    """``hip_graph`` hparam: True / False / "auto" (default: replay when the rollout's graph has few rows)"""
    if setting in (True, False):
        return bool(setting)
    if setting == "rebuild":
        return True
    return rows <= AUTO_MAX_ROWS
