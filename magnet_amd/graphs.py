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


class TrainStepGraph:
    """``loss = model.training_step(batch, 0); loss.backward()`` captured ONCE per (batch shapes, coordinates) into a hipGraph and replayed
    (VERDICT r5 item 5): at the reference's own training shapes - 2 304 / 512 / 1 280 nodes per sample at batch 8 / 32 / 32,
    scripts/magnet_gnn/magnet_gnn_2d_b1_64_regular.sh:26-36, configs/model/magnet_gnn.yaml:3-14 - a step is 1 500 - 2 700 launches of
    5 - 100 us and ~40 ms of Python between them; the replay enqueues nothing per kernel.

        step = TrainStepGraph(model, optimizer)          # optimizer: the model's FusedAdam (configure_optimizers)
        loss = step(batch)                               # = zero_grad; training_step; backward; [all-reduce;] optimizer.step

    What is captured: the forward with its tape, the loss, the whole backward (the HIP Functions of magnet_amd.autograd / the MPNN
    rollout's backward), INCLUDING the re-pack of the weights the optimizer moved (the pack launches read the parameters: a replay packs
    the current values).  What stays outside, per step: one fill of the flat gradient buffer, the copy of the batch into the capture's
    static inputs, the range-guard word (read once behind the replay; a trip falls back to the eager step in the wider split), the
    gradient all-reduce (``reduce``: a callable, e.g. sharding.allreduce_gradients) and the Adam launch (its step count is a kernel
    argument).  The host generator is advanced as an eager step would advance it (MAgNet[GNN] draws its input noise there even when
    ``noise`` is 0, models/magnet_gnn.py:402-407).  Same kernels on the same data in the same order: the losses of a replayed
    trajectory equal the eager one's bit for bit (tests/test_train_gpu.py)."""

    def __init__(self, model, optimizer, reduce=None, max_entries: int = 2):
        self.model, self.opt, self.reduce, self.max_entries = model, optimizer, reduce, max_entries
        self._entries, self._bad = {}, set()
        self.replays = self.eager_steps = 0

    def _eager(self, batch):
        self.opt.zero_grad()
        loss = self.model.training_step(batch, 0)
        loss.backward()
        return loss

    def _finish(self, loss):
        if self.reduce is not None:
            self.reduce(self.model.parameters())
        self.opt.step()
        return loss

    def __call__(self, batch: dict):
        from . import packing
        m = self.model
        topo = tuple(getattr(m, "TOPOLOGY_KEYS", ())) or tuple(k for k in batch if k.startswith(("coords", "x")))
        key = (packing.wide(),) + tuple((k, tuple(v.shape), v.dtype, str(v.device)) for k, v in sorted(batch.items()))
        if key in self._bad:                                   # (this model / batch shape cannot be captured: plain steps)
            self.eager_steps += 1
            return self._finish(self._eager(batch))
        e = self._entries.get(key)
        if e is not None:
            for k in topo:                                     # the capture bakes the graph topology in: same coordinates?
                if k not in batch:
                    continue
                seen = e["seen"].get(k)                        # (tensor kept alive, its identity - address, shape, version - when compared)
                if seen is not None and seen[0] is batch[k] and seen[1] == _ident(batch[k]):
                    continue
                if not bool(torch.equal(e["static"][k], batch[k].to(e["static"][k].dtype))):
                    e = None
                    break
                e["seen"][k] = (batch[k], _ident(batch[k]))
        if e is None:
            e = self._capture(batch, key, topo)
            if e is None:
                self.eager_steps += 1
                return self._finish(self._eager(batch))
        for k, v in batch.items():
            if k not in topo:
                e["static"][k].copy_(v)
        for shape in e["draws"]:                               # the host generator moves as in an eager step
            torch.randn(shape)
        self.opt.zero_grad()
        e["graph"].replay()
        self.replays += 1
        if hasattr(m, "_range_trip"):
            from .packing import note_unchecked
            note_unchecked(next(m.parameters()).device)        # (every replay re-packs the weights: the packers' word is due as well)
            why = m._range_trip()
        else:                                                  # MPNN: the data guard's word + the packers' (the captured re-pack left it unread)
            from .packing import bad_weights
            why = m._guard.tripped()
            if bad_weights(next(m.parameters()).device, training=True):
                why = why or "a weight left the range of the fp16 planes"
        if why:                                                # out of the fp16 pieces' range: this step again, eagerly, in the wider split
            m._widen(why)
            self._entries.clear()
            self.eager_steps += 1
            return self._finish(self._eager(batch))
        return self._finish(e["loss"].clone())

    def _capture(self, batch, key, topo):
        static = {k: (v.float().clone() if v.is_floating_point() else v.clone()) for k, v in batch.items()}
        keep, draws = [], []
        _holders.append(keep)
        orig_randn = torch.randn

        def counting_randn(*shape, **kw):                      # (which host draws does a step make?  device draws are not ours)
            if not kw.get("device") and not kw.get("generator"):
                draws.append(shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape)
            return orig_randn(*shape, **kw)

        try:
            state = torch.get_rng_state()
            for i in range(2):                                 # warm: topology, packs, lazy buffers, the backward's transposed packs
                if i == 1:
                    torch.randn = counting_randn
                try:
                    self._eager(static)
                finally:
                    torch.randn = orig_randn
            torch.cuda.synchronize()
            torch.set_rng_state(state)                         # (the warm-up steps are not steps of the run)
            self.opt.zero_grad()
            from .optim import flat_gradient
            if flat_gradient([p for p in self.model.parameters() if p.requires_grad]) is None:
                # the replay ADDS into the gradient tensors the capture saw: they must be the optimizer's persistent flat views
                # (FusedAdam.zero_grad; MAGNET_FLAT_GRADS=0 or a foreign optimizer set .grad to None every step)
                self._bad.add(key)
                return None
            if hasattr(self.model, "_weights_changed"):        # the captured forward must RE-PACK the weights (in place, from the parameters'
                self.model._weights_changed()                  # current values): every replay follows an optimizer step
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    loss = self.model.training_step(static, 0)
                    loss.backward()
            except Exception as ex:                            # noqa: BLE001  (something in the step synchronises: say so once, go eager)
                import os
                import warnings
                if os.environ.get("MAGNET_GRAPH_DEBUG"):
                    import traceback
                    traceback.print_exc()
                warnings.warn(f"TrainStepGraph: this training step cannot be captured ({type(ex).__name__}: {str(ex)[:200]}); running eagerly",
                              RuntimeWarning)
                torch.cuda.synchronize()
                torch.set_rng_state(state)
                self._bad.add(key)
                return None
            torch.set_rng_state(state)
        finally:
            _holders.pop()
        if len(self._entries) >= self.max_entries:
            self._entries.pop(next(iter(self._entries)))
        e = {"graph": graph, "static": static, "loss": loss.detach(), "keep": keep, "draws": draws,
             "seen": {k: (batch[k], _ident(batch[k])) for k in topo if k in batch}}
        self._entries[key] = e
        return e
