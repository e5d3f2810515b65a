"""Pieces shared by MAgNet[GNN] and MAgNet[CNN]: the Encoder / InteractionNetwork / Processor /
Decoder parameter containers (reference models/magnet_gnn.py:11-137, copied verbatim into
magnet_cnn.py:13-140 and magnet_cnn_2d.py:13-140 there) and the HIP execution of the
encode-process part on a CSR graph.

Reference semantics kept on purpose: mean aggregation (Q1); edge latents are only
doubled per hop, e_k = 2^k e_0 (Q2) - so e_0 is stored once and scaled on load; the radius graph is
capped at 32 neighbours in ascending index and then *swapped*, i.e. messages are averaged at the
found neighbour, not at the query (Q3, Q11); node k gets the time of sample k mod B (Q5).
"""
from __future__ import annotations

import os
import types

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..lightning_compat import LightningModule
from ..graphs import RolloutGraph, capturing, hold, wants_graph  # noqa: F401
from ..packing import pad_to, pad_vec, param_list
from ..topo_cache import TopologyCache
from .backbones.mlp import MLP


def _mlp_ln(i, hidden, layers, o):
    return nn.Sequential(MLP(i, [hidden] * layers, o), nn.LayerNorm(o))


class Encoder(nn.Module):
    def __init__(self, node_in, node_out, edge_in, edge_out, mlp_layers, mlp_hidden):
        super().__init__()
        self.node_fn = _mlp_ln(node_in, mlp_hidden, mlp_layers, node_out)
        self.edge_fn = _mlp_ln(edge_in, mlp_hidden, mlp_layers, edge_out)


class InteractionNetwork(nn.Module):
    def __init__(self, node_in, node_out, edge_in, edge_out, mlp_layers, mlp_hidden):
        super().__init__()
        self.node_fn = _mlp_ln(node_in + edge_out, mlp_hidden, mlp_layers, node_out)
        self.edge_fn = _mlp_ln(node_in + node_in + edge_in, mlp_hidden, mlp_layers, edge_out)


class Processor(nn.Module):
    def __init__(self, node_in, node_out, edge_in, edge_out, num_message_passing_steps, mlp_num_layers, mlp_hidden_dim):
        super().__init__()
        self.gnn_stacks = nn.ModuleList([
            InteractionNetwork(node_in, node_out, edge_in, edge_out, mlp_num_layers, mlp_hidden_dim)
            for _ in range(num_message_passing_steps)])


class Decoder(nn.Module):
    def __init__(self, node_in, node_out, mlp_layers, mlp_hidden):
        super().__init__()
        self.node_fn = MLP(node_in, [mlp_hidden] * mlp_layers, node_out)


def pack_mlp_ln(seq, k_pad=None, col_map=None, mode=None):
    """(packed 5 Linears, (gain, bias, eps)) of a Sequential(MLP, LayerNorm)."""
    mlp, ln = seq[0], seq[1]
    return types.SimpleNamespace(layers=mlp.packed(k_pad=k_pad, col_map=col_map, mode=mode),
                                 ln=(ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), ln.eps),
                                 ht=mlp.hidden // 32, ot=pad_to(mlp.out_dim) // 32, out_dim=mlp.out_dim, col_map=col_map,
                                 params=[p for lin in mlp.linears() for p in (lin.weight, lin.bias)] + [ln.weight, ln.bias])


def pack_mlp(mlp, k_pad=None, mode=None):
    return types.SimpleNamespace(layers=mlp.packed(k_pad=k_pad, mode=mode), ln=None, ht=mlp.hidden // 32,
                                 ot=pad_to(mlp.out_dim) // 32, out_dim=mlp.out_dim, col_map=None,
                                 params=[p for lin in mlp.linears() for p in (lin.weight, lin.bias)])


def pack_interaction_edge(seq, D):
    """InteractionNetwork.edge_fn (models/magnet_gnn.py:79-83) with its first Linear factored per node:
        W0 [x_i, x_j, e] + b0 = (W0a x)[i] + (W0b x)[j] + W0c e + b0          (exact algebra; summation order changes)
    -> (chain pack whose first layer is W0c only (K = D instead of 3 D), packed W0a, packed W0b).  The two per-node products
    are computed once per NODE by a single-Linear launch and gathered into the chain's first pre-activation (magnet_hip.h:
    magnet_mlp_args_t.pre): the first layer of the edge chain does a third of its flops and reads D instead of 3 D columns
    per edge - the factorisation SURVEY.md 7 (step 4) names, as for MPNN's message_net_1."""
    from ..packing import chain_gemm, default_gemm, derived, pack
    mlp, ln = seq[0], seq[1]
    lins = mlp.linears()
    mode = default_gemm()
    cmode = chain_gemm(len(lins), mlp.hidden, mode)                     # the chain's own Linears (the per-node products run the node kernels)
    W0 = lins[0].weight.detach()
    assert W0.shape[1] == 3 * D and D % 32 == 0
    # b0 travels with the receiver's product (the chain kernel starts its first accumulator FROM the gathered row, magnet_hip.h: pre);
    # `e_scale` = 2^k of hop k (Q2: e_k = 2^k e_0) is folded into W0c - a power of two, exact - so that the kernel loads e_0 unscaled
    def chain_for(e_scale):
        first = derived(lambda: lins[0].weight.detach()[:, 2 * D:] * float(e_scale))      # (refreshed in place on a pack replay)
        layers = [pack(first, None, cmode)] + [pack(l.weight, l.bias, cmode) for l in lins[1:]]
        return types.SimpleNamespace(layers=layers, ln=(ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), ln.eps),
                                     ht=mlp.hidden // 32, ot=pad_to(mlp.out_dim) // 32, out_dim=mlp.out_dim, col_map=None)
    return chain_for, pack(W0[:, :D], lins[0].bias, mode), pack(W0[:, D:2 * D], None, mode)


def pack_graph_net(enc: Encoder, proc: Processor, ldn: int):
    hops = []
    for k, g in enumerate(proc.gnn_stacks):
        D = g.node_fn[0].out_dim
        chain_for, pre_i, pre_j = pack_interaction_edge(g.edge_fn, D)
        factored = g.edge_fn[0].hidden <= D and len(g.edge_fn[0].linears()) == 5      # see MagnetBase._graph_net (fused chain only)
        hops.append(types.SimpleNamespace(edge=pack_mlp_ln(g.edge_fn), node=pack_mlp_ln(g.node_fn),
                                          edge_f=chain_for(2 ** k) if factored else None, pre_i=pre_i, pre_j=pre_j))
    return types.SimpleNamespace(node=pack_mlp_ln(enc.node_fn, k_pad=ldn), edge=pack_mlp_ln(enc.edge_fn, k_pad=ldn), hops=hops)


def _tag_recorder(tree, rec, _seen=None):
    """every chain namespace of a packed-weight tree learns its recorder: the backward builds its transposed packs lazily and caches
    them ON the namespace (autograd._tpack / _tpack16) - with a recorder they are re-packed in place with the forward's packs"""
    seen = _seen if _seen is not None else set()
    if id(tree) in seen:
        return
    seen.add(id(tree))
    if isinstance(tree, types.SimpleNamespace):
        if hasattr(tree, "layers"):
            tree._rec = rec
        for v in list(vars(tree).values()):
            _tag_recorder(v, rec, seen)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            _tag_recorder(v, rec, seen)
    elif isinstance(tree, dict):
        for v in tree.values():
            _tag_recorder(v, rec, seen)


def run_mlp(M, kts, pk, y, *, epi=ops.EPI_STORE, res=None, ncols=None, ldy=None, **seg):
    return ops.mlp(M, kts, pk.layers, nl=len(pk.layers), ht=pk.ht, ot=pk.ot, act=ops.ACT_RELU, last_act=False,
                   ln=pk.ln, epi=epi, y=y, res=res, ncols=ncols if ncols is not None else pk.out_dim, ldy=ldy, **seg)


class _RolloutScope:
    """deferred weight checks (packing.deferred_weight_checks) + the range guard armed for the chain kernels (ops.RangeGuard.armed)"""

    def __init__(self, model):
        from ..packing import deferred_weight_checks
        self.parts = (deferred_weight_checks(), model._guard.armed())
        if torch.is_grad_enabled() and model.training:      # a training step: loss.backward() follows (autograd.accumulate_param_grads)
            from ..autograd import accumulate_param_grads
            self.parts += (accumulate_param_grads(),)

    def __enter__(self):
        for p in self.parts:
            p.__enter__()

    def __exit__(self, *exc):
        for p in reversed(self.parts):
            p.__exit__(*exc)
        return False


class MagnetBase(LightningModule):
    """Graph building + encode-process on HIP, shared by the three MAgNet models."""
    TOPOLOGY_KEYS = ()          # batch entries that determine the graph topology (set by the models)

    def __getstate__(self):
        """pickle / copy.deepcopy: parameters and hparams travel, the per-instance device caches (packed weights, their recorder and
        launch plans, topology caches, captured rollouts) do not - the copy rebuilds them on first use"""
        from ..packing import state_without_device_caches
        return state_without_device_caches(self)

    def _graphed(self, fn, batch, keys, rows):
        """run ``fn(batch)`` (an inference rollout returning a tuple of tensors) - replayed from one hipGraph when the graph
        is small (launch bound), else directly"""
        with self._rollout_scope():
            if next(iter(batch.values())).is_cuda and wants_graph(self.hip_graph, rows):
                self._packed()
                out = self._rollout_graph(fn, {k: batch[k] for k in keys}, generation=self._pack_key)
            else:
                out = fn(batch)
        why = self._range_trip()
        if why:                                 # out of the fp16 pieces' range: the same rollout in the full-range split
            self._widen(why)
            return self._graphed(fn, batch, keys, rows)
        return out

    def _rollout_scope(self):
        """``with self._rollout_scope():`` a rollout / training step whose caller reads the range status at its END (_range_trip): the
        fp16 range check of weights packed inside is read there too, instead of right behind the pack launches - where the host would
        wait for the previous step's backward and optimizer kernels (7.8 ms per read at the reference's batch sizes)"""
        return _RolloutScope(self)

    def _range_trip(self):
        """the reason this rollout has to be repeated in the full-range split, or None: the data guard's status word and - when packs
        were made inside the rollout scope - the packers' (one host read each, at a point where the stream is nearly empty)"""
        from ..packing import pending_weight_check
        why = self._guard.tripped()
        if pending_weight_check(param_list(self)[0].device):
            why = self._weights_trip() or why
        return why

    def _weights_trip(self):
        from ..packing import bad_weights
        params = param_list(self)
        # the fp16x3 packs' range check: ONE host read per re-pack (not one per Linear); with gradients on, also the 31.98 limit of the
        # backward's scaled planes
        if not bad_weights(params[0].device, training=torch.is_grad_enabled() and self.training):
            return None
        self._pack_key = None
        if not all(bool(torch.isfinite(p).all()) for p in params):      # a diverged run must not be papered over
            raise FloatingPointError(f"{type(self).__name__}: non-finite weights")
        return "a weight does not fit the fp16 planes (|w| >= 65504; 31.98 for the one-accumulator chains)"

    def _widen(self, reason: str):
        """the range guard tripped (or a weight does not fit the fp16 planes): from here on the full-range bf16x6 split
        (packing.widen, process-wide).  The caller repeats what it was doing."""
        from ..packing import widen
        widen(f"{type(self).__name__}: {reason}")
        self._guard.clear()
        self._weights_changed()                 # every pack (EDSR's launch plans included) is rebuilt in the new mode

    def _init_common(self, hparams):
        self.lr, self.weight_decay = hparams.lr, hparams.weight_decay
        self.factor, self.step_size, self.loss = hparams.factor, hparams.step_size, hparams.loss
        self.criterion = {"l1": nn.L1Loss(), "l2": nn.MSELoss(), "smooth_l1": nn.SmoothL1Loss()}[self.loss]
        self.mse_criterion, self.mae_criterion = nn.MSELoss(), nn.L1Loss()
        self._pack_key, self._pack, self._pack_gen = None, None, 0
        # re-pack in place after an optimizer step (packing.PackRecorder) instead of rebuilding the packed namespace; A/B knob
        self._pack_replay = os.environ.get("MAGNET_PACK_REPLAY", "1") != "0"
        self._topo_cache = TopologyCache()
        # fp16x3 only: sticky device-side range / finiteness status, read once per rollout (ops.RangeGuard); a trip repeats the
        # rollout in the full-range bf16x6 split (_widen)
        from ..packing import default_gemm
        self._guard = ops.RangeGuard(default_gemm() == "fp16x3" and bool(getattr(hparams, "range_check", True)))
        self.optimizer_kind = getattr(hparams, "optimizer", "fused")     # "fused" (magnet_amd.optim.FusedAdam) | "torch"
        # hipGraph replay of whole inference rollouts (magnet_amd/graphs.py): True / False / "auto" = when the graph is small
        self.hip_graph = getattr(hparams, "hip_graph", "auto")
        self._rollout_graph = RolloutGraph(topology_keys=self.TOPOLOGY_KEYS)

    # ---- packed-weight cache -------------------------------------------------------------
    def _packed(self):
        from ..packing import PackRecorder, default_gemm, recording, weights_check_due
        params = param_list(self)
        key = (params[0].device, sum(p._version for p in params), self._pack_gen, default_gemm())
        if self._pack_key != key:
            rec = self.__dict__.get("_pack_rec")
            rkey = (params[0].device, default_gemm(), len(params))
            if self._pack is not None and rec is not None and rec.key == rkey and rec.usable() and self._pack_replay:
                # the weights changed, the structure did not: the recorded pack launches again, into the same buffers (packing.PackRecorder)
                rec.replay()
            else:
                rec = PackRecorder(params, rkey) if params[0].is_cuda else None
                with recording(rec):
                    self._pack = self._pack_weights()
                self.__dict__["_pack_rec"] = rec
                _tag_recorder(self._pack, rec)
            self._pack_key = key
            if capturing() and torch.is_grad_enabled():
                # a training step under capture (graphs.TrainStepGraph): its re-pack is part of the graph; the packers' word is read
                # behind every replay (_range_trip -> pending_weight_check)
                from ..packing import note_unchecked
                note_unchecked(params[0].device)
            elif capturing() or weights_check_due(params[0].device):    # (else: read at the end of the rollout, _range_trip)
                why = self._weights_trip()
                if why:
                    self._widen(why)
                    return self._packed()
        hold(self._pack)                    # a capture in progress keeps the packs its launches point into (magnet_amd/graphs.py)
        return self._pack

    def _weights_changed(self):
        """FusedAdam writes the parameters behind torch's version counters: drop every packed-weight cache"""
        self._pack_gen += 1
        for m in self.modules():
            if m is not self and hasattr(m, "weights_changed"):
                m.weights_changed()                    # EDSR keeps its own packs and prebuilt launch plans

    # ---- graph (reference _build_graph, models/magnet_gnn.py:286-310) ---------------------
    def _topology(self, coords: torch.Tensor, key):
        """radius_graph(loop=True, cap 32) on [B, N, d] coords, swapped (senders = query, receivers =
        found neighbour) and sorted by receiver; cached while the coordinate tensors are unchanged.
        ``key`` = (tag, tensors): the tensors ``coords`` was assembled from - the cache entry keeps them alive
        and recognises them by identity, or by content after a miss (magnet_amd/topo_cache.py)."""
        tag, tensors = key

        def build():
            B, N, d = coords.shape
            flat = coords.reshape(B * N, d).float().contiguous()
            gptr = ops.graph_ptr(B, N, flat.device)
            found, query, _, E = ops.radius_pairs(flat, self.radius, gptr, loop=True)
            csr = ops.csr_by_target(query, found, B * N)      # src = senders (query), tgt = receivers (found)
            return types.SimpleNamespace(csr=csr, n_graphs=B, nodes_per_graph=N, flat=flat)

        return self._topo_cache.get((tag, float(self.radius)), tensors, build)

    @staticmethod
    def _tkey(tag, *tensors):
        return (tag, tensors)

    def drop_topology_cache(self):
        self._topo_cache.clear()
        self._rollout_graph.clear()           # captured rollouts have their topology baked in

    def _build_graph(self, u, x, t, key=None):
        """u [B, N, T], x [B, N, d], t [B, T] -> (node_features [B*N, ld], topology).  Edge features
        (u_s - u_r, x_s - x_r) are formed inside the edge-encoder kernel from node_features."""
        B, N, T = u.shape
        d = x.shape[-1]
        topo = self._topology(x, key if key is not None else self._tkey("x", x))
        ld = pad_to(T + d + 1)
        nf = torch.empty(B * N, ld, device=u.device, dtype=torch.float32)
        uf = ops._f32(u.reshape(B * N, T))
        t_last = ops._f32(t[:, -1])
        L.check(L.lib().magnet_node_features_f32(ops._p(uf), ops._p(topo.flat), ops._p(t_last), B, N, T, d, ld,
                                                 ops._p(nf), ops._stream()), "node_features")
        self._guard.inputs(nf)
        return nf, topo

    # ---- Encoder + Processor (models/magnet_gnn.py:38-42, 70-117) --------------------------
    def _graph_net(self, pk, nf, topo, D):
        csr = topo.csr
        N, E, dev = csr.N, csr.E, nf.device
        ldn = nf.shape[1]
        T_ = ops.Term
        x = torch.empty(N, D, device=dev, dtype=torch.float32)
        x2 = torch.empty_like(x)
        agg = torch.empty_like(x)
        e0 = torch.empty(max(E, 1), D, device=dev, dtype=torch.float32)
        partials = torch.empty(((E + 31) // 32) * 2 * D + 1, device=dev, dtype=torch.float32)
        run_mlp(N, ops.ktiles(T_(nf, ldn)), pk.node, x)
        run_mlp(E, ops.ktiles((T_(nf, ldn, idx=csr.src, scale=1.0), T_(nf, ldn, idx=csr.tgt, scale=-1.0))), pk.edge, e0)
        # edge_fn([x_i, x_j, e]) with the first Linear factored per node (pack_interaction_edge) when that does not widen the
        # gathered rows (hidden <= latent: MAgNet[GNN] 128 / 128; the CNN models' 64-wide chains over 32-wide latents keep the
        # concatenated form - measured: C5 edge chain 1.55 -> 1.21 ms, C4b 0.38 -> 0.39 ms)
        Hm = pk.hops[0].edge.ht * 32 if pk.hops else 0
        factored = 0 < Hm <= D and pk.hops[0].edge_f is not None
        resident_halves = (factored and D == 128 and Hm == 128 and pk.hops[0].pre_i[0].dtype == torch.float16       # (fp16x3 planes)
                           and os.environ.get("MAGNET_FUSED_HALVES", "1") != "0")
        if factored:
            PQ = torch.empty(2, N, Hm, device=dev, dtype=torch.float32)   # per-node halves of the edge chains' first Linear
        for k, hop in enumerate(pk.hops):
            if factored:
                # P = W0a x, Q = W0b x in one launch (two column groups run back to back inside each workgroup), then the
                # chain over e_k = 2^k e_0 (Q2) alone
                if resident_halves:        # 128 -> 128 + 128, fp16x3: the MPNN per-node-halves kernel (weights resident in LDS), kin = 0
                    ops.mpnn_node_halves(x, None, hop.pre_i, hop.pre_j, PQ[0], PQ[1])
                else:
                    ops.mlp(N, ops.ktiles(T_(x, D)), [hop.pre_i, hop.pre_j], nl=1, ht=Hm // 32, ot=Hm // 32, act=ops.ACT_NONE, y=PQ,
                            ldy=Hm, colgroups=2, groups_in_block=True, y_group_stride=N * Hm)
                run_mlp(E, ops.ktiles(T_(e0, D)), hop.edge_f, agg, epi=ops.EPI_SEGMEAN, tgt=csr.tgt,
                        rowptr=csr.rowptr, partials=partials, pre=[(PQ[0], csr.tgt), (PQ[1], csr.src)])
            else:
                kts = ops.ktiles(T_(x, D, idx=csr.tgt), T_(x, D, idx=csr.src), T_(e0, D, scale=float(2 ** k)))
                run_mlp(E, kts, hop.edge, agg, epi=ops.EPI_SEGMEAN, tgt=csr.tgt, rowptr=csr.rowptr, partials=partials)
            ops.segmean_fixup(csr.rowptr, N, D, partials, agg)
            run_mlp(N, ops.ktiles(T_(agg, D), T_(x, D)), hop.node, x2, epi=ops.EPI_STORE_RES, res=x)
            x, x2 = x2, x
        return x

    # ---- the same two blocks with gradients (training): every op is a magnet_amd.autograd Function, outputs are
    # fresh tensors (autograd keeps the inputs of every op alive for its backward)
    def _build_graph_grad(self, u, x, t, key=None):
        from .. import autograd as AG
        B, N, T = u.shape
        topo = self._topology(x, key if key is not None else self._tkey("x", x))
        nf = AG.node_features(u.float(), topo.flat, t[:, -1].float(), pad_to(T + x.shape[-1] + 1))
        self._guard.inputs(nf.detach())
        return nf, topo

    def _graph_net_grad(self, pk, nf, topo, D):
        from .. import autograd as AG
        csr = topo.csr
        N, E = csr.N, csr.E
        ldn = nf.shape[1]
        T_ = ops.Term
        x = AG.mlp(N, [T_(nf, ldn)], pk.node, pk.node.params)
        e0 = AG.mlp(E, [(T_(nf, ldn, idx=csr.src, scale=1.0), T_(nf, ldn, idx=csr.tgt, scale=-1.0))], pk.edge, pk.edge.params)
        e0_grad = {}                    # d(e0): every hop's backward adds its share to one running sum (AG.mlp: grad_chain), hop 0 hands it over
        for k, hop in enumerate(pk.hops):
            agg = AG.mlp(E, [T_(x, D, idx=csr.tgt), T_(x, D, idx=csr.src), T_(e0, D, scale=float(2 ** k))], hop.edge,
                         hop.edge.params, epi=ops.EPI_SEGMEAN, csr=csr, out_rows=N, grad_chain=(e0, e0_grad, k == 0))
            x = AG.mlp(N, [T_(agg, D), T_(x, D)], hop.node, hop.node.params, epi=ops.EPI_STORE_RES, res=x)
        return x

    def _interaction_and_update_grad(self, pk, feats, coords, t, T, last, key):
        from .. import autograd as AG
        B, n, _ = feats.shape
        nf, topo = self._build_graph_grad(feats, coords, t[:, :T], key=key)
        x = self._graph_net_grad(pk.graph2, nf, topo, self.latent_dim)
        T_out = t.shape[1] - T
        ret = AG.mlp(B * n, [ops.Term(x, self.latent_dim)], pk.decoder, pk.decoder.params)
        out = AG.euler_update(ret, last, t, B, n, T, T_out)
        self._guard.outputs(out.detach())
        return out

    # ---- second half of every MAgNet forward: joint graph, decoder, Euler update ------------
    def _interaction_and_update(self, pk, feats, coords, t, T, last, key):
        """feats [B, n, T], coords [B, n, d] -> out [B, T_out, n, 1] (models/magnet_gnn.py:351-369)."""
        B, n, _ = feats.shape
        nf, topo = self._build_graph(feats, coords, t[:, :T], key=key)
        x = self._graph_net(pk.graph2, nf, topo, self.latent_dim)
        T_out = t.shape[1] - T
        ret = torch.empty(B * n, pk.decoder.out_dim, device=x.device, dtype=torch.float32)
        run_mlp(B * n, ops.ktiles(ops.Term(x, self.latent_dim)), pk.decoder, ret)
        out = torch.empty(B, T_out, n, 1, device=x.device, dtype=torch.float32)
        tt = ops._f32(t)
        L.check(L.lib().magnet_euler_update_f32(ops._p(ret), ret.shape[1], ops._p(ops._f32(last.reshape(-1))), ops._p(tt),
                                                tt.shape[1], B, n, T, T_out, ops._p(out), ops._stream()), "euler_update")
        self._guard.outputs(out)
        return out

    def _projector(self, pk, z, rows):
        """projector MLP n_chan -> hidden x4 -> 1 on [rows, n_chan] (models/magnet_gnn.py:339)."""
        out = torch.empty(rows, 1, device=z.device, dtype=torch.float32)
        run_mlp(rows, ops.ktiles(ops.Term(z, z.shape[1])), pk.projector, out)
        return out

    def configure_optimizers(self):
        """Adam(lr, weight_decay) + StepLR(step_size, gamma=factor) as in the reference (models/magnet_gnn.py:377-386); on a HIP
        device the optimizer is magnet_amd.optim.FusedAdam (same update rule, one kernel over a flat parameter buffer)."""
        if next(self.parameters()).is_cuda and self.optimizer_kind == "fused":
            from ..optim import FusedAdam
            opt = FusedAdam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
            opt.on_step = self._weights_changed
        else:
            opt = torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        sch = torch.optim.lr_scheduler.StepLR(opt, step_size=self.step_size, gamma=self.factor)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sch}}


__all__ = ["Encoder", "InteractionNetwork", "Processor", "Decoder", "MagnetBase", "pack_mlp", "pack_mlp_ln",
           "pack_graph_net", "run_mlp", "pad_vec"]
