"""MAgNet[GNN] on MI355X - drop-in for the reference's ``models/magnet_gnn.py`` (``MAgNetGNN``).

forward(x_lr, lr_coords, hr_coords, t, hr_last) -> (out_hr, out_lr, hr_points), same shapes and
state-dict keys.  ``coord_dim`` (optional hparam, default 2 = the reference, which is hard-wired to
2-D coordinates) parametrises the +3 / +2 / +4 widths for the 1-D configs.

  encoder + processor on the LR mesh   -> MagnetBase._graph_net (magnet_mlp_f32, fused CSR mean)
  knn(k = codec_neighbors)             -> magnet_knn_f32                     magnet_gnn.py:247
  proj_head on the two nearest + blend -> magnet_mlp_f32 (1 Linear on [enc[nn] | rel]) +
                                          magnet_inr_knn_blend_f32           magnet_gnn.py:254-280
  projector                            -> magnet_mlp_f32                     magnet_gnn.py:339
  joint graph, decoder, Euler update   -> MagnetBase._interaction_and_update magnet_gnn.py:342-369
"""
from __future__ import annotations

import types

import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..graphs import capturing
from ..packing import default_gemm, pack, pad_to, param_list
from .backbones.mlp import MLP
from .magnet_common import (Decoder, Encoder, MagnetBase, Processor, pack_graph_net, pack_mlp)

_MODES = {"area": 0, "knn": 1, "sph": 2}


class MAgNetGNN(MagnetBase):
    TOPOLOGY_KEYS = ("coords_hr", "coords_lr")

    def __init__(self, hparams):
        super().__init__()
        self.save_hyperparameters()
        self._init_common(hparams)
        self.time_slice = hparams.time_slice
        self.num_message_passing_steps = hparams.num_message_passing_steps
        self.latent_dim, self.mlp_layers, self.mlp_hidden = hparams.latent_dim, hparams.mlp_layers, hparams.mlp_hidden
        self.n_chan, self.radius = hparams.n_chan, hparams.radius
        self.codec_neighbors, self.teacher_forcing = hparams.codec_neighbors, hparams.teacher_forcing
        self.noise, self.interpolation = hparams.noise, hparams.interpolation
        self.coord_dim = d = getattr(hparams, "coord_dim", 2)
        if not 1 <= self.mlp_layers <= 5:        # 4 (every shipped config): the fused 5-Linear kernel; other depths run layer by layer
            raise NotImplementedError("mlp_layers must be between 1 and 5")
        T, D, Lm, Hm, K = self.time_slice, self.latent_dim, self.mlp_layers, self.mlp_hidden, self.num_message_passing_steps
        self.encoder = Encoder(T + d + 1, D, T + d, D, Lm, Hm)
        self.processor = Processor(D, D, D, D, K, Lm, Hm)
        self.proj_head = nn.Linear(D + d + 2, self.n_chan)
        self.projector = MLP(self.n_chan, [Hm] * Lm, 1)
        self._encoder = Encoder(T + d + 1, D, T + d, D, Lm, Hm)
        self._processor = Processor(D, D, D, D, K, Lm, Hm)
        self._decoder = Decoder(D, T, Lm, Hm)

    def _pack_weights(self):
        from ..packing import derived
        T, D, d = self.time_slice, self.latent_dim, self.coord_dim
        ldn = pad_to(T + d + 1)
        head = self.proj_head
        dev = head.weight.device
        cols = torch.cat([torch.arange(D), D + 1 + torch.arange(d)]).to(dev)   # enc, rel
        cmap = torch.cat([torch.arange(D), D + torch.arange(d)])
        # (derived: computed from parameters - refreshed in place when the packs are replayed after an optimizer step)
        return types.SimpleNamespace(
            graph1=pack_graph_net(self.encoder, self.processor, ldn),
            graph2=pack_graph_net(self._encoder, self._processor, ldn),
            proj_base=pack(derived(lambda: head.weight.detach()[:, cols]), head.bias, default_gemm(), k_pad=D + 32, col_map=cmap),
            proj_wv=derived(lambda: head.weight.detach()[:, D].float().contiguous()),
            proj_wt=derived(lambda: head.weight.detach()[:, D + d + 1].float().contiguous()),
            projector=pack_mlp(self.projector), decoder=pack_mlp(self._decoder.node_fn))

    def forward(self, x_lr, lr_coords, hr_coords, t, hr_last):
        if torch.is_grad_enabled() and any(p.requires_grad for p in param_list(self)):
            return self._forward_grad(x_lr, lr_coords, hr_coords, t, hr_last)
        with torch.no_grad():
            return self._forward(x_lr, lr_coords, hr_coords, t, hr_last)

    def _forward_grad(self, x_lr, lr_coords, hr_coords, t, hr_last):
        """the forward of ``_forward`` built from magnet_amd.autograd Functions (fused HIP forward kernels, HIP backward
        kernels); torch autograd only chains them and differentiates the cat / permute / reshape glue."""
        from .. import autograd as AG
        B, T, C, Lr = x_lr.shape
        Nq, d, D, Cn = hr_coords.shape[1], self.coord_dim, self.latent_dim, self.n_chan
        dev = x_lr.device
        pk = self._packed()
        t = t.float()
        lr_coords = lr_coords.reshape(B, Lr, d).float()
        hr_coords = hr_coords.reshape(B, Nq, d).float()
        u = x_lr.permute(0, 3, 1, 2).reshape(B, Lr, T * C).float().contiguous()
        nf, topo = self._build_graph_grad(u, lr_coords, t[:, :T], key=self._tkey("lr", lr_coords))
        enc = self._graph_net_grad(pk.graph1, nf, topo, D)
        with torch.no_grad():
            flr, fhr = topo.flat, ops._f32(hr_coords.reshape(B * Nq, d))
            nn_idx = ops.knn(flr, ops.graph_ptr(B, Lr, dev), fhr, ops.graph_ptr(B, Nq, dev), self.codec_neighbors)
            M2 = 2 * B * Nq
            nn2 = torch.empty(M2, dtype=torch.int32, device=dev)
            rel = torch.empty(M2, 32, device=dev, dtype=torch.float32)
            w = torch.empty(M2, device=dev, dtype=torch.float32)
            L.check(L.lib().magnet_inr_knn_prepare_f32(ops._p(flr), ops._p(fhr), ops._p(nn_idx), self.codec_neighbors, d, B,
                                                       Lr, Nq, ops._p(nn2), ops._p(rel), ops._p(w), ops._stream()), "inr_knn_prepare")
        ot = Cn // 32
        Wp = self.proj_head.weight
        # one namespace for as long as the pack tree lives: the backward caches proj_head's transposed packs on it (a fresh one per
        # call re-packed them in each of the four forwards of a training rollout)
        pb = pk.__dict__.get("_proj_ns")
        if pb is None:
            cols = torch.cat([torch.arange(D), D + 1 + torch.arange(d)]).to(dev)       # enc, rel columns of proj_head
            cmap = torch.cat([torch.arange(D), D + torch.arange(d)]).to(dev)
            pb = pk.__dict__["_proj_ns"] = types.SimpleNamespace(layers=[pk.proj_base], ln=None, ht=ot, ot=ot, out_dim=Cn, col_map=cmap,
                                                                 cols=cols, _rec=self.__dict__.get("_pack_rec"))
        base = AG.mlp(M2, [ops.Term(enc, D, idx=nn2), ops.Term(rel, 32)], pb, [Wp, self.proj_head.bias], first_cols=pb.cols)
        meta = types.SimpleNamespace(w=w, nn2=nn2, t=t.contiguous(), B=B, L=Lr, Nq=Nq, T=T, C=Cn, mode=_MODES[self.interpolation],
                                     col_v=D, col_t=D + d + 1, inv_cache={})
        z = AG.knn_blend(base, u.reshape(B * Lr, T), Wp, meta)
        pj = pk.projector
        hr_points = AG.mlp(B * Nq * T, [ops.Term(z, Cn)], pj, pj.params).reshape(B, Nq, T)
        feats = torch.cat([u, hr_points], dim=1)
        coords = torch.cat([lr_coords, hr_coords], dim=1)
        last = torch.cat([x_lr[:, -1].permute(0, 2, 1).float(), hr_last.float()], dim=1)
        out = self._interaction_and_update_grad(pk, feats, coords, t, T, last, key=self._tkey("all", lr_coords, hr_coords))
        return out[:, :, Lr:], out[:, :, :Lr], hr_points.permute(0, 2, 1)[..., None]

    def _forward(self, x_lr, lr_coords, hr_coords, t, hr_last):
        B, T, C, Lr = x_lr.shape
        Nq, d, D, Cn = hr_coords.shape[1], self.coord_dim, self.latent_dim, self.n_chan
        dev = x_lr.device
        pk = self._packed()
        t = t.float()
        lr_coords = lr_coords.reshape(B, Lr, d).float()
        hr_coords = hr_coords.reshape(B, Nq, d).float()
        u = x_lr.permute(0, 3, 1, 2).reshape(B, Lr, T * C).float().contiguous()      # magnet_gnn.py:331-332
        # encode-process the LR mesh
        nf, topo = self._build_graph(u, lr_coords, t[:, :T], key=self._tkey("lr", lr_coords))
        enc = self._graph_net(pk.graph1, nf, topo, D)
        # INR decoder (continuous_decoder, magnet_gnn.py:224-283)
        M2 = 2 * B * Nq

        def build_inr():
            # k nearest LR nodes of every query + their relative coordinates and blend weights: a function of the coordinates alone
            flr, fhr = topo.flat, ops._f32(hr_coords.reshape(B * Nq, d))
            nn_idx = ops.knn(flr, ops.graph_ptr(B, Lr, dev), fhr, ops.graph_ptr(B, Nq, dev), self.codec_neighbors)
            nn2_ = torch.empty(M2, dtype=torch.int32, device=dev)
            rel_ = torch.empty(M2, 32, device=dev, dtype=torch.float32)
            w_ = torch.empty(M2, device=dev, dtype=torch.float32)
            L.check(L.lib().magnet_inr_knn_prepare_f32(ops._p(flr), ops._p(fhr), ops._p(nn_idx), self.codec_neighbors, d, B,
                                                       Lr, Nq, ops._p(nn2_), ops._p(rel_), ops._p(w_), ops._stream()), "inr_knn_prepare")
            return types.SimpleNamespace(nn2=nn2_, rel=rel_, w=w_)

        # cached with the topology (same cache, same identity / content rules): a free-running rollout decodes at the same queries in
        # every one of its forward calls
        inr = self._topo_cache.get(("inr", (self.codec_neighbors, B, Lr, Nq)), (lr_coords, hr_coords), build_inr)
        nn2, rel, w = inr.nn2, inr.rel, inr.w
        base = torch.empty(M2, Cn, device=dev, dtype=torch.float32)
        ot = Cn // 32
        ops.mlp(M2, ops.ktiles(ops.Term(enc, D, idx=nn2), ops.Term(rel, 32)), [pk.proj_base], nl=1, ht=ot, ot=ot,
                act=ops.ACT_NONE, y=base)
        z = torch.empty(B * Nq * T, Cn, device=dev, dtype=torch.float32)
        vals = u.reshape(B * Lr, T)
        L.check(L.lib().magnet_inr_knn_blend_f32(ops._p(base), ops._p(w), ops._p(vals), ops._p(nn2), ops._p(t), t.shape[1],
                                                 ops._p(pk.proj_wv), ops._p(pk.proj_wt), B, Lr, Nq, T, Cn,
                                                 _MODES[self.interpolation], ops._p(z), ops._stream()), "inr_knn_blend")
        hr_points = self._projector(pk, z, B * Nq * T).reshape(B, Nq, T)
        # joint graph over LR nodes + queries
        feats = torch.cat([u, hr_points], dim=1)
        coords = torch.cat([lr_coords, hr_coords], dim=1)
        last = torch.cat([x_lr[:, -1].permute(0, 2, 1).float(), hr_last.float()], dim=1)
        out = self._interaction_and_update(pk, feats, coords, t, T, last, key=self._tkey("all", lr_coords, hr_coords))
        return out[:, :, Lr:], out[:, :, :Lr], hr_points.permute(0, 2, 1)[..., None]

    def _noisy(self, x, on):
        """training-time input noise, magnet_gnn.py:402-407 / :424-428: ``noise * torch.randn(shape)`` drawn from torch's
        default HOST generator (as the reference does - same seed, same call order, same numbers), added on the device.
        The reference draws even when ``noise`` is 0 (the shipped config, configs/model/magnet_gnn.yaml:13): the numbers are
        then drawn and discarded here too, so that everything that reads the host generator after a training step
        (DataLoader shuffle seeds, later initialisations) sees the stream the reference would have left."""
        if not on:
            return x
        draw = torch.randn(x.shape)
        if not self.noise:
            return x
        return x + (self.noise * draw).to(x.device)

    def _rollout(self, batch, teacher_forcing=False, noise=False):
        with self._rollout_scope():
            pred, target, hr_hat, vals = self._rollout_body(batch, teacher_forcing, noise)
        if not capturing():
            why = self._range_trip()
            if why:                             # out of the fp16 pieces' range: the same rollout in the full-range split (the input
                self._widen(why)                # noise, when configured, is drawn again)
                return self._rollout(batch, teacher_forcing, noise)
        return pred, target, hr_hat, vals

    def _rollout_body(self, batch, teacher_forcing, noise):
        t, u = batch["t"].float(), batch["lr_frames"].float()
        vals = batch["hr_points"].float()
        hr_c, lr_c = batch["coords_hr"].float(), batch["coords_lr"].float()
        T = self.time_slice
        fut = vals[:, T:]
        outs, hrs = [], []
        inp = self._noisy(u[:, :T], noise)                     # reference order: input noise first, then hr_last
        hr_last = self._noisy(vals[:, T - 1], noise)
        for i in range(fut.shape[1] // T):
            out_hr, out_lr, hr_points = self.forward(inp, lr_c, hr_c, t[:, i * T:(i + 2) * T], hr_last)
            outs.append(torch.cat([out_hr, out_lr], dim=2))
            hrs.append(hr_points)
            if teacher_forcing:
                inp, hr_last = u[:, (i + 1) * T:(i + 2) * T], vals[:, (i + 2) * T - 1]
            else:
                inp, hr_last = out_lr.permute(0, 1, 3, 2), out_hr[:, -1]
            inp = self._noisy(inp, noise)
            hr_last = self._noisy(hr_last, noise)
        pred = torch.cat(outs, dim=1)
        target = torch.cat([fut, u[:, T:].permute(0, 1, 3, 2)], dim=2)
        return pred, target, torch.cat(hrs, dim=1), vals

    def training_step(self, train_batch, batch_idx):
        pred, target, hr_hat, vals = self._rollout(train_batch, self.teacher_forcing, noise=True)
        T = self.time_slice
        l_pred, l_interp = self.criterion(pred, target), self.criterion(hr_hat, vals[:, :-T])
        loss = l_pred + l_interp
        self.log("train_loss", loss, prog_bar=True)
        same = getattr(self, "loss", None) == "l1"          # (the criterion IS the mean absolute error: logged once computed)
        self.log("train_mae_loss", l_pred.detach() if same else self.mae_criterion(pred, target), prog_bar=True)
        self.log("train_interp_loss", l_interp.detach() if same else self.mae_criterion(hr_hat, vals[:, :-T]), prog_bar=True)
        return loss

    @torch.no_grad()          # validation never differentiates (Lightning runs it under no_grad as well)
    def validation_step(self, val_batch, batch_idx):
        keys = ("t", "lr_frames", "hr_points", "coords_hr", "coords_lr")
        rows = val_batch["lr_frames"].shape[0] * (val_batch["lr_frames"].shape[-1] + val_batch["coords_hr"].shape[1])
        pred, target = self._graphed(lambda b: self._rollout(b)[:2], val_batch, keys, rows)
        val = self.criterion(pred, target)
        self.log("val_loss", val, prog_bar=True)
        # (loss = 'l1', every shipped config: the criterion IS the mean absolute error - logged once computed)
        self.log("val_mae_loss", val if getattr(self, "loss", None) == "l1" else self.mae_criterion(pred, target), prog_bar=True)
        return pred
