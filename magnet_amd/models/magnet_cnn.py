"""MAgNet[CNN] on MI355X - drop-in for the reference's ``models/magnet_cnn.py`` (``MAgNetCNN``, 1-D)
and, through ``DIM = 2``, ``models/magnet_cnn_2d.py`` (``MAgNetCNN_2d``).

forward(x_t, coords, cell, t, hr_last, hiddens=None) -> (out_hr, out_lr, hr_points).

  EDSR feature map                       -> magnet_mlp_* as gathered-tap implicit GEMMs (backbones/edsr.py)
  shifted nearest-cell lookups (LIIF)    -> magnet_inr_grid_prepare_f32         magnet_cnn_2d.py:248-283
  proj_head MLP + LayerNorm per shift    -> magnet_mlp_f32 (row gather of the channels-last feature map)
  area blend with the opposite cell      -> magnet_inr_grid_blend_f32           magnet_cnn_2d.py:285-290
  projector, joint graph, Euler update   -> MagnetBase
"""
from __future__ import annotations

import types

import torch

from ..graphs import capturing

from .. import _lib as L
from .. import ops
from ..packing import pad_to, param_list
from ..utils import make_coord
from .backbones.edsr import EDSR
from .backbones.mlp import MLP
from .magnet_common import (Decoder, Encoder, MagnetBase, Processor, _mlp_ln, pack_graph_net, pack_mlp, pack_mlp_ln,
                            run_mlp)


class MAgNetCNN(MagnetBase):
    DIM = 1
    TOPOLOGY_KEYS = ("coords",)

    def __init__(self, hparams):
        super().__init__()
        self.save_hyperparameters()
        self._init_common(hparams)
        d = self.DIM
        self.time_slice = hparams.time_slice
        self.num_message_passing_steps = hparams.num_message_passing_steps
        self.latent_dim, self.mlp_layers, self.mlp_hidden = hparams.latent_dim, hparams.mlp_layers, hparams.mlp_hidden
        self.scales, self.res_layers, self.n_chan = hparams.scales, hparams.res_layers, hparams.n_chan
        self.kernel_size, self.res_scale = hparams.kernel_size, hparams.res_scale
        self.interpolation, self.radius, self.teacher_forcing = hparams.interpolation, hparams.radius, hparams.teacher_forcing
        # A/B knob: the 2-D area blend inside the decoder chain's epilogue (EPI_QUADBLEND) instead of a blend pass over materialised predictions
        self.fused_blend = bool(getattr(hparams, "fused_blend", __import__("os").environ.get("MAGNET_FUSED_BLEND", "1") != "0"))
        if not 1 <= self.mlp_layers <= 5:        # 4 (every shipped config): the fused 5-Linear kernel; other depths run layer by layer
            raise NotImplementedError("mlp_layers must be between 1 and 5")
        T, D, Lm, Hm, C, K = self.time_slice, self.latent_dim, self.mlp_layers, self.mlp_hidden, self.n_chan, self.num_message_passing_steps
        self.encoder = EDSR(in_chan=T, n_chan=C, res_layers=self.res_layers, kernel_size=self.kernel_size,
                            res_scale=self.res_scale, mode="2d" if d == 2 else "1d")
        self.proj_head = _mlp_ln(C + 2 * d + 2, Hm, Lm, C)       # [feat | q_inp | rel | cell | t]
        self.projector = MLP(C, [Hm] * Lm, 1)
        self._encoder = Encoder(T + d + 1, D, T + d, D, Lm, Hm)
        self._processor = Processor(D, D, D, D, K, Lm, Hm)
        self._decoder = Decoder(D, T, Lm, Hm)

    def _pack_weights(self):
        T, d, C = self.time_slice, self.DIM, self.n_chan
        return types.SimpleNamespace(
            proj_head=pack_mlp_ln(self.proj_head, k_pad=C + 32),
            projector=pack_mlp(self.projector),
            graph2=pack_graph_net(self._encoder, self._processor, pad_to(T + d + 1)),
            decoder=pack_mlp(self._decoder.node_fn))

    def feature_encoding(self, x_t):
        """EDSR on [B, T*C, W(, W)] (magnet_cnn_2d.py:294-301)."""
        B, T = x_t.shape[:2]
        return self.encoder(x_t.reshape(B, T * x_t.shape[2], *x_t.shape[3:]))

    def forward(self, x_t, coords, cell, t, hr_last, hiddens=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in param_list(self)):
            return self._forward_grad(x_t, coords, cell, t, hr_last)
        with torch.no_grad():
            return self._forward(x_t, coords, cell, t, hr_last)

    def _forward_grad(self, x_t, coords, cell, t, hr_last):
        """``_forward`` built from magnet_amd.autograd Functions (HIP forward and backward kernels); torch autograd chains
        them and differentiates the layout glue.  The nearest-cell value q_inp, which the inference path copies into column
        0 of the per-row tail, is gathered here as its own (differentiable) source summed into the same k tile."""
        from .. import autograd as AG
        d, C = self.DIM, self.n_chan
        B, T = x_t.shape[:2]
        W = x_t.shape[-1]
        nlr = W ** d
        Nq = coords.shape[1]
        S = 2 ** d
        dev = x_t.device
        pk = self._packed()
        t = ops._f32(t)
        x_t = x_t.float()
        coords = coords.reshape(B, Nq, d).float()
        featT = self.encoder.rows_grad(x_t.reshape(B, T * x_t.shape[2], *x_t.shape[3:]))
        rows = B * Nq * S
        with torch.no_grad():
            cell_idx = torch.empty(rows * T, dtype=torch.int32, device=dev)
            tail = torch.empty(rows * T, 32, device=dev, dtype=torch.float32)
            area = torch.empty(rows, device=dev, dtype=torch.float32)
            fco, fce = ops._f32(coords.reshape(B * Nq, d)), ops._f32(cell.reshape(B * Nq, d))
            L.check(L.lib().magnet_inr_grid_prepare_f32(ops._p(fco), ops._p(fce), ops._p(ops._f32(x_t.detach())), ops._p(t), t.shape[1],
                                                        B, Nq, T, W, d, ops._p(cell_idx), ops._p(tail), ops._p(area), ops._stream()),
                    "inr_grid_prepare")
            tail[:, 0] = 0                                                   # q_inp comes in through its own source below
            r = torch.arange(rows * T, device=dev)
            b_of, i_of = (r // T) // (Nq * S), r % T
            flat = cell_idx.long() - b_of * nlr
            idx_q = ((b_of * T + i_of) * nlr + flat).int()                     # row of x_t[b, i, cell] in [B*T*cells]
        xq = torch.cat([x_t.reshape(B * T * nlr, 1), torch.zeros(B * T * nlr, 31, device=dev)], 1)
        ph = pk.proj_head
        p = AG.mlp(rows * T, [ops.Term(featT, C, idx=cell_idx), (ops.Term(xq, 32, idx=idx_q), ops.Term(tail, 32))], ph, ph.params)
        z = AG.grid_blend(p, area, B * Nq, T, C, S)
        pj = pk.projector
        hr_points = AG.mlp(B * Nq * T, [ops.Term(z, C)], pj, pj.params).reshape(B, Nq, T)
        lr_points = x_t[:, :, 0].reshape(B, T, nlr).permute(0, 2, 1)
        lr_coords = make_coord([W] * d, device=dev)[None].expand(B, -1, -1)
        feats = torch.cat([lr_points, hr_points], dim=1)
        allc = torch.cat([lr_coords, coords], dim=1)
        last = torch.cat([x_t[:, -1, 0].reshape(B, nlr, 1), hr_last.float()], dim=1)
        out = self._interaction_and_update_grad(pk, feats, allc, t, T, last, key=self._tkey(("all", W), coords))
        out_lr = out[:, :, :nlr]
        if d == 2:
            out_lr = out_lr.permute(0, 1, 3, 2).reshape(B, -1, 1, W, W)
        return out[:, :, nlr:], out_lr, hr_points.permute(0, 2, 1)[..., None]

    def _forward(self, x_t, coords, cell, t, hr_last, hiddens=None):
        d, C = self.DIM, self.n_chan
        B, T = x_t.shape[:2]
        W = x_t.shape[-1]
        nlr = W ** d
        Nq = coords.shape[1]
        S = 2 ** d
        dev = x_t.device
        pk = self._packed()
        t = ops._f32(t)
        x_t = ops._f32(x_t)
        coords = coords.reshape(B, Nq, d).float()
        featT = self.encoder.rows(x_t.reshape(B, T * x_t.shape[2], *x_t.shape[3:]))   # channels-last rows [B * W^d, C]
        # continuous_decoder
        rows = B * Nq * S
        cell_idx = torch.empty(rows * T, dtype=torch.int32, device=dev)
        tail = torch.empty(rows * T, 32, device=dev, dtype=torch.float32)
        fco, fce = ops._f32(coords.reshape(B * Nq, d)), ops._f32(cell.reshape(B * Nq, d))
        ph = pk.proj_head
        if (d == 2 and self.fused_blend and len(ph.layers) == 5 and ph.ht == 2 and ph.ot == 4 and ph.ln is not None
                and ph.layers[0][0].dtype != torch.float32):
            # the area blend in the decoder chain's epilogue (EPI_QUADBLEND): the 4 shifted predictions of a (query, step) are adjacent
            # rows, weighted and summed across the quad before anything is stored - p [4 rows][C] is neither written nor re-read
            bw = torch.empty(rows * T, device=dev, dtype=torch.float32)
            L.check(L.lib().magnet_inr_grid_prepare_blend_f32(ops._p(fco), ops._p(fce), ops._p(x_t), ops._p(t), t.shape[1], B, Nq, T, W,
                                                              ops._p(cell_idx), ops._p(tail), ops._p(bw), ops._stream()),
                    "inr_grid_prepare_blend")
            z = torch.empty(B * Nq * T, C, device=dev, dtype=torch.float32)
            run_mlp(rows * T, ops.ktiles(ops.Term(featT, C, idx=cell_idx), ops.Term(tail, 32)), ph, z, epi=ops.EPI_QUADBLEND, res=bw)
            return self._finish_forward(pk, z, x_t, coords, t, T, hr_last, B, Nq, W, nlr, d, dev)
        area = torch.empty(rows, device=dev, dtype=torch.float32)
        L.check(L.lib().magnet_inr_grid_prepare_f32(ops._p(fco), ops._p(fce), ops._p(x_t), ops._p(t), t.shape[1], B, Nq, T, W,
                                                    d, ops._p(cell_idx), ops._p(tail), ops._p(area), ops._stream()),
                "inr_grid_prepare")
        p = torch.empty(rows * T, C, device=dev, dtype=torch.float32)
        run_mlp(rows * T, ops.ktiles(ops.Term(featT, C, idx=cell_idx), ops.Term(tail, 32)), pk.proj_head, p)
        z = torch.empty(B * Nq * T, C, device=dev, dtype=torch.float32)
        L.check(L.lib().magnet_inr_grid_blend_f32(ops._p(p), ops._p(area), B * Nq, T, C, S, ops._p(z), ops._stream()),
                "inr_grid_blend")
        return self._finish_forward(pk, z, x_t, coords, t, T, hr_last, B, Nq, W, nlr, d, dev)

    def _finish_forward(self, pk, z, x_t, coords, t, T, hr_last, B, Nq, W, nlr, d, dev):
        """projector + joint graph of LR grid nodes and queries (the part of ``_forward`` after the INR blend)"""
        hr_points = self._projector(pk, z, B * Nq * T).reshape(B, Nq, T)
        # joint graph: LR grid nodes (cell centres, row-major) + queries
        lr_points = x_t[:, :, 0].reshape(B, T, nlr).permute(0, 2, 1)
        lr_coords = make_coord([W] * d, device=dev)[None].expand(B, -1, -1)
        feats = torch.cat([lr_points, hr_points], dim=1)
        allc = torch.cat([lr_coords, coords], dim=1)
        last = torch.cat([x_t[:, -1, 0].reshape(B, nlr, 1), hr_last.float()], dim=1)
        out = self._interaction_and_update(pk, feats, allc, t, T, last, key=self._tkey(("all", W), coords))
        out_lr = out[:, :, :nlr]
        if d == 2:
            out_lr = out_lr.permute(0, 1, 3, 2).reshape(B, -1, 1, W, W)
        return out[:, :, nlr:], out_lr, hr_points.permute(0, 2, 1)[..., None]

    def _lr_as_nodes(self, frames):
        """[B, T, 1, W(, W)] -> [B, T, W^d, 1] (the layout out_lr is compared in)."""
        B, T = frames.shape[:2]
        return frames.reshape(B, T, 1, -1).permute(0, 1, 3, 2)

    def training_step(self, train_batch, batch_idx):
        t, u = train_batch["t"].float(), train_batch["lr_frames"].float()
        vals, coords, cells = train_batch["hr_points"].float(), train_batch["coords"].float(), train_batch["cells"].float()
        T = self.time_slice
        fut = vals[:, T:]
        inp, hr_last, outs, hrs = u[:, :T], vals[:, T - 1], [], []
        with self._rollout_scope():
            for i in range(fut.shape[1] // T):
                out_hr, out_lr, hr_points = self.forward(inp, coords, cells, t[:, i * T:(i + 2) * T], hr_last)
                outs.append(torch.cat([out_hr, self._lr_as_nodes(out_lr) if self.DIM == 2 else out_lr], dim=2))
                hrs.append(hr_points)
                if self.teacher_forcing:
                    inp, hr_last = u[:, (i + 1) * T:(i + 2) * T], vals[:, (i + 2) * T - 1]
                else:
                    inp = out_lr if self.DIM == 2 else out_lr.permute(0, 1, 3, 2)
                    hr_last = out_hr[:, -1]
        pred, hr_hat = torch.cat(outs, dim=1), torch.cat(hrs, dim=1)
        target = torch.cat([fut, self._lr_as_nodes(u[:, T:])], dim=2)
        why = None if capturing() else self._range_trip()      # (a captured step - graphs.TrainStepGraph - reads the words behind its replay)
        if why:                                 # out of the fp16 pieces' range: the same step in the full-range split
            self._widen(why)
            return self.training_step(train_batch, batch_idx)
        l_pred, l_interp = self.criterion(pred, target), self.criterion(hr_hat, vals[:, :-T])
        loss = l_pred + l_interp
        self.log("train_loss", loss, prog_bar=True)
        same = getattr(self, "loss", None) == "l1"          # (the criterion IS the mean absolute error: logged once computed)
        self.log("train_mae_loss", l_pred.detach() if same else self.mae_criterion(pred, target), prog_bar=True)
        self.log("train_interp_loss", l_interp.detach() if same else self.mae_criterion(hr_hat, vals[:, :-T]), prog_bar=True)
        return loss

    def _inference_rollout(self, val_batch):
        """Free-running rollout; the next LR input is the predicted HR field down-sampled with
        (bi)linear interpolation (magnet_cnn.py:456-461, magnet_cnn_2d.py:466-472).  -> (pred, fut)"""
        t, u = val_batch["t"].float(), val_batch["lr_frames"].float()
        vals, coords, cells = val_batch["hr_points"].float(), val_batch["coords"].float(), val_batch["cells"].float()
        T, W, B = self.time_slice, u.shape[-1], u.shape[0]
        fut = vals[:, T:]
        inp, hr_last, outs = u[:, :T], vals[:, T - 1], []
        for i in range(fut.shape[1] // T):
            y, _, _ = self.forward(inp, coords, cells, t[:, i * T:(i + 2) * T], hr_last)
            outs.append(y)
            img = y.permute(0, 1, 3, 2)
            if self.DIM == 2:
                wi = int(img.shape[-1] ** 0.5)
                inp = ops.resize_linear(img.reshape(-1, wi, wi), (W, W), 2).reshape(B, -1, 1, W, W)
            else:
                inp = ops.resize_linear(img.reshape(-1, img.shape[-1]), (W,), 1).reshape(B, -1, 1, W)
            hr_last = y[:, -1]
        return torch.cat(outs, dim=1), fut

    @torch.no_grad()          # validation never differentiates (Lightning runs it under no_grad as well)
    def validation_step(self, val_batch, batch_idx):
        keys = ("t", "lr_frames", "hr_points", "coords", "cells")
        u = val_batch["lr_frames"]
        rows = u.shape[0] * (u.shape[-1] ** self.DIM + val_batch["coords"].shape[1])
        pred, fut = self._graphed(self._inference_rollout, val_batch, keys, rows)
        val = self.criterion(pred, fut)
        self.log("val_loss", val, prog_bar=True)
        # (loss = 'l1', every shipped config: the criterion IS the mean absolute error - logged once computed)
        self.log("val_mae_loss", val if getattr(self, "loss", None) == "l1" else self.mae_criterion(pred, fut), prog_bar=True)
        return pred
