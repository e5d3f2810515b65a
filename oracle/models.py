"""torch-CPU restatement of the five hot-path models (TEST INFRASTRUCTURE, see oracle/__init__.py).

Module trees reproduce the reference's state-dict keys so the same
deterministic weights load into the imported reference, this oracle and the HIP models.
The arithmetic follows SURVEY.md Appendix A; every function cites the reference lines it
restates.  The op sequence is deliberately the reference's (gather -> concat -> Linear ->
activation -> ... -> scatter-mean -> norm; no fusion, no algebraic shortcuts): this is also
the CPU baseline timed by bench.py.
"""
from __future__ import annotations

import types

import torch
import torch.nn.functional as F
from torch import nn

from . import pyg_semantics as S


# ----------------------------------------------------------------------------- small pieces
def swish(x):
    """models/mpnn.py:23-24 (beta = 1)."""
    return x * torch.sigmoid(x)


def make_coord(shape, flatten=True):
    """utils.py:19-35: cell centres in [-1, 1], 'ij' meshgrid, row-major flatten."""
    seqs = []
    for n in shape:
        r = 1.0 / n
        seqs.append(-1 + r + (2 * r) * torch.arange(n).float())
    ret = torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)
    return ret.view(-1, ret.shape[-1]) if flatten else ret


class MLP(nn.Module):
    """models/backbones/mlp.py:9-28: Linear,ReLU,(Linear,ReLU)*(n-1),Linear; keys layers.{0,2,..}."""

    def __init__(self, in_dim, hidden_list, out_dim):
        super().__init__()
        dims = [in_dim] + list(hidden_list)
        mods = []
        for a, b in zip(dims[:-1], dims[1:]):
            mods += [nn.Linear(a, b), nn.ReLU()]
        mods.append(nn.Linear(dims[-1], out_dim))
        self.layers = nn.ModuleList(mods)

    def forward(self, x):
        for m in self.layers:
            x = m(x)
        return x


def _mlp_ln(i, hidden, layers, o):
    return nn.Sequential(MLP(i, [hidden] * layers, o), nn.LayerNorm(o))


class _ResBlock(nn.Module):
    def __init__(self, c, k, conv):
        super().__init__()
        self.conv_1 = conv(c, c, k, padding=k // 2)
        self.conv_2 = conv(c, c, k, padding=k // 2)


class EDSR(nn.Module):
    """models/backbones/edsr.py:32-60 (effective res_scale is always 1)."""

    def __init__(self, in_chan, n_chan, res_layers, kernel_size, mode):
        super().__init__()
        conv = nn.Conv2d if mode == "2d" else nn.Conv1d
        self.head_conv = conv(in_chan, n_chan, kernel_size, padding=kernel_size // 2)
        self.res_layers = nn.Sequential(*[_ResBlock(n_chan, kernel_size, conv) for _ in range(res_layers)])
        self.tail_conv = conv(n_chan, n_chan, kernel_size, padding=kernel_size // 2)

    def forward(self, x):
        x = self.head_conv(x)
        r = x
        for blk in self.res_layers:
            r = blk.conv_2(torch.relu(blk.conv_1(r))) + r
        return self.tail_conv(r) + x


# ----------------------------------------------------------------------------- MPNN (1D / 2D)
_READOUT = {10: (16, 6, 10), 16: (16, 5, 8), 20: (15, 4, 10), 25: (16, 3, 14), 50: (12, 2, 10)}


class _GNNLayer(nn.Module):
    def __init__(self, h, tw, dpos):
        super().__init__()
        self.message_net_1 = nn.Sequential(nn.Linear(2 * h + tw + dpos + 1, h))
        self.message_net_2 = nn.Sequential(nn.Linear(h, h))
        self.update_net_1 = nn.Sequential(nn.Linear(2 * h + 1, h))
        self.update_net_2 = nn.Sequential(nn.Linear(h, h))


class MPNN(nn.Module):
    """models/mpnn.py:93-333 (dim=1) and models/mpnn_2d.py:93-333 (dim=2)."""

    def __init__(self, hp, dim=1):
        super().__init__()
        self.dim, self.tw, self.h = dim, hp.time_window, hp.hidden_features
        self.n_layers, self.n = hp.hidden_layer, hp.neighbors
        self.dpos = 1 if dim == 1 else 2                     # Q4: 2D feeds (x/Lx, x/Ly)
        self.gnn_layers = nn.ModuleList([_GNNLayer(self.h, self.tw, self.dpos) for _ in range(self.n_layers)])
        self.embedding_mlp = nn.Sequential(nn.Linear(self.tw + self.dpos + 1, self.h), nn.Identity(),
                                           nn.Linear(self.h, self.h))
        k1, s1, k2 = _READOUT[self.tw]
        self.readout_swish = not (dim == 1 and self.tw == 10)  # Q6
        mods = [nn.Conv1d(1, 8, k1, stride=s1)] + ([nn.Identity()] if self.readout_swish else []) + [nn.Conv1d(8, 1, k2)]
        self.output_mlp = nn.Sequential(*mods)
        self.loss = hp.loss
        self.logged = {}

    # models/mpnn.py:175-203
    def forward(self, data, L, tmax, dt):
        u, pos, ei, batch = data.x, data.pos, data.edge_index, data.batch
        pos_x = pos[:, 1:2] / L.reshape(1, -1)               # mpnn.py:180 / mpnn_2d.py:179 (Q4)
        var = pos[:, 0:1] / tmax                             # mpnn.py:181,188
        h = swish(self.embedding_mlp[2](swish(self.embedding_mlp[0](torch.cat([u, pos_x, var], -1)))))
        src, tgt = ei[0], ei[1]
        for lyr in self.gnn_layers:                          # mpnn.py:193-194, GNN_Layer :65-90
            m_in = torch.cat([h[tgt], h[src], u[tgt] - u[src], pos_x[tgt] - pos_x[src], var[tgt]], -1)
            m = swish(lyr.message_net_2[0](swish(lyr.message_net_1[0](m_in))))
            agg = S.scatter_mean(m, tgt, h.shape[0])         # aggr='mean', mpnn.py:46
            upd = swish(lyr.update_net_2[0](swish(lyr.update_net_1[0](torch.cat([h, agg, var], -1)))))
            h = S.instance_norm(h + upd, batch)              # mpnn.py:70,87-88
        steps = dt * torch.arange(1, self.tw + 1, dtype=h.dtype)   # cumsum of tw copies of dt, mpnn.py:197-198
        z = self.output_mlp[0](h[:, None])
        if self.readout_swish:
            z = swish(z)
        diff = self.output_mlp[-1](z).squeeze(1)
        return u[:, -1:] + steps[None, :] * diff             # mpnn.py:201

    # models/mpnn.py:216-251 / models/mpnn_2d.py:215-251
    def _build_graph(self, data, t, x, steps):
        B, tw, N = data.shape
        u = data.permute(0, 2, 1).reshape(B * N, tw)
        x0 = x[0]                                            # Q8: sample 0's mesh for everybody
        x_pos = x0.repeat(B, *([1] * (x0.dim() - 1)))
        t_pos = torch.stack([t[b, steps[b]] for b in range(B)]).repeat_interleave(N)
        batch = torch.arange(B).repeat_interleave(N)
        if self.dim == 1:
            radius = self.n * (x0[1] - x0[0]) + 0.0001
            pos = torch.stack([t_pos, x_pos], 1)
        else:
            dx = x0[1] - x0[0]
            dy = x0[int(N ** 0.5)] - x0[0]
            radius = self.n * torch.norm(dx - dy, p=2) + 0.0001
            pos = torch.cat([t_pos[:, None], x_pos], 1)
        ei = S.radius_graph(x_pos, radius, batch, loop=False)
        return types.SimpleNamespace(x=u, pos=pos, edge_index=ei, batch=batch)

    def _step_index(self, i_next):
        """time index used for the 'variables' feature (1D freezes it at 0; mpnn.py:265 vs mpnn_2d.py:265,287)."""
        return 0 if self.dim == 1 else (i_next + 1) * self.tw - 1

    # models/mpnn.py:299-333
    def rollout(self, batch):
        u = batch["u"].float().permute(0, 2, 1)
        x = batch["x"].float()
        if self.dim == 1:
            x = x.squeeze(-1)
        t = batch["t"].float()
        B, _, N = u.shape
        dt = t[0][1] - t[0][0]
        graph = self._build_graph(u[:, : self.tw], t, x, [self._step_index(0)] * B)
        target = u[:, self.tw:]
        outs = []
        for i in range(target.shape[1] // self.tw):
            y = self.forward(graph, x[0, -1], t[0, -1], dt).reshape(B, N, -1).permute(0, 2, 1)
            outs.append(y)
            graph = self._build_graph(y, t, x, [self._step_index(i + 1)] * B)
        return torch.cat(outs, 1), target

    def validation_step(self, batch, batch_idx=0):
        u_hat, target = self.rollout(batch)
        self.logged["val_loss"] = _criterion(self.loss)(u_hat, target)
        self.logged["val_mae_loss"] = F.l1_loss(u_hat, target)
        return u_hat


def _criterion(name):
    return {"l1": F.l1_loss, "l2": F.mse_loss, "smooth_l1": F.smooth_l1_loss}[name]


# ----------------------------------------------------------------------------- MAgNet shared graph part
class _Encoder(nn.Module):
    def __init__(self, node_in, edge_in, latent, layers, hidden):
        super().__init__()
        self.node_fn = _mlp_ln(node_in, hidden, layers, latent)
        self.edge_fn = _mlp_ln(edge_in, hidden, layers, latent)


class _Interaction(nn.Module):
    def __init__(self, latent, layers, hidden):
        super().__init__()
        self.node_fn = _mlp_ln(2 * latent, hidden, layers, latent)
        self.edge_fn = _mlp_ln(3 * latent, hidden, layers, latent)


class _Processor(nn.Module):
    def __init__(self, latent, steps, layers, hidden):
        super().__init__()
        self.gnn_stacks = nn.ModuleList([_Interaction(latent, layers, hidden) for _ in range(steps)])


class _Decoder(nn.Module):
    def __init__(self, latent, out, layers, hidden):
        super().__init__()
        self.node_fn = MLP(latent, [hidden] * layers, out)


def magnet_build_graph(u, x, t, radius):
    """models/magnet_gnn.py:286-310 (copies magnet_cnn.py:296-320, magnet_cnn_2d.py:303-327)."""
    B, N, _ = u.shape
    u_, x_ = u.reshape(B * N, -1), x.reshape(B * N, -1)
    batch = torch.arange(B).repeat_interleave(N)
    edges = S.radius_graph(x_, radius, batch, loop=True)
    receivers, senders = edges[0], edges[1]                  # Q11: aggregate at the *found* node
    node = torch.cat([u_, x_, t[:, -1:].repeat(N, 1)], -1)   # Q5 tile
    edge = torch.cat([u_[senders] - u_[receivers], x_[senders] - x_[receivers]], -1)
    return node, senders, receivers, edge


def magnet_graph_net(enc, proc, node, senders, receivers, edge):
    """Encoder (magnet_gnn.py:38-42) + K InteractionNetwork hops (magnet_gnn.py:70-90, Q2)."""
    x, e = enc.node_fn(node), enc.edge_fn(edge)
    for k, g in enumerate(proc.gnn_stacks):
        m = g.edge_fn(torch.cat([x[receivers], x[senders], e], -1))   # x_i = target = receivers
        agg = S.scatter_mean(m, receivers, x.shape[0])
        x = g.node_fn(torch.cat([agg, x], -1)) + x
        e = e + e                                            # Q2: edge latents only double
    return x


def _euler(ret, t, T, last):
    """magnet_gnn.py:359-369: out[b,k,n] = last[b,n] + (t[b,T+k]-t[b,T-1]) * ret[b,n,k]."""
    delta = (t[:, T:] - t[:, T - 1:T])                       # [B, T_out]
    return last[:, None, :, :] + delta[:, :, None, None] * ret.permute(0, 2, 1)[..., None]


# ----------------------------------------------------------------------------- MAgNet[GNN]
class MAgNetGNN(nn.Module):
    """models/magnet_gnn.py:139-475.  ``coord_dim`` (default 2 = the reference, Q7) parametrises
    the hard-wired +3/+2/+4 widths so the 1D configs have a runnable counterpart."""

    def __init__(self, hp):
        super().__init__()
        d = getattr(hp, "coord_dim", 2)
        T, D, Lm, Hm = hp.time_slice, hp.latent_dim, hp.mlp_layers, hp.mlp_hidden
        self.T, self.k, self.radius, self.interp, self.loss = T, hp.codec_neighbors, hp.radius, hp.interpolation, hp.loss
        self.encoder = _Encoder(T + d + 1, T + d, D, Lm, Hm)
        self.processor = _Processor(D, hp.num_message_passing_steps, Lm, Hm)
        self.proj_head = nn.Linear(D + d + 2, hp.n_chan)
        self.projector = MLP(hp.n_chan, [Hm] * Lm, 1)
        self._encoder = _Encoder(T + d + 1, T + d, D, Lm, Hm)
        self._processor = _Processor(D, hp.num_message_passing_steps, Lm, Hm)
        self._decoder = _Decoder(D, T, Lm, Hm)
        self.logged = {}

    # magnet_gnn.py:224-283
    def continuous_decoder(self, x_lr, enc, lr_coords, hr_coords, t):
        B, T, _, L = x_lr.shape
        Nq = hr_coords.shape[1]
        flr, fhr = lr_coords.reshape(B * L, -1), hr_coords.reshape(B * Nq, -1)
        bl, bh = torch.arange(B).repeat_interleave(L), torch.arange(B).repeat_interleave(Nq)
        nn_idx = S.knn(flr, fhr, self.k, bl, bh)[1].view(B * Nq, self.k)
        vals = x_lr[:, :, 0, :].permute(0, 2, 1).reshape(B * L, T)      # value of node at each input step
        tq = t[:, :T].repeat_interleave(Nq, 0)                           # [B*Nq, T]
        lat, w = [], []
        for j in range(2):                                   # only the two nearest are used (Q9)
            nb = nn_idx[:, j]
            rel = flr[nb] - fhr
            inp = torch.cat([enc[nb][:, None, :].expand(-1, T, -1), vals[nb][:, :, None],
                             rel[:, None, :].expand(-1, T, -1), tq[:, :, None]], -1)
            lat.append(self.proj_head(inp))
            n2 = torch.norm(rel, 2, dim=-1) ** 2
            w.append({"area": n2, "knn": 1 / n2, "sph": torch.pow(1 - L * n2, 3)}[self.interp][:, None, None])
        if self.interp == "area":
            return (lat[0] * w[1] + lat[1] * w[0]) / (w[1] + w[0])
        return (lat[0] * w[0] + lat[1] * w[1]) / (w[1] + w[0])

    # magnet_gnn.py:312-376
    def forward(self, x_lr, lr_coords, hr_coords, t, hr_last):
        B, T, C, L = x_lr.shape
        Nq = hr_coords.shape[1]
        u = x_lr.permute(0, 3, 1, 2).reshape(B, L, -1)
        enc = magnet_graph_net(self.encoder, self.processor, *magnet_build_graph(u, lr_coords, t[:, :T], self.radius))
        z = self.continuous_decoder(x_lr, enc, lr_coords, hr_coords, t)
        hr_points = self.projector(z).reshape(B, Nq, T)
        feats = torch.cat([u, hr_points], 1)
        coords = torch.cat([lr_coords, hr_coords], 1)
        x = magnet_graph_net(self._encoder, self._processor, *magnet_build_graph(feats, coords, t[:, :T], self.radius))
        ret = self._decoder.node_fn(x).reshape(B, L + Nq, -1)
        last = torch.cat([x_lr[:, -1].permute(0, 2, 1), hr_last], 1)
        out = _euler(ret, t, T, last)
        return out[:, :, L:], out[:, :, :L], hr_points.permute(0, 2, 1)[..., None]

    # magnet_gnn.py:442-475
    def validation_step(self, b, batch_idx=0):
        t, u, vals = b["t"].float(), b["lr_frames"].float(), b["hr_points"].float()
        hr_c, lr_c = b["coords_hr"].float(), b["coords_lr"].float()
        T = self.T
        fut = vals[:, T:]
        inp, hr_last, outs = u[:, :T], vals[:, T - 1], []
        for i in range(fut.shape[1] // T):
            out_hr, out_lr, _ = self.forward(inp, lr_c, hr_c, t[:, i * T:(i + 2) * T], hr_last)
            outs.append(torch.cat([out_hr, out_lr], 2))
            inp, hr_last = out_lr.permute(0, 1, 3, 2), out_hr[:, -1]
        pred = torch.cat(outs, 1)
        target = torch.cat([fut, u[:, T:].permute(0, 1, 3, 2)], 2)
        self.logged["val_loss"] = _criterion(self.loss)(pred, target)
        self.logged["val_mae_loss"] = F.l1_loss(pred, target)
        return pred


# ----------------------------------------------------------------------------- MAgNet[CNN] 1D / 2D
def _nearest(src, grid):
    return F.grid_sample(src, grid, mode="nearest", padding_mode="border", align_corners=False)


class MAgNetCNN(nn.Module):
    """models/magnet_cnn.py:142-467 (dim=1) and models/magnet_cnn_2d.py:142-479 (dim=2)."""

    def __init__(self, hp, dim=1):
        super().__init__()
        d = self.dim = dim
        T, D, Lm, Hm, C = hp.time_slice, hp.latent_dim, hp.mlp_layers, hp.mlp_hidden, hp.n_chan
        self.T, self.radius, self.loss = T, hp.radius, hp.loss
        self.encoder = EDSR(T, C, hp.res_layers, hp.kernel_size, "2d" if d == 2 else "1d")
        self.proj_head = _mlp_ln(C + 2 * d + 2, Hm, Lm, C)   # 1D: C+3+1, 2D: C+5+1
        self.projector = MLP(C, [Hm] * Lm, 1)
        self._encoder = _Encoder(T + d + 1, T + d, D, Lm, Hm)
        self._processor = _Processor(D, hp.num_message_passing_steps, Lm, Hm)
        self._decoder = _Decoder(D, T, Lm, Hm)
        self.logged = {}

    # magnet_cnn.py:225-285 / magnet_cnn_2d.py:225-292
    def continuous_decoder(self, x_t, feat, cell, coord_hr, t):
        B, T = x_t.shape[:2]
        W = feat.shape[-1]
        Nq = coord_hr.shape[1]
        tq = t[:, :T][:, None, :].expand(B, Nq, T)
        preds, areas = [], []
        if self.dim == 2:
            fc = make_coord([W, W], flatten=False).permute(2, 0, 1)[None].expand(B, 2, W, W)
            shifts = [(vx, vy) for vx in (-1, 1) for vy in (-1, 1)]
        else:
            c1 = make_coord([W], flatten=False).permute(1, 0)[None].expand(B, 1, W)
            fc = torch.cat([c1, torch.zeros_like(c1)], 1)[:, :, None, :]
            shifts = [(-1,), (1,)]
        for sh in shifts:
            if self.dim == 2:
                c = coord_hr.clone()
                c[..., 0] += sh[0] / W + 1e-6
                c[..., 1] += sh[1] / W + 1e-6
                c.clamp_(-1 + 1e-6, 1 - 1e-6)
                grid = c.flip(-1)[:, None]
                src_feat = feat
            else:
                c = torch.cat([coord_hr, torch.zeros_like(coord_hr)], -1)[:, None]
                c[..., 0] += sh[0] / W + 1e-6
                c.clamp_(-1 + 1e-6, 1 - 1e-6)
                grid = c
                src_feat = feat[:, :, None, :]
            q_feat = _nearest(src_feat, grid)[:, :, 0].permute(0, 2, 1)            # [B, Nq, C]
            q_coord = _nearest(fc, grid)[:, :, 0].permute(0, 2, 1)[..., : self.dim]
            rel = (coord_hr - q_coord) * W
            cellw = cell * W
            xin = x_t[:, :, 0]                                                      # [B, T, (W,) W]
            q_inp = _nearest(xin if self.dim == 2 else xin[:, :, None, :], grid)[:, :, 0].permute(0, 2, 1)  # [B, Nq, T]
            inp = torch.cat([q_feat[:, :, None, :].expand(-1, -1, T, -1), q_inp[..., None],
                             rel[:, :, None, :].expand(-1, -1, T, -1), cellw[:, :, None, :].expand(-1, -1, T, -1),
                             tq[..., None]], -1).reshape(B * Nq, T, -1)
            preds.append(self.proj_head(inp))
            if self.dim == 2:
                areas.append((rel[..., 0] * rel[..., 1]).abs().reshape(B * Nq, 1, 1) + 1e-9)
            else:
                areas.append(rel.abs().reshape(B * Nq, 1, 1))
        if self.dim == 2:
            tot = areas[0] + areas[1] + areas[2] + areas[3]
            sw = [areas[3], areas[2], areas[1], areas[0]]     # diagonally opposite cell's area
            ret = 0
            for p, a in zip(preds, sw):
                ret = ret + p * (a / tot)
            return ret
        return (preds[0] * areas[1] + preds[1] * areas[0]) / (areas[1] + areas[0])

    # magnet_cnn.py:322-381 / magnet_cnn_2d.py:329-390
    def forward(self, x_t, coords, cell, t, hr_last, hiddens=None):
        B, T = x_t.shape[:2]
        W = x_t.shape[-1]
        Nq = coords.shape[1]
        nlr = W * W if self.dim == 2 else W
        feat = self.encoder(x_t.reshape(B, T, *x_t.shape[3:]))
        z = self.continuous_decoder(x_t, feat, cell, coords, t)
        hr_points = self.projector(z).reshape(B, Nq, T)
        lr_points = x_t[:, :, 0].reshape(B, T, nlr).permute(0, 2, 1)
        lr_coords = make_coord([W] * self.dim)[None].expand(B, -1, -1)
        feats = torch.cat([lr_points, hr_points], 1)
        allc = torch.cat([lr_coords, coords], 1)
        x = magnet_graph_net(self._encoder, self._processor, *magnet_build_graph(feats, allc, t[:, :T], self.radius))
        ret = self._decoder.node_fn(x).reshape(B, nlr + Nq, -1)
        last = torch.cat([x_t[:, -1, 0].reshape(B, nlr, 1), hr_last], 1)
        out = _euler(ret, t, T, last)
        out_lr = out[:, :, :nlr]
        if self.dim == 2:
            out_lr = out_lr.permute(0, 1, 3, 2).reshape(B, -1, 1, W, W)
        return out[:, :, nlr:], out_lr, hr_points.permute(0, 2, 1)[..., None]

    # magnet_cnn.py:438-467 / magnet_cnn_2d.py:447-479
    def validation_step(self, b, batch_idx=0):
        t, u, vals = b["t"].float(), b["lr_frames"].float(), b["hr_points"].float()
        coords, cells = b["coords"].float(), b["cells"].float()
        T, W = self.T, u.shape[-1]
        B = u.shape[0]
        fut = vals[:, T:]
        inp, hr_last, outs = u[:, :T], vals[:, T - 1], []
        for i in range(fut.shape[1] // T):
            y, _, _ = self.forward(inp, coords, cells, t[:, i * T:(i + 2) * T], hr_last)
            outs.append(y)
            img = y.permute(0, 1, 3, 2)                       # [B, T, 1, Nq]
            if self.dim == 2:
                wi = int(img.shape[-1] ** 0.5)
                inp = F.interpolate(img.reshape(-1, 1, wi, wi), size=W, mode="bilinear",
                                    align_corners=False).reshape(B, -1, 1, W, W)
            else:
                inp = F.interpolate(img.reshape(-1, 1, img.shape[-1]), size=W, mode="linear",
                                    align_corners=False).reshape(B, -1, 1, W)
            hr_last = y[:, -1]
        pred = torch.cat(outs, 1)
        self.logged["val_loss"] = _criterion(self.loss)(pred, fut)
        self.logged["val_mae_loss"] = F.l1_loss(pred, fut)
        return pred


def build(name, hp):
    """name -> oracle model, mirroring models/factory.py:10-19 for the hot-path entries."""
    return {"mpnn": lambda: MPNN(hp, 1), "mpnn_2d": lambda: MPNN(hp, 2), "magnet_gnn": lambda: MAgNetGNN(hp),
            "magnet_cnn": lambda: MAgNetCNN(hp, 1), "magnet_cnn_2d": lambda: MAgNetCNN(hp, 2)}[name]()
