"""Shared helpers for the parity tests."""
import json
import os
import types

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["mpnn_1d", "mpnn_2d", "magnet_gnn_full", "magnet_gnn_tiny", "magnet_cnn_1d", "magnet_cnn_2d",
         # round 2: shipped hyper-parameters on reduced grids (tests/golden/make_golden.py round2)
         "magnet_cnn_1d_shipped", "magnet_cnn_2d_shipped", "magnet_gnn_c5_reduced"]


def load_case(tag):
    d = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    hp = types.SimpleNamespace(**json.loads(str(d["hparams"])))
    batch = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("in_")}
    return str(d["name"]), hp, int(d["salt"]), batch, d


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def forward_args(name, hp, batch):
    """the arguments the golden generator passed to ``forward`` (single call)."""
    if name.startswith("mpnn"):
        raise ValueError
    T = hp.time_slice
    if name == "magnet_gnn":
        return (batch["lr_frames"][:, :T], batch["coords_lr"], batch["coords_hr"], batch["t"][:, :2 * T],
                batch["hr_points"][:, T - 1])
    return (batch["lr_frames"][:, :T], batch["coords"], batch["cells"], batch["t"][:, :2 * T],
            batch["hr_points"][:, T - 1])


def oracle_grads(oracle, batch, teacher_forcing, loss_name, dtype=torch.float64):
    """loss + gradients by torch autograd through the oracle's rollout (models/mpnn.py:254-297), on the CPU,
    arithmetic in ``dtype`` (fp64 gives a clean target; fp32 is what the reference runs)."""
    from oracle.models import _criterion
    o = oracle.to(dtype)
    f32 = {k: v.float() for k, v in batch.items()}
    u = f32["u"].permute(0, 2, 1)
    x = f32["x"] if o.dim == 2 else f32["x"].squeeze(-1)
    t = f32["t"]
    B, _, N = u.shape
    tw = o.tw
    dt = (t[0][1] - t[0][0]).to(dtype)

    def graph_of(window, step):
        # topology / positions from the fp32 inputs (what the reference sees), features in fp64 with their graph
        g = o._build_graph(window.detach().float(), t, x, [step] * B)
        g.x = window.to(dtype).permute(0, 2, 1).reshape(B * N, tw)
        g.pos = g.pos.to(dtype)
        return g

    graph = graph_of(u[:, :tw], o._step_index(0))
    target = u[:, tw:].to(dtype)
    outs = []
    for i in range(target.shape[1] // tw):
        y = o.forward(graph, x[0, -1].to(dtype), t[0, -1].to(dtype), dt).reshape(B, N, -1).permute(0, 2, 1)
        outs.append(y)
        nxt = u[:, (i + 1) * tw:(i + 2) * tw] if teacher_forcing else y
        graph = graph_of(nxt, o._step_index(i + 1))
    loss = _criterion(loss_name)(torch.cat(outs, 1), target)
    o.zero_grad()
    loss.backward()
    return float(loss.detach()), {n: p.grad.clone() for n, p in o.named_parameters()}


def load_grads(tag, teacher_forcing):
    """tests/golden/<tag>_grad.npz -> (loss, {param name: (norm, strided sample, stride)})"""
    d = np.load(os.path.join(GOLDEN, tag + "_grad.npz"), allow_pickle=False)
    key = "tf" if teacher_forcing else "free"
    stride = int(d["stride"])
    names = [k.split(":", 1)[1] for k in d.files if k.startswith(key + "_norm:")]
    return float(d[key + "_loss"]), {n: (float(d[f"{key}_norm:{n}"]), torch.from_numpy(d[f"{key}_sample:{n}"]), stride)
                                     for n in names}


def oracle_magnet_gnn_grads(oracle, batch, teacher_forcing, loss_name, dtype=torch.float64, noise=0.0):
    """loss + parameter gradients of the reference's MAgNetGNN.training_step (models/magnet_gnn.py:388-440) by
    torch autograd through the oracle restatement; ``noise`` > 0 draws the reference's host-side randn in its order."""
    from oracle.models import _criterion
    o = oracle.to(dtype)
    b = {k: v.to(dtype) for k, v in batch.items()}
    t, u, vals, hr_c, lr_c = b["t"], b["lr_frames"], b["hr_points"], b["coords_hr"], b["coords_lr"]
    T = o.T
    fut = vals[:, T:]
    noisy = (lambda x: x + (noise * torch.randn(x.shape)).to(dtype)) if noise else (lambda x: x)   # magnet_gnn.py:402-407
    inp, outs, hrs = noisy(u[:, :T]), [], []
    hr_last = noisy(vals[:, T - 1])
    for i in range(fut.shape[1] // T):
        out_hr, out_lr, hr_points = o.forward(inp, lr_c, hr_c, t[:, i * T:(i + 2) * T], hr_last)
        outs.append(torch.cat([out_hr, out_lr], 2))
        hrs.append(hr_points)
        if teacher_forcing:
            inp, hr_last = u[:, (i + 1) * T:(i + 2) * T], vals[:, (i + 2) * T - 1]
        else:
            inp, hr_last = out_lr.permute(0, 1, 3, 2), out_hr[:, -1]
        inp = noisy(inp)
        hr_last = noisy(hr_last)
    pred, hr_hat = torch.cat(outs, 1), torch.cat(hrs, 1)
    target = torch.cat([fut, u[:, T:].permute(0, 1, 3, 2)], 2)
    crit = _criterion(loss_name)
    loss = crit(pred, target) + crit(hr_hat, vals[:, :-T])
    o.zero_grad()
    loss.backward()
    return float(loss.detach()), {n: p.grad.clone() for n, p in o.named_parameters()}


def oracle_magnet_cnn_grads(oracle, batch, teacher_forcing, loss_name, dtype=torch.float64):
    """loss + parameter gradients of the reference's MAgNetCNN(_2d).training_step (models/magnet_cnn.py:393-437,
    models/magnet_cnn_2d.py:399-443) by torch autograd through the oracle restatement."""
    from oracle.models import _criterion
    o = oracle.to(dtype)
    b = {k: v.to(dtype) for k, v in batch.items()}
    t, u, vals, coords, cells = b["t"], b["lr_frames"], b["hr_points"], b["coords"], b["cells"]
    T, B = o.T, u.shape[0]
    fut = vals[:, T:]
    inp, hr_last, outs, hrs = u[:, :T], vals[:, T - 1], [], []
    as_nodes = lambda f: f.reshape(B, f.shape[1], 1, -1).permute(0, 1, 3, 2)  # noqa: E731
    for i in range(fut.shape[1] // T):
        out_hr, out_lr, hr_points = o.forward(inp, coords, cells, t[:, i * T:(i + 2) * T], hr_last)
        outs.append(torch.cat([out_hr, as_nodes(out_lr) if o.dim == 2 else out_lr], 2))
        hrs.append(hr_points)
        if teacher_forcing:
            inp, hr_last = u[:, (i + 1) * T:(i + 2) * T], vals[:, (i + 2) * T - 1]
        else:
            inp, hr_last = (out_lr if o.dim == 2 else out_lr.permute(0, 1, 3, 2)), out_hr[:, -1]
    pred, hr_hat = torch.cat(outs, 1), torch.cat(hrs, 1)
    target = torch.cat([fut, as_nodes(u[:, T:])], 2)
    crit = _criterion(loss_name)
    loss = crit(pred, target) + crit(hr_hat, vals[:, :-T])
    o.zero_grad()
    loss.backward()
    return float(loss.detach()), {n: p.grad.clone() for n, p in o.named_parameters()}
