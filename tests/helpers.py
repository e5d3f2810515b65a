"""Shared helpers for the parity tests."""
import json
import os
import types

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["mpnn_1d", "mpnn_2d", "magnet_gnn_full", "magnet_gnn_tiny", "magnet_cnn_1d", "magnet_cnn_2d"]


def load_case(tag):
    d = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    hp = types.SimpleNamespace(**json.loads(str(d["hparams"])))
    batch = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("in_")}
    return str(d["name"]), hp, int(d["salt"]), batch, d


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
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
