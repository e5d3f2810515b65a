"""``pytorch_lightning.LightningModule`` when importable, else a minimal local base so the models
run on a box that only has torch (SURVEY.md 8(b) "must also work without")."""
from __future__ import annotations

import torch
from torch import nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl

    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        """device / log / save_hyperparameters, nothing else."""

        def __init__(self):
            super().__init__()
            self.logged = {}
            self.hparams_saved = None

        def save_hyperparameters(self, *args, **kwargs):
            self.hparams_saved = args[0] if args else kwargs

        def log(self, name, value, **kwargs):
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")
