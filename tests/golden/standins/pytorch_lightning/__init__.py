"""Minimal stand-in so the reference's LightningModules import (golden generation only)."""
import torch
from torch import nn


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.logged = {}

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, name, value, **k):
        self.logged[name] = value.detach().clone() if torch.is_tensor(value) else value

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")


class LightningDataModule:
    pass


class Callback:
    pass


class Trainer:
    def __init__(self, *a, **k):
        raise RuntimeError("stand-in")


def seed_everything(seed, workers=False):
    torch.manual_seed(seed)
