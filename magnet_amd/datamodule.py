"""HDF5 batch schemas of the reference's datasets, host side.

Mirrors ``datamodule/dataset.py`` (1-D) and ``datamodule/dataset_2d.py`` (2-D): the three dataset flavours that feed the
hot-path models - graph (``mpnn`` / ``mpnn_2d``), implicit-GNN (``magnet_gnn``) and implicit (``magnet_cnn`` /
``magnet_cnn_2d``) - with the same constructor arguments, the same ``__getitem__`` dictionaries (keys, shapes, dtypes)
and the same use of ``np.random`` for the training-time query sampling, so that a seeded run draws the same queries.
File layout: ``f[mode]['pde_{nt}-{nx}']`` + ``'x'`` / ``'y'`` / ``'t'`` / ``'coords'`` (dataset.py:25-28, dataset_2d.py:24-29).

``h5py`` is imported lazily - it is not installed in the build image - and every class also accepts ``source=`` (any
nested mapping of arrays, e.g. an already opened ``h5py.File`` or a dict of numpy arrays), which is how the tests pin
these classes to the reference's own (run over the same in-memory tree, tests/golden/make_golden_data.py).
``to_device`` stacks a list of items into the batch dictionary the models take.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

from .utils import make_coord


def _open(path, source):
    if source is not None:
        return source
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise RuntimeError("reading .h5 files needs h5py (pip install h5py); pass source=<mapping> to use in-memory data") from e
    return h5py.File(path, "r")


class _H5Base(Dataset):
    def __init__(self, path, nt, res, mode, load_all, source):
        assert mode in ["train", "valid", "test"], "mode must belong to one of these ['train', 'val', 'test']"
        f = _open(path, source)
        self.mode, self.data, self.dataset = mode, f[mode], f"pde_{nt}-{res}"
        if load_all:                                              # dataset.py:32-35 (only the field array is kept)
            self.data = {self.dataset: self.data[self.dataset][:]}
            if hasattr(f, "close"):
                f.close()

    def __len__(self):
        return self.data[self.dataset].shape[0]


def _pixel_values(img):
    """``to_pixel_samples(img)[1]`` (utils.py:60-71): [C, ...] -> [prod(...), C]"""
    return img.reshape(img.shape[0], -1).permute(1, 0)


# ------------------------------------------------------------------------------------------------ 1-D (dataset.py)
class HDF5DatasetGraph(_H5Base):
    """dataset.py:12-51 -> {u [N, T], x [N, 1], t [T]}"""

    def __init__(self, path, nt, nx, mode="train", load_all=False, in_timesteps=16, radius=2, source=None):
        super().__init__(path, nt, nx, mode, load_all, source)
        self.radius, self.in_timesteps = radius, in_timesteps

    def __getitem__(self, idx):
        x = torch.from_numpy(np.asarray(self.data["x"][idx])).unsqueeze(-1)
        t = torch.from_numpy(np.asarray(self.data["t"][idx]))
        u = torch.from_numpy(np.asarray(self.data[self.dataset][idx])).permute(1, 0)
        return {"u": u, "x": x, "t": t}


class HDF5DatasetImplicitGNN(_H5Base):
    """dataset.py:53-124: LR mesh = every second node, queries = the odd nodes (a random subset of them when training)"""

    def __init__(self, path, nt, nx, sampling="uniform", mode="train", load_all=False, samples=256, source=None):
        super().__init__(path, nt, nx, mode, load_all, source)
        self.samples, self.sampling = samples, sampling

    def __getitem__(self, idx):
        x = np.asarray(self.data["x"][idx])
        x = 2 * (x - x.min()) / (x.max() - x.min()) - 1
        t = self.data["t"][idx]
        u_hr = torch.from_numpy(np.asarray(self.data[self.dataset][idx])).unsqueeze(1)          # T, 1, L
        L = u_hr.shape[-1]
        u_lr, lr_coord = u_hr[:, :, ::2], x[::2]
        left = np.setdiff1d(np.arange(0, L), np.arange(0, L)[::2])
        if self.mode == "train":
            sample_lst = torch.tensor(sorted(np.random.choice(left, self.samples, replace=False)))
            return {"t": t, "sample_idx": sample_lst, "lr_frames": u_lr, "hr_frames": u_hr,
                    "hr_points": u_hr[:, :, sample_lst].permute(0, 2, 1), "coords_hr": x[sample_lst], "coords_lr": lr_coord}
        return {"t": t, "lr_frames": u_lr, "hr_frames": u_hr, "hr_points": u_hr[:, :, left].permute(0, 2, 1),
                "coords_hr": x[left], "coords_lr": lr_coord}


class HDF5DatasetImplicit(_H5Base):
    """dataset.py:126-200: LR frames = linear down-sampling by 2, queries = cell centres of the HR grid"""

    def __init__(self, path, nt, nx, sampling="uniform", mode="train", load_all=False, samples=256, source=None):
        super().__init__(path, nt, nx, mode, load_all, source)
        self.samples, self.sampling = samples, sampling

    def __getitem__(self, idx):
        t = self.data["t"][idx]
        u_hr = torch.from_numpy(np.asarray(self.data[self.dataset][idx])).unsqueeze(1)          # T, 1, L
        T, _, L = u_hr.shape
        u_lr = F.interpolate(u_hr, size=(L // 2), mode="linear", align_corners=False)
        if self.mode == "train":
            if self.sampling == "uniform":
                sample_lst = torch.tensor(sorted(np.random.choice(L, self.samples, replace=False)))
            else:                                                 # 'boundary' (dataset.py:167-169)
                p = torch.softmax(torch.pow(torch.abs(torch.arange(L) - L // 2) / L, 2) / 0.1, dim=0).numpy()
                sample_lst = torch.tensor(sorted(np.random.choice(L, self.samples, p=p, replace=False)))
            hr_coord = make_coord([L])[sample_lst]
            cell = torch.ones_like(hr_coord) * (2 / L)
            hr_points = torch.stack([_pixel_values(u_hr[i])[sample_lst] for i in range(T)], dim=0)
            return {"t": t, "sample_idx": sample_lst, "lr_frames": u_lr, "hr_frames": u_hr, "hr_points": hr_points,
                    "coords": hr_coord, "cells": cell}
        hr_coord = make_coord([L])
        cell = torch.ones_like(hr_coord) * (2 / L)
        hr_points = torch.stack([_pixel_values(u_hr[i]) for i in range(T)], dim=0)
        return {"t": t, "lr_frames": u_lr, "hr_frames": u_hr, "hr_points": hr_points, "coords": hr_coord, "cells": cell}


# ------------------------------------------------------------------------------------------------ 2-D (dataset_2d.py)
class HDF5DatasetGraph_2d(_H5Base):
    """dataset_2d.py:12-59 -> {u [W*W, T], x [W*W, 2], t [T]}"""

    def __init__(self, path, nt, res, mode="train", regular=True, load_all=False, source=None):
        super().__init__(path, nt, res, mode, load_all, source)
        self.regular = regular

    def __getitem__(self, idx):
        u = torch.from_numpy(np.asarray(self.data[self.dataset][idx]))
        u = u.reshape(u.shape[0], -1).permute(1, 0)
        if self.regular:
            x, y = torch.from_numpy(np.asarray(self.data["x"][idx])), torch.from_numpy(np.asarray(self.data["y"][idx]))
            coords = torch.stack(torch.meshgrid(x, y, indexing="ij"), dim=-1).reshape(-1, 2)
        else:
            coords = torch.from_numpy(np.asarray(self.data["coords"][idx]))
        return {"u": u, "x": coords, "t": torch.from_numpy(np.asarray(self.data["t"][idx]))}


class HDF5DatasetImplicitGNN_2d(_H5Base):
    """dataset_2d.py:61-140 (note: np.meshgrid's default 'xy' indexing, as the reference)"""

    def __init__(self, path, nt, res, mode="train", regular=True, load_all=False, samples=256, source=None):
        super().__init__(path, nt, res, mode, load_all, source)
        self.samples, self.regular = samples, regular

    def __getitem__(self, idx):
        u_hr = torch.from_numpy(np.asarray(self.data[self.dataset][idx])).unsqueeze(1)
        if self.regular:
            coords = np.stack(np.meshgrid(np.asarray(self.data["x"][idx]), np.asarray(self.data["y"][idx])), axis=-1)
            coords = coords.reshape(-1, coords.shape[-1])
            u_hr = u_hr.reshape(u_hr.shape[0], 1, -1)
        else:
            coords = np.asarray(self.data["coords"][idx])
        coords = 2 * (coords - coords.min(0)) / (coords.max(0) - coords.min(0)) - 1
        t = self.data["t"][idx]
        N = u_hr.shape[-1]
        u_lr, lr_coord = u_hr[:, :, ::2], coords[::2]
        left = np.setdiff1d(np.arange(0, N), np.arange(0, N)[::2])
        if self.mode == "train":
            sample_lst = torch.tensor(sorted(np.random.choice(left, self.samples, replace=False)))
            return {"t": t, "sample_idx": sample_lst, "lr_frames": u_lr, "hr_frames": u_hr,
                    "hr_points": u_hr[:, :, sample_lst].permute(0, 2, 1), "coords_hr": coords[sample_lst], "coords_lr": lr_coord}
        return {"t": t, "lr_frames": u_lr, "hr_frames": u_hr, "hr_points": u_hr[:, :, left].permute(0, 2, 1),
                "coords_hr": coords[left], "coords_lr": lr_coord}


class HDF5DatasetImplicit_2d(_H5Base):
    """dataset_2d.py:142-210: LR frames = bilinear down-sampling by 2, queries = cell centres of the HR grid"""

    def __init__(self, path, nt, res, mode="train", load_all=False, samples=256, source=None):
        super().__init__(path, nt, res, mode, load_all, source)
        self.samples = samples

    def __getitem__(self, idx):
        t = self.data["t"][idx]
        u_hr = torch.from_numpy(np.asarray(self.data[self.dataset][idx])).unsqueeze(1)          # T, 1, W, W
        T, _, W, _ = u_hr.shape
        u_lr = F.interpolate(u_hr, size=(W // 2), mode="bilinear", align_corners=False)
        if self.mode == "train":
            sample_lst = torch.tensor(sorted(np.random.choice(W * W, self.samples, replace=False)))
            hr_coord = make_coord([W, W])[sample_lst]
            cell = torch.ones_like(hr_coord) * (2 / W)
            hr_points = torch.stack([_pixel_values(u_hr[i])[sample_lst] for i in range(T)], dim=0)
            return {"t": t, "sample_idx": sample_lst, "lr_frames": u_lr, "hr_frames": u_hr, "hr_points": hr_points,
                    "coords": hr_coord, "cells": cell}
        hr_coord = make_coord([W, W])
        cell = torch.ones_like(hr_coord) * (2 / W)
        hr_points = torch.stack([_pixel_values(u_hr[i]) for i in range(T)], dim=0)
        return {"t": t, "lr_frames": u_lr, "hr_frames": u_hr, "hr_points": hr_points, "coords": hr_coord, "cells": cell}


DATASETS = {"graph": HDF5DatasetGraph, "implicit_gnn": HDF5DatasetImplicitGNN, "implicit": HDF5DatasetImplicit,
            "graph_2d": HDF5DatasetGraph_2d, "implicit_gnn_2d": HDF5DatasetImplicitGNN_2d, "implicit_2d": HDF5DatasetImplicit_2d}


def to_device(items, device):
    """default-collate a list of dataset items (numpy arrays / tensors per key) and move the batch to ``device``: the
    dictionaries ``training_step`` / ``validation_step`` take"""
    out = {}
    for k in items[0]:
        vals = [torch.as_tensor(np.asarray(it[k]) if not torch.is_tensor(it[k]) else it[k]) for it in items]
        out[k] = torch.stack(vals, 0).to(device)
    return out
