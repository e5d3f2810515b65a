"""Golden batches of the reference's dataset classes (datamodule/dataset.py, dataset_2d.py), produced by running THEM over
an in-memory HDF5-like tree through the h5py stand-in (tests/golden/standins/h5py).  Runs only in the build container
(needs /root/reference).  Output: tests/golden/data_items.npz = the synthetic tree + every class's __getitem__ result
(train items with np.random seeded) - what tests/test_datamodule_cpu.py checks magnet_amd/datamodule.py against.

    python tests/golden/make_golden_data.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(HERE, "standins"), "/root/reference"]
warnings.filterwarnings("ignore")

import h5py  # noqa: E402  (the stand-in)
from datamodule import dataset as D1  # noqa: E402  (the reference)
from datamodule import dataset_2d as D2  # noqa: E402


def synthetic_tree(seed=0):
    rng = np.random.default_rng(seed)
    nt, nx, res, n_irr, S = 12, 40, 8, 30, 3
    one_d = {m: {f"pde_{nt}-{nx}": rng.standard_normal((S, nt, nx)).astype(np.float32),
                 "x": np.tile(np.linspace(0, 16, nx, dtype=np.float32), (S, 1)) + rng.uniform(0, 0.01, (S, nx)).astype(np.float32),
                 "t": np.tile(np.linspace(0, 4, nt, dtype=np.float32), (S, 1))} for m in ("train", "valid", "test")}
    two_d = {m: {f"pde_{nt}-{res}": rng.standard_normal((S, nt, res, res)).astype(np.float32),
                 "x": np.tile(np.linspace(0, 1, res, dtype=np.float32), (S, 1)),
                 "y": np.tile(np.linspace(0, 2, res, dtype=np.float32), (S, 1)),
                 "t": np.tile(np.linspace(0, 1, nt, dtype=np.float32), (S, 1))} for m in ("train", "valid", "test")}
    irr = {m: {f"pde_{nt}-{n_irr}": rng.standard_normal((S, nt, n_irr)).astype(np.float32),
               "coords": rng.uniform(-1, 1, (S, n_irr, 2)).astype(np.float32),
               "t": np.tile(np.linspace(0, 1, nt, dtype=np.float32), (S, 1))} for m in ("train", "valid", "test")}
    return dict(nt=nt, nx=nx, res=res, n_irr=n_irr), {"1d": one_d, "2d": two_d, "irr": irr}


CASES = [  # (tag, module, class, tree, ctor kwargs)
    ("graph", D1, "HDF5DatasetGraph", "1d", lambda c: dict(nt=c["nt"], nx=c["nx"])),
    ("implicit_gnn", D1, "HDF5DatasetImplicitGNN", "1d", lambda c: dict(nt=c["nt"], nx=c["nx"], samples=7)),
    ("implicit", D1, "HDF5DatasetImplicit", "1d", lambda c: dict(nt=c["nt"], nx=c["nx"], samples=9)),
    ("implicit_boundary", D1, "HDF5DatasetImplicit", "1d", lambda c: dict(nt=c["nt"], nx=c["nx"], samples=9, sampling="boundary")),
    ("graph_2d", D2, "HDF5DatasetGraph_2d", "2d", lambda c: dict(nt=c["nt"], res=c["res"])),
    ("graph_2d_irregular", D2, "HDF5DatasetGraph_2d", "irr", lambda c: dict(nt=c["nt"], res=c["n_irr"], regular=False)),
    ("implicit_gnn_2d", D2, "HDF5DatasetImplicitGNN_2d", "2d", lambda c: dict(nt=c["nt"], res=c["res"], samples=11)),
    ("implicit_gnn_2d_irregular", D2, "HDF5DatasetImplicitGNN_2d", "irr", lambda c: dict(nt=c["nt"], res=c["n_irr"], regular=False, samples=6)),
    ("implicit_2d", D2, "HDF5DatasetImplicit_2d", "2d", lambda c: dict(nt=c["nt"], res=c["res"], samples=13)),
]


def main():
    cfg, trees = synthetic_tree()
    out = {"cfg_" + k: v for k, v in cfg.items()}
    for name, tree in trees.items():
        h5py.register(name, tree)
        for mode, grp in tree.items():
            for key, arr in grp.items():
                out[f"tree:{name}:{mode}:{key}"] = arr
    for tag, mod, cls, tree, kw in CASES:
        for mode in ("train", "valid"):
            for load_all in (False,):
                ds = getattr(mod, cls)(tree, mode=mode, load_all=load_all, **kw(cfg))
                np.random.seed(123)
                for idx in (0, 2):
                    item = ds[idx]
                    for k, v in item.items():
                        out[f"item:{tag}:{mode}:{idx}:{k}"] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
                out[f"len:{tag}:{mode}"] = len(ds)
    path = os.path.join(HERE, "data_items.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)} arrays -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
