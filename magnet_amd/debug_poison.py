"""Debug switch: every ``torch.empty`` / ``empty_like`` / ``empty_strided`` / ``new_empty`` of a floating HIP tensor comes back filled
with NaN (``MAGNET_POISON=nan``) or with a huge finite value (``MAGNET_POISON=big``: 1e30).

The C ABI never allocates: every scratch array, partial-sum block and output of a kernel is a ``torch.empty`` made by ``ops.py`` /
``models/*.py`` / ``autograd.py``.  A kernel that reads scratch nobody wrote gets whatever the caching allocator left there - in an
eager run usually the previous call's values of the same array (plausible, often even right), in a fresh hipGraph pool anything.
With the poison a read of never-written memory shows up as NaN in the result instead of as a plausible stale value.  Integer arrays are
left alone (a poisoned index would be dereferenced).

Development / test use only: ``install()`` patches the torch module for the whole process; nothing in the product path imports this
module unless the environment variable is set (magnet_amd/__init__.py).
"""
from __future__ import annotations

import os

import torch

_installed = None


def install(kind: str = "nan") -> None:
    global _installed
    if _installed:
        return
    value = {"nan": float("nan"), "big": 1e30}[kind]
    real_empty, real_like, real_strided, real_new = torch.empty, torch.empty_like, torch.empty_strided, torch.Tensor.new_empty

    def poison(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(value)
        return t

    def empty(*a, **k):
        return poison(real_empty(*a, **k))

    def empty_like(*a, **k):
        return poison(real_like(*a, **k))

    def empty_strided(*a, **k):
        return poison(real_strided(*a, **k))

    def new_empty(self, *a, **k):
        return poison(real_new(self, *a, **k))

    torch.empty, torch.empty_like, torch.empty_strided, torch.Tensor.new_empty = empty, empty_like, empty_strided, new_empty
    _installed = kind


def install_from_env() -> None:
    kind = os.environ.get("MAGNET_POISON", "")
    if kind in ("nan", "big"):
        install(kind)
    elif kind not in ("", "0"):
        raise ValueError(f"MAGNET_POISON={kind!r}: expected 'nan' or 'big'")


def installed():
    return _installed
