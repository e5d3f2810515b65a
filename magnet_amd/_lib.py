"""ctypes binding of libmagnet_hip.so (the C ABI declared in include/magnet_hip.h).

The product path has no CPU fallback: if the library is missing or a symbol cannot be resolved
this module raises, and every op in magnet_amd.ops refuses tensors that are not on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch  # noqa: F401  (must come first: the library has to bind to the HIP runtime torch has loaded, not a second copy)

HERE = os.path.dirname(os.path.abspath(__file__))
# MAGNET_DEV_LIB=1 (development scripts only): the library built by `python -m magnet_amd.build --dev`, which additionally
# exports the run-time variant / ablation knobs of include/magnet_hip_dev.h
DEV = os.environ.get("MAGNET_DEV_LIB", "0") == "1"
# MAGNET_LIB_PATH (development scripts only): another build of the same library, e.g. an experiment variant
LIB_PATH = os.environ.get("MAGNET_LIB_PATH") or os.path.join(HERE, "csrc", "libmagnet_hip_dev.so" if DEV else "libmagnet_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "magnet_hip.h")

MAX_KT, MAX_LAYERS = 36, 6
ACT_NONE, ACT_SWISH, ACT_RELU = 0, 1, 2
EPI_STORE, EPI_STORE_RES, EPI_SEGMEAN, EPI_STORE_MASK, EPI_QUADBLEND = 0, 1, 2, 3, 4


class Src(C.Structure):
    _fields_ = [("p", C.c_void_p), ("idx", C.c_void_p), ("ld", C.c_int32), ("col", C.c_int32),
                ("scale", C.c_float), ("rdiv", C.c_int32)]


class KTile(C.Structure):
    _fields_ = [("s", Src * 2), ("n", C.c_int32), ("pad_", C.c_int32 * 3)]


class Layer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class MlpArgs(C.Structure):
    _fields_ = [("M", C.c_int32), ("nkt", C.c_int32), ("kt", KTile * MAX_KT), ("layer", Layer * MAX_LAYERS),
                ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float),
                ("y", C.c_void_p), ("ldy", C.c_int32), ("ncols", C.c_int32),
                ("res", C.c_void_p), ("ldres", C.c_int32),
                ("tgt", C.c_void_p), ("rowptr", C.c_void_p), ("partials", C.c_void_p), ("ncolgroups", C.c_int32),
                ("groups_in_block", C.c_int32), ("y_group_stride", C.c_int64), ("pre", Src * 2),
                ("x0_out", C.c_void_p), ("hid", C.c_void_p * MAX_LAYERS), ("x0_skip", C.c_uint64),
                ("status", C.c_void_p), ("status_bit", C.c_int32), ("pad2_", C.c_int32)]


DEV_HEADER = os.path.join(os.path.dirname(HERE), "include", "magnet_hip_dev.h")


def declared_symbols(header: str = HEADER) -> list[str]:
    """every function name declared in include/magnet_hip.h (or the development header)"""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(magnet_\w+)\s*\(", txt)))


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m magnet_amd.build` (hipcc, gfx950). "
                "magnet_amd has no CPU fallback.")
        if os.environ.get("MAGNET_LIB_PATH"):
            _check_variant(LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name in declared_symbols():
            fn = getattr(_lib, name)          # AttributeError if the .so does not export it
            fn.restype = C.c_int
            if name.endswith("_generation"):
                # a development switch between kernel generations changes what the *_grid / *_workspace queries answer: every call
                # through this handle drops the cached answers
                setattr(_lib, name, _forgetting(fn))
        if DEV:
            for name in declared_symbols(DEV_HEADER):   # the development library's extra entries (include/magnet_hip_dev.h)
                getattr(_lib, name).restype = C.c_int
    return _lib


def _check_variant(path: str) -> None:
    """a library that did not come out of magnet_amd.build (MAGNET_LIB_PATH: experiment variants of the dev scripts) gets the same
    code-object hazard check before it is loaded - once per file (a stamp next to it remembers the size and mtime that passed)"""
    st = os.stat(path)
    stamp, sig = path + ".hazard_ok", f"{st.st_size} {st.st_mtime_ns}"
    if os.path.exists(stamp) and open(stamp).read().strip() == sig:
        return
    from . import hazard_check
    bad = hazard_check.check(path)
    if bad:
        raise RuntimeError(f"{path}: {len(bad)} instruction pair(s) without the wait states they need (magnet_amd/hazard_check.py); "
                           f"first: {bad[0][3]}  ->  {bad[0][4]}")
    try:
        open(stamp, "w").write(sig)
    except OSError:
        pass


_QUERIES = {}


def _forgetting(fn):
    def call(*args):
        _QUERIES.clear()
        return fn(*args)
    return call


def query(name: str, *args) -> int:
    """value of a query entry point; ``*_supported`` / ``*_workspace`` / ``*_grid`` are pure functions of their integer arguments
    (and of the kernel generation: the ``*_generation`` switches clear the cache, see lib()) and cached: a MAgNet training step at the
    reference's batch sizes is host bound and asks these ~1 000 times per step.  (``*_grid`` take the row count as a C long.)"""
    key = (name, args)
    if key not in _QUERIES:
        cargs = [C.c_long(a) for a in args] if name.endswith("_grid") else args
        _QUERIES[key] = int(getattr(lib(), name)(*cargs))
    return _QUERIES[key]


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}"
                           + (" (unsupported argument)" if code < 0 else " (hipError_t)"))
