"""Version-stable deterministic parameter fill (TEST INFRASTRUCTURE).

Golden fixtures do not store weights.  Both the golden generator (which loads
them into the imported reference classes) and the tests (which load them into
the oracle and the HIP models) regenerate identical fp32 parameters from the
state-dict key alone with a splitmix64 counter generator, independent of torch
or numpy RNG streams.
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(key: str, n: int, salt: int = 0) -> np.ndarray:
    """n float64 values in [-1, 1) determined by (key, salt)."""
    base = np.uint64((_fnv1a64(key) + 0x632BE59BD9B4E019 * salt) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
    bits = _splitmix64(ctr) >> np.uint64(11)               # 53 random bits
    return bits.astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def fill_state_dict(module: torch.nn.Module, salt: int = 0) -> None:
    """Overwrite every parameter/buffer of ``module`` in place (torch-default-like scales)."""
    sd = module.state_dict()
    new = {}
    for key, t in sd.items():
        shape = tuple(t.shape)
        n = int(np.prod(shape)) if shape else 1
        u = uniform(key, n, salt).reshape(shape)
        if key.endswith("weight") and t.dim() == 1:         # LayerNorm gain
            v = 1.0 + 0.1 * u
        elif key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            v = u / np.sqrt(fan_in)
        else:                                               # biases (Linear/Conv/LayerNorm)
            v = 0.1 * u
        new[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape)
    module.load_state_dict(new)


def checksum(module: torch.nn.Module) -> float:
    return float(sum(float(v.double().abs().sum()) for v in module.state_dict().values()))
