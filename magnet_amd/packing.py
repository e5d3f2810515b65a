"""Weight layout transforms: torch ``Linear`` [N_out, K] -> MFMA fragment order (include/magnet_hip.h).

    Wp[kt][nt][g][lane][c] = W[nt*32 + (lane & 31)][kt*32 + 8*g + 4*(lane >> 5) + c]

Pure index shuffling (plumbing); done once per weight version and cached by the models.
"""
from __future__ import annotations

import torch


def pad_to(n: int, m: int = 32) -> int:
    return (n + m - 1) // m * m


def pack_linear(w: torch.Tensor, b: torch.Tensor | None, k_pad: int | None = None, n_pad: int | None = None,
                col_map: torch.Tensor | None = None):
    """Return (packed weight [KT*NT*4*64*4] fp32, padded bias [NT*32]) on w's device.

    ``col_map`` (optional, int64 [K]) places input column k of ``w`` at padded column col_map[k]."""
    n, k = w.shape
    kp = k_pad or pad_to(k if col_map is None else int(col_map.max()) + 1)
    np_ = n_pad or pad_to(n)
    wp = torch.zeros(np_, kp, dtype=torch.float32, device=w.device)
    if col_map is None:
        wp[:n, :k] = w.detach().float()
    else:
        wp[:n, col_map.to(w.device)] = w.detach().float()
    kt, nt = kp // 32, np_ // 32
    # [nt, i, kt, g, half, c] -> [kt, nt, g, half, i, c]
    v = wp.view(nt, 32, kt, 4, 2, 4).permute(2, 0, 3, 4, 1, 5).contiguous()
    bp = torch.zeros(np_, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:n] = b.detach().float()
    return v.view(-1), bp


def pad_vec(v: torch.Tensor, n_pad: int) -> torch.Tensor:
    out = torch.zeros(n_pad, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.detach().float()
    return out
