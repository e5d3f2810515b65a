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


def split_bf16x3(w: torch.Tensor):
    """Exact 3-piece bf16 split of an fp32 tensor (truncation): w == hi + mid + lo, each piece has 8
    significant bits.  Returned as int16 tensors holding the bf16 bit patterns."""
    def top(t):
        return (t.contiguous().view(torch.int32) >> 16).to(torch.int16)

    def trunc(t):
        return (t.contiguous().view(torch.int32) & -65536).view(torch.float32)

    w = w.detach().float().contiguous()
    hi = trunc(w)
    r1 = w - hi
    mid = trunc(r1)
    lo = r1 - mid
    return top(hi), top(mid), top(lo)


def pack_linear_bf16x3(w: torch.Tensor, b: torch.Tensor | None, k_pad: int | None = None, n_pad: int | None = None,
                       col_map: torch.Tensor | None = None):
    """Split planes in MFMA fragment order [kt][nt][m][plane][lane][8] (int16) + padded fp32 bias.

        slot c' of lane (i, h), k-slot m  <->  W[nt*32 + i][kt*32 + 8*(2m + (c'>>2)) + 4*h + (c'&3)]
    """
    n, k = w.shape
    kp = k_pad or pad_to(k if col_map is None else int(col_map.max()) + 1)
    np_ = n_pad or pad_to(n)
    wp = torch.zeros(np_, kp, dtype=torch.float32, device=w.device)
    if col_map is None:
        wp[:n, :k] = w.detach().float()
    else:
        wp[:n, col_map.to(w.device)] = w.detach().float()
    kt, nt = kp // 32, np_ // 32
    planes = []
    for pl in split_bf16x3(wp):
        # [nt, i, kt, m, gg, h, c] -> [kt, nt, m, h, i, gg, c]
        planes.append(pl.view(nt, 32, kt, 2, 2, 2, 4).permute(2, 0, 3, 5, 1, 4, 6).reshape(kt, nt, 2, 64, 8))
    out = torch.stack(planes, dim=3).contiguous()           # [kt, nt, m, plane, lane, 8]
    bp = torch.zeros(np_, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:n] = b.detach().float()
    return out.view(-1), bp


def split_fp16x2(w: torch.Tensor):
    """Two-piece fp16 split: w ~= h + l / 2048 with h = fp16(w), l = fp16((w - h) * 2048) (both round-to-nearest;
    the remainder is exact in fp32 and the 2^11 scaling keeps it a normal fp16).  |w| must be < 65504."""
    w = w.detach().float().contiguous()
    if not bool(torch.isfinite(w).all()) or float(w.abs().max()) >= 65504.0:
        raise ValueError("fp16x3 GEMM mode needs finite weights with |w| < 65504; use MAGNET_GEMM=bf16x6")
    h = w.half()
    lo = ((w - h.float()) * 2048.0).half()
    return h, lo


def pack_linear_fp16x2(w: torch.Tensor, b: torch.Tensor | None, k_pad: int | None = None, n_pad: int | None = None,
                       col_map: torch.Tensor | None = None):
    """fp16 planes in MFMA fragment order [kt][nt][m][2][lane][8] (float16) + padded fp32 bias (same slot map as
    pack_linear_bf16x3)."""
    n, k = w.shape
    kp = k_pad or pad_to(k if col_map is None else int(col_map.max()) + 1)
    np_ = n_pad or pad_to(n)
    wp = torch.zeros(np_, kp, dtype=torch.float32, device=w.device)
    if col_map is None:
        wp[:n, :k] = w.detach().float()
    else:
        wp[:n, col_map.to(w.device)] = w.detach().float()
    kt, nt = kp // 32, np_ // 32
    planes = [pl.view(nt, 32, kt, 2, 2, 2, 4).permute(2, 0, 3, 5, 1, 4, 6).reshape(kt, nt, 2, 64, 8) for pl in split_fp16x2(wp)]
    out = torch.stack(planes, dim=3).contiguous()           # [kt, nt, m, plane, lane, 8]
    bp = torch.zeros(np_, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:n] = b.detach().float()
    return out.view(-1), bp


def pack_edge16_fp16x2(w: torch.Tensor, b: torch.Tensor | None):
    """message_net_2 (128 x 128) for the 16-edge-tile edge kernel (k_mpnn_edge16): the two fp16 planes in the
    v_mfma_f32_16x16x32_f16 A-operand order [4 k steps][8 output tiles][2 planes][64 lanes][8 fp16] - lane l of (s, t)
    holds W[16 t + (l & 15)][32 s + 8 (l >> 4) + 0..7] - plus the fp32 bias."""
    assert tuple(w.shape) == (128, 128), w.shape
    planes = [pl.view(8, 16, 4, 4, 8).permute(2, 0, 3, 1, 4).reshape(4, 8, 64, 8) for pl in split_fp16x2(w)]
    out = torch.stack(planes, dim=2).contiguous()            # [s, t, plane, lane, 8]
    bp = torch.zeros(128, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:] = b.detach().float()
    return out.view(-1), bp


def pack_edge32t_fp16x2(w: torch.Tensor, b: torch.Tensor | None):
    """message_net_2 (128 x 128) for the transposed 32-edge-tile edge kernel (k_mpnn_edge32t, magnet_amd/csrc/mpnn_edge32t.hip):
    fp16 planes in the v_mfma_f32_32x32x16_f16 B-operand order [8 k slots][4 output blocks][2 planes][64 lanes][8 fp16] - lane l
    of (s, nt) holds W[32 nt + (l & 31)][16 s + 8 (l >> 5) + 0..7].  Both planes carry the factor 2^11: plane 0 = 2^11 fp16(w),
    plane 1 = 2^11 (w - fp16(w)), so that all three products of the fp16x3 split (h.hw, h.lw, l.hw with the activation's low
    piece l UNSCALED) land in one accumulator on a common scale.  Raises ValueError when 2^11 |w| leaves the fp16 range
    (|w| >= 31.98): the caller keeps the 16-edge kernel then."""
    assert tuple(w.shape) == (128, 128), w.shape
    h, lo = split_fp16x2(w)
    if float(h.float().abs().max()) * 2048.0 >= 65504.0:
        raise ValueError("pack_edge32t_fp16x2: |w| >= 31.98 does not survive the 2^11 pre-scaling of the high plane")
    hs = (h.float() * 2048.0).half()                      # exact: a power-of-two scaling inside the fp16 range
    planes = [pl.view(4, 32, 8, 2, 8).permute(2, 0, 3, 1, 4).reshape(8, 4, 64, 8) for pl in (hs, lo)]
    out = torch.stack(planes, dim=2).contiguous()            # [s, nt, plane, lane, 8]
    bp = torch.zeros(128, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:] = b.detach().float()
    return out.view(-1), bp


def pack_linear_bf16x1(w: torch.Tensor, b: torch.Tensor | None, k_pad: int | None = None, n_pad: int | None = None,
                       col_map: torch.Tensor | None = None):
    """ONE bf16 plane (round-to-nearest-even) in MFMA fragment order [kt][nt][m][1][lane][8] (bfloat16) + padded fp32 bias:
    the reduced-precision "bf16" mode (bf16 inputs and weights, fp32 accumulate) BASELINE.json config 2 asks for."""
    n, k = w.shape
    kp = k_pad or pad_to(k if col_map is None else int(col_map.max()) + 1)
    np_ = n_pad or pad_to(n)
    wp = torch.zeros(np_, kp, dtype=torch.float32, device=w.device)
    if col_map is None:
        wp[:n, :k] = w.detach().float()
    else:
        wp[:n, col_map.to(w.device)] = w.detach().float()
    kt, nt = kp // 32, np_ // 32
    pl = wp.to(torch.bfloat16).view(torch.int16)
    out = pl.view(nt, 32, kt, 2, 2, 2, 4).permute(2, 0, 3, 5, 1, 4, 6).reshape(kt, nt, 2, 1, 64, 8).contiguous()
    bp = torch.zeros(np_, dtype=torch.float32, device=w.device)
    if b is not None:
        bp[:n] = b.detach().float()
    return out.view(-1).view(torch.bfloat16), bp


# ---- which arithmetic the Linears run on (ops picks the kernel from the dtype of the packed weights):
#   "fp16x3" (default)  two fp16 pieces, 3 MFMA products, |values| < 65504   -> float16 planes
#   "bf16x6"            three bf16 pieces, 6 MFMA products, full fp32 range    -> int16 planes
#   "fp32"              v_mfma_f32_32x32x2_f32                                 -> float32 fragments
#   "bf16"              REDUCED precision: one bf16 piece (RNE), 1 MFMA product -> bfloat16 plane.  Opt-in only (BASELINE
#                       config 2); ~1e-2 deviation from the fp32 result, reported as a deviation, never as parity
GEMM_MODES = ("fp16x3", "bf16x6", "fp32", "bf16")


def default_gemm() -> str:
    import os
    mode = os.environ.get("MAGNET_GEMM", "fp16x3")
    assert mode in GEMM_MODES, mode
    return mode


def pack(w, b, mode: str, k_pad=None, n_pad=None, col_map=None):
    if mode == "fp16x3":
        return pack_linear_fp16x2(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
    if mode == "bf16x6":
        return pack_linear_bf16x3(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
    if mode == "bf16":
        return pack_linear_bf16x1(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
    assert mode == "fp32", mode
    return pack_linear(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
