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


def _too_large_for_scaled_plane(w, h) -> bool:
    """does 2^11 fp16(w) leave the fp16 range (|w| >= 31.98)?  One host read - except while a training step is being CAPTURED
    (graphs.TrainStepGraph: the re-pack of the weights is part of the captured step and nothing in it may synchronise): then the
    verdict goes to the device status word (flag_if_abs_reaches), the packer carries on as if the weight fitted, and the step's
    owner reads the word behind the replay - a trip repeats the step in the full-range split."""
    from .graphs import capturing
    if w.is_cuda and capturing():
        flag_if_abs_reaches(w, 31.98)
        return False
    return float(h.float().abs().max()) * 2048.0 >= 65504.0


def split_fp16x2(w: torch.Tensor):
    """Two-piece fp16 split: w ~= h + l / 2048 with h = fp16(w), l = fp16((w - h) * 2048) (both round-to-nearest;
    the remainder is exact in fp32 and the 2^11 scaling keeps it a normal fp16).  |w| must be < 65504."""
    w = w.detach().float().contiguous()
    if w.is_cuda:            # no host round trip: the verdict lands in the device status word (raise_if_bad_weights reads it once)
        bad = (~(w.abs() < 65504.0)).any().to(torch.int32)
        _status_word(w.device).bitwise_or_(bad * _WEIGHT_RANGE_BIT)
    elif not bool(torch.isfinite(w).all()) or float(w.abs().max()) >= 65504.0:
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


def pack_edge16_fp16x2(w: torch.Tensor, b: torch.Tensor | None, scaled: bool = False):
    """message_net_2 (128 x 128) for the 16-edge-tile edge kernel (k_mpnn_edge16): the two fp16 planes in the
    v_mfma_f32_16x16x32_f16 A-operand order [4 k steps][8 output tiles][2 planes][64 lanes][8 fp16] - lane l of (s, t)
    holds W[16 t + (l & 15)][32 s + 8 (l >> 4) + 0..7] - plus the fp32 bias.  ``scaled``: both planes carry 2^11 (the kernel's
    one-accumulator variant, ops.mpnn_edge16(single=True))."""
    assert tuple(w.shape) == (128, 128), w.shape
    h, lo = split_fp16x2(w)
    if scaled:     # the single-accumulator variant of the kernel: plane 0 = 2^11 fp16(w) (exact while |w| < 31.98), see pack_edge32t_fp16x2
        if _too_large_for_scaled_plane(w, h):
            raise ValueError("pack_edge16_fp16x2(scaled=True): |w| >= 31.98 does not survive the 2^11 pre-scaling of the high plane")
        h = (h.float() * 2048.0).half()
    planes = [pl.view(8, 16, 4, 4, 8).permute(2, 0, 3, 1, 4).reshape(4, 8, 64, 8) for pl in (h, lo)]
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
    if _too_large_for_scaled_plane(w, h):
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


class SinglePlanes(torch.Tensor):
    """fp16 planes of packing mode "fp16x3s" (high plane and bias times 2^11: the one-accumulator chain kernel,
    magnet_mlp_fp16x3s_f32).  Same dtype and size as an "fp16x3" pack - the type is what tells ops.mlp_plan which entry to call."""

    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t, False)


def chain_gemm(n_linears: int, hidden: int, mode: str) -> str:
    """GEMM mode of a register-chained MLP with ``n_linears`` Linears of width ``hidden`` whose model-wide mode is ``mode``: the
    5-Linear 64-wide chains run the one-accumulator form of fp16x3 - 124 instead of 142 registers, i.e. four waves per SIMD instead
    of three, and 40 % fewer VALU instructions per layer (the 128-wide chains stay at two waves per SIMD either way
    and gain nothing).  MAGNET_CHAIN_GEMM=fp16x3 keeps the two-accumulator form (A/B runs, |w| >= 31.98)."""
    import os
    if mode == "fp16x3" and n_linears == 5 and hidden == 64:
        m = os.environ.get("MAGNET_CHAIN_GEMM", "fp16x3s")
        assert m in ("fp16x3", "fp16x3s"), m
        return m
    return mode


_wide = False            # process-wide: the fp16x3 modes have been widened to bf16x6 (a value left the fp16 range, see widen())


def param_list(module) -> list:
    """list(module.parameters()), built once per module: nn.Module.parameters() walks the module tree (named_modules -> named_members)
    on every call - 2.4 ms of a 80 ms MAgNet training step that looks the packed-weight cache up four times.  The
    Parameter objects of a built model stay (load_state_dict, .to(), optimizers write them in place); a module that registers new
    parameters later calls ``module._plist = None``."""
    pl = module.__dict__.get("_plist")
    if pl is None:
        pl = module.__dict__["_plist"] = list(module.parameters())
    return pl


def wide() -> bool:
    return _wide


def widen(reason: str) -> bool:
    """Switch every fp16x3 / fp16x3s GEMM of this process to the full-range bf16x6 split (same fp32-class accuracy, the whole fp32
    exponent range, ~1.7 x the rollout time).  Called by the models when their range guard trips - an input, an intermediate or a
    weight reached the limit of the two-fp16-piece representation (65504; 31.98 for the pre-scaled planes) - so that such data gets
    the reference's result instead of an error (configs/trainer/default.yaml:5: the reference computes in fp32 over the whole fp32
    range).  Sticky until narrow(): data of that scale rarely comes alone.  -> True when this call made the switch."""
    global _wide
    if _wide:
        return False
    import warnings
    warnings.warn(f"magnet_amd: {reason}; switching the fp16x3 GEMMs of this process to the full-range bf16x6 split "
                  "(same accuracy, slower) - magnet_amd.packing.narrow() goes back", RuntimeWarning, stacklevel=3)
    _wide = True
    return True


def narrow() -> None:
    """back to the configured GEMM modes (the models re-pack on their next call)"""
    global _wide
    _wide = False


def effective(mode: str) -> str:
    """the mode a Linear configured as ``mode`` runs in right now"""
    return "bf16x6" if _wide and mode in ("fp16x3", "fp16x3s") else mode


def default_gemm() -> str:
    import os
    mode = os.environ.get("MAGNET_GEMM", "fp16x3")
    assert mode in GEMM_MODES, mode
    return effective(mode)


# ---- device path: one HIP launch per Linear (magnet_pack_linear_f32, csrc/pack_kernels.hip) instead of the torch index shuffles above.
# A training step re-packs every weight the optimizer touched; the torch path costs ~30 small kernels per Linear plus a host
# synchronisation for the fp16 range check - ~60 000 launches and ~600 `.item()` round trips per MAgNet training step.
_MODE_ID = {"fp32": 0, "fp16x3": 1, "bf16x6": 2, "bf16": 3, "fp16x3s": 4}
_OUT_DTYPE = {"fp32": torch.float32, "fp16x3": torch.float16, "bf16x6": torch.int16, "bf16": torch.bfloat16, "fp16x3s": torch.float16}
_PLANES = {"fp32": 2, "fp16x3": 2, "bf16x6": 3, "bf16": 1, "fp16x3s": 2}        # out elements per padded weight entry / (fp32: 1 float = 2 x 16 bit)
_WEIGHT_RANGE_BIT = 1
_status = {}             # device -> int32[1]: bit 0 = an fp16x3 pack met |w| >= 65504 or a non-finite weight
_src_maps = {}           # (column placement, padded width, device) -> int32 source-column map


def _status_word(dev):
    if dev not in _status:
        _status[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _status[dev]


def flag_if_abs_reaches(w: torch.Tensor, limit: float) -> None:
    """device-side (no host read): set the range bit of w's device when max|w| >= limit - for the kernels that pre-scale the
    fp16 high plane by 2^11 (k_mpnn_edge_bwd_w: |w| < 31.98); reported by the next raise_if_bad_weights"""
    if w.is_cuda:
        # ONE launch (magnet_status_check_f32) instead of the ten small torch kernels of `isfinite().all() | abs().amax() >= limit` and
        # their host-side dispatch: a MAgNet training step at the reference's batch sizes is host bound, and this runs once per Linear
        # and backward
        import ctypes as C

        from . import _lib as L
        x = w.detach()
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        fn = L.lib().magnet_status_check_f32
        sw = _status_word(w.device)
        args = (C.c_void_p(x.data_ptr()), C.c_long(x.numel()), C.c_float(float(limit)), _WEIGHT_RANGE_BIT, _WEIGHT_RANGE_BIT, 0,
                C.c_void_p(sw.data_ptr()))
        L.check(fn(*args, C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)), "status_check")
        rec = _active_recorder
        if rec is not None:
            if rec.backed(x):
                rec.launches.append((fn, args, "status_check", (x, sw)))
            else:
                rec.ok = False
    elif not bool(torch.isfinite(w).all()) or float(w.detach().abs().max()) >= limit:
        raise ValueError(f"fp16x3 GEMM mode: this kernel needs finite weights with |w| < {limit}; use MAGNET_GEMM=bf16x6")


def bad_weights(dev, training: bool = False) -> bool:
    """ONE host read for all packs since the last call (the models call it once per re-pack of their weights) -> True when an
    fp16x3 pack met a weight outside its range; the flag is cleared.  ``training``: a weight in [31.98, 65504) counts as well - it
    fits the forward's two-accumulator planes, but the backward kernels that keep 2^11 x the high plane (linear_bwd128 / 64, the edge
    backward) would overflow on it; the pack kernel notes it (bit 2) so that the step is widened BEFORE its backward runs, not one
    optimizer step of NaN gradients later"""
    dev = torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if dev not in _status:
        return False
    st = int(_status[dev].item())
    if st:
        _status[dev].zero_()
    return bool(st & _WEIGHT_RANGE_BIT) or (training and bool(st & (_WEIGHT_RANGE_BIT << 1)))


# A rollout / training step whose caller reads the range status at its END opens a scope; packs made inside leave the status word unread
# until then.  Reading it right behind the pack launches makes the host wait for everything queued before them - at the start of a
# training step that is the previous step's backward and optimizer (7.8 ms per read at the reference's batch sizes).
_defer_depth = 0
_unchecked = set()


class deferred_weight_checks:
    def __enter__(self):
        global _defer_depth
        _defer_depth += 1

    def __exit__(self, *exc):
        global _defer_depth
        _defer_depth -= 1
        return False


def weights_check_due(dev) -> bool:
    """packs were just made on ``dev``: True = read the status word now (bad_weights); False = a scope is open and its owner will read
    it at the end (pending_weight_check)"""
    if _defer_depth > 0:
        _unchecked.add(torch.device(dev))
        return False
    return True


def note_unchecked(dev) -> None:
    """packs were made on ``dev`` and nobody has read the status word yet (pending_weight_check will say so)"""
    _unchecked.add(torch.device(dev))


def pending_weight_check(dev) -> bool:
    dev = torch.device(dev)
    if dev in _unchecked:
        _unchecked.discard(dev)
        return True
    return False


def raise_if_bad_weights(dev) -> None:
    """bad_weights() as an exception (callers without a fallback): the fp16x3 mode represents a weight with two fp16 pieces and
    needs finite |w| < 65504"""
    if bad_weights(dev):
        raise ValueError("fp16x3 GEMM mode needs finite weights with |w| < 65504 (|w| < 31.98 for message_net_2 when training with "
                         "edge_bwd='fused_w' and for the 64-wide 5-Linear chains of the MAgNet models); use MAGNET_GEMM=bf16x6, "
                         "edge_bwd='fused' or MAGNET_CHAIN_GEMM=fp16x3")


def source_map(placed_at, width: int, dev, source_cols=None):
    """int32 [width] on ``dev``: padded position p takes source column source_cols[i] (default i) where placed_at[i] == p, -1
    elsewhere; cached (the placements are static per call site)"""
    if torch.is_tensor(placed_at) and placed_at.is_cuda:       # built on the device: reading it back would be a host synchronisation
        m = torch.full((int(width),), -1, dtype=torch.int32, device=dev)
        srcs = (torch.arange(placed_at.numel(), dtype=torch.int32, device=dev) if source_cols is None
                else torch.as_tensor(source_cols, device=dev).to(torch.int32))
        m[placed_at.to(dev).long()] = srcs
        return m
    pl = tuple(int(v) for v in (placed_at.tolist() if torch.is_tensor(placed_at) else placed_at))
    sc = tuple(range(len(pl))) if source_cols is None else tuple(int(v) for v in (source_cols.tolist() if torch.is_tensor(source_cols)
                                                                                   else source_cols))
    key = (pl, sc, int(width), str(dev))
    if key not in _src_maps:
        m = torch.full((int(width),), -1, dtype=torch.int32)
        m[torch.tensor(pl, dtype=torch.long)] = torch.tensor(sc, dtype=torch.int32)
        if len(_src_maps) > 256:
            _src_maps.clear()
        _src_maps[key] = m.to(dev)
    return _src_maps[key]


# ------------------------------------------------------------------------------------------------ re-packing in place
class PackRecorder:
    """The pack launches of one model, recorded while its packed-weight namespace is built, so that after an optimizer step the SAME
    launches run again into the SAME output buffers instead of rebuilding the namespace (at the reference's batch sizes
    a MAgNet training step is host bound, and re-packing ~200 Linears + ~170 transposed packs of the backward through the generic
    path was 20 of its 80 ms).  Pointer-stable packs also keep everything cached ON the pack namespaces (transposed packs, EDSR launch
    plans) alive across steps.

    Valid only while every recorded input is a view of a parameter's storage (or a tensor registered with ``derived``, which is
    refreshed first): a pack whose input was a temporary copy marks the recorder ``ok = False`` and the owner falls back to a rebuild."""

    def __init__(self, params, key=None):
        self.params = list(params)
        self.storages = {p.untyped_storage().data_ptr() for p in self.params}
        self.launches, self.derived, self.ok, self.key = [], [], True, key
        self.generation = 0          # replays so far: what is cached on a pack namespace but NOT replayed carries it in its key

    def backed(self, t) -> bool:
        return t is None or t.untyped_storage().data_ptr() in self.storages

    # the launches hold ctypes function pointers and argument blocks of ONE model instance's buffers: a copy.deepcopy / pickle of a
    # model (EMA / best-model snapshots, spawn-style launchers) gets a dead recorder and rebuilds its packs on first use
    def __deepcopy__(self, memo):
        dead = PackRecorder([], None)
        dead.ok = False
        return dead

    def __getstate__(self):
        return {"params": [], "storages": set(), "launches": [], "derived": [], "ok": False, "key": None, "generation": 0}

    def usable(self) -> bool:
        """the parameters still live where they lived when the launches were recorded (.to() / load_state_dict(assign=True) move them)"""
        return self.ok and all(p.untyped_storage().data_ptr() in self.storages for p in self.params)

    def replay(self) -> None:
        from . import _lib as L
        from .ops import _stream
        self.generation += 1
        with torch.no_grad():
            for t, fn in self.derived:
                t.copy_(fn())
        st = _stream()
        for fn, args, what, _keep in self.launches:
            L.check(fn(*args, st), what)


_active_recorder = None

# per-instance device caches of the models (packed weights with their recorder and the launch plans cached on them, parameter lists,
# EDSR tap tables / plans): never part of a copy or a pickle - the copy rebuilds them on first use
_DEVICE_CACHE_ATTRS = {"_pack": None, "_pack_key": None, "_pack_rec": None, "_plist": None, "_plan_key": None, "_plan_val": None,
                       "_taps": {}, "_topo": None}


def state_without_device_caches(module) -> dict:
    """``__getstate__`` of the model classes (used by pickle AND copy.deepcopy): the instance dict with the device caches reset"""
    st = dict(module.__dict__)
    for k, v in _DEVICE_CACHE_ATTRS.items():
        if k in st:
            st[k] = type(v)() if v is not None else None
    return st


class recording:
    """``with recording(rec):`` pack_device / flag_if_abs_reaches / derived calls inside are added to ``rec`` (None: no recording)"""

    def __init__(self, rec):
        self.rec = rec

    def __enter__(self):
        global _active_recorder
        self.prev, _active_recorder = _active_recorder, self.rec
        return self.rec

    def __exit__(self, *exc):
        global _active_recorder
        _active_recorder = self.prev
        return False


def derived(fn):
    """``fn()`` -> a tensor computed FROM parameters (a column selection, a scaled block, a permuted convolution weight) that a pack or a
    kernel reads: under a recorder it is kept and refreshed with ``copy_(fn())`` before the pack launches are replayed"""
    t = fn()
    rec = _active_recorder
    if rec is not None and t.is_cuda:
        if rec.backed(t):                              # a view of a parameter needs no refresh (and copy_ onto itself would be wrong)
            return t
        rec.derived.append((t, fn))
        rec.storages.add(t.untyped_storage().data_ptr())
    return t


def pack_device(w, b, mode: str, np_: int, kp: int, row_src=None, col_src=None):
    """``w``: 2-D fp32 CUDA tensor or view (any strides: W.t() costs nothing); row_src / col_src: int32 device maps from padded
    row / column to source row / column (-1 = zero), None = identity.  -> (flat packed planes, padded bias [np_])"""
    import ctypes as C

    from . import _lib as L
    w = w.detach()
    if w.dtype != torch.float32:
        w = w.float()
    dev = w.device
    assert w.dim() == 2 and np_ % 32 == 0 and kp % 32 == 0, (w.shape, np_, kp)
    out = torch.empty(np_ * kp * _PLANES[mode] // (2 if mode == "fp32" else 1), dtype=_OUT_DTYPE[mode], device=dev)
    bp = torch.empty(np_, dtype=torch.float32, device=dev)
    if b is not None:
        b = b.detach()
        b = b.float() if b.dtype != torch.float32 else b
        b = b if b.is_contiguous() else b.contiguous()
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    st = _status_word(dev) if mode in ("fp16x3", "fp16x3s") else None
    fn = L.lib().magnet_pack_linear_f32
    args = (ptr(w), C.c_long(w.stride(0)), C.c_long(w.stride(1)), w.shape[0], w.shape[1], ptr(row_src), ptr(col_src), ptr(b),
            b.numel() if b is not None else 0, np_, kp, _MODE_ID[mode], ptr(out), ptr(bp), ptr(st), _WEIGHT_RANGE_BIT)
    L.check(fn(*args, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "pack_linear")
    rec = _active_recorder
    if rec is not None:
        if rec.backed(w) and rec.backed(b):
            rec.launches.append((fn, args, "pack_linear", (w, b, row_src, col_src, out, bp, st)))
        else:
            rec.ok = False                            # packed from a temporary: a replay would read stale data
    return (SinglePlanes(out) if mode == "fp16x3s" else out), bp


def pack(w, b, mode: str, k_pad=None, n_pad=None, col_map=None):
    """(packed weight, padded bias) of a torch Linear weight [n_out, k_in] in the layout of GEMM mode ``mode``.  CUDA tensors are
    packed by one HIP launch (any strides; the fp16 range check lands in a device status word, see raise_if_bad_weights); CPU
    tensors by the torch index shuffles above (layout tests, no GPU needed)."""
    if w.is_cuda:
        n, k = w.shape
        kp = k_pad or pad_to(k if col_map is None else int(max(col_map.tolist() if torch.is_tensor(col_map) else col_map)) + 1)
        np_ = n_pad or pad_to(n)
        cs = source_map(col_map, kp, w.device) if col_map is not None else None
        return pack_device(w, b, mode, np_, kp, col_src=cs)
    if mode == "fp16x3":
        return pack_linear_fp16x2(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
    if mode == "fp16x3s":
        flag_if_abs_reaches(w, 31.98)
        planes, bp = pack_linear_fp16x2(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
        planes = planes.view(-1, 2, 512).clone()
        planes[:, 0] *= 2048.0                                  # exact: a power of two, |w| < 31.98
        return SinglePlanes(planes.view(-1)), bp * 2048.0
    if mode == "bf16x6":
        return pack_linear_bf16x3(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
    if mode == "bf16":
        return pack_linear_bf16x1(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
    assert mode == "fp32", mode
    return pack_linear(w, b, k_pad=k_pad, n_pad=n_pad, col_map=col_map)
