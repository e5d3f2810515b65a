"""Differentiable wrappers of the HIP ops used by the MAgNet models (training of MAgNet[GNN], BASELINE.json configs[4]).

The reference trains through torch autograd over ATen / PyG ops.  Here every arithmetic op of the forward is a HIP
kernel, so each one is wrapped in a ``torch.autograd.Function`` whose backward is again HIP kernels; torch's autograd
only chains them and differentiates the data movement in between (``cat`` / ``permute`` / ``reshape`` / slicing), which
keeps the rollout feedback (``models/magnet_gnn.py:442-475``) differentiable without bespoke glue.

``mlp`` is the workhorse: the fused gather/concat -> N x (Linear + ReLU) -> [LayerNorm] -> store | +residual | CSR-mean
kernel.  Its forward is the inference kernel; when the backward's tape (assembled first-layer input, hidden activations,
pre-LayerNorm output) fits in memory the same launch writes it (``_keep_tape``), else the backward recomputes the chain with one
tape-only launch.  The backward then walks the chain back:

* LayerNorm backward (``magnet_layernorm_bwd_amax_f32``: 16-byte accesses, gathers dy[tgt] / deg itself behind a CSR mean, leaves the
  block maxima of its output for the next kernel);
* hidden W -> W Linears (W = 128 / 64) over many rows in ONE kernel each - data gradient, ReLU mask, weight and bias gradient
  (``ops.linear_bwd128``); other Linears: a backward-data GEMM with transposed packs in the bf16x6 split (full fp32 exponent range
  for ~1e-7 gradients) with the ReLU mask in its epilogue + ``magnet_wgrad_tiles_f32``, equal-width column blocks in one launch;
* the first Linear: k tiles that are pure row gathers are walked back at source-row granularity (dz summed over each source row's
  gatherers first, ``_first_layer_at_source_rows``), the rest at row granularity followed by the scatter-add through the gathers
  (``magnet_segment_sum_cols_f32`` / ``_runs_f32`` over the inverse CSR of each gather index, a column copy for dense sources).
"""
from __future__ import annotations

import ctypes as C
import os
import types

import torch

from . import _lib as L
from . import ops
from .packing import pack, pad_to

BWD_GEMM = "bf16x6"
# MAGNET_DIRECT_SRC_GRAD=0 (development A/B): dense-source gradients through dX0 + column copies + torch adds again
DIRECT_SRC_GRAD = __import__("os").environ.get("MAGNET_DIRECT_SRC_GRAD", "1") != "0"
# MAGNET_ACCUM_PARAM_GRADS=0 (development A/B): see accumulate_param_grads
ACCUM_PARAM_GRADS = __import__("os").environ.get("MAGNET_ACCUM_PARAM_GRADS", "1") != "0"
_ACCUM = {"on": False}


def in_ddp_forward() -> bool:
    """True while a ``DistributedDataParallel`` wrapper is running its module's forward (Lightning's DDP strategy routes
    ``training_step`` through it): the reducer's hooks sit on the AccumulateGrad nodes and must see every parameter gradient."""
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        return DDP._get_active_ddp_module() is not None
    except Exception:                                   # no distributed build: no DDP
        return False


def takes_inplace_grad(p) -> bool:
    """a leaf parameter whose gradient may be added to ``p.grad`` behind autograd's back: nothing hangs on it that would notice"""
    return (p.is_leaf and p.requires_grad and not getattr(p, "_backward_hooks", None)
            and not getattr(p, "_post_accumulate_grad_hooks", None))


class accumulate_param_grads:
    """``with accumulate_param_grads():`` around the forward of a training step that will be differentiated with ``loss.backward()``
    (the models' ``training_step``): the chains built inside let their one-kernel Linear backward (``ops.linear_bwd128``) ADD the weight
    and bias gradient of a leaf parameter to ``p.grad`` in the kernel (the first contribution of a step becomes ``p.grad``) and hand
    autograd None for it.  A rollout calls every chain once per forward, so without this torch's AccumulateGrad adds four [128, 128]
    gradients per parameter one tiny kernel at a time: 893 ``aten::add_`` per magnet_gnn step at the reference's batch size.  Not for
    ``torch.autograd.grad`` (which must not touch ``.grad``): outside this context the Functions return every gradient as before.

    A gradient that never passes through autograd's AccumulateGrad fires none of the hooks that hang on it: DistributedDataParallel's
    reducer (Lightning ``strategy='ddp'``, the reference's multi-GPU mode, README.md:124-125), ``Tensor.register_hook`` and
    ``register_post_accumulate_grad_hook``.  So the scope switches itself OFF inside a DDP forward (``in_ddp_forward``), and a
    parameter that carries a hook is always handed to autograd (``takes_inplace_grad``)."""

    def __enter__(self):
        self.prev = _ACCUM["on"]
        _ACCUM["on"] = ACCUM_PARAM_GRADS and not in_ddp_forward()
        return self

    def __exit__(self, *exc):
        _ACCUM["on"] = self.prev
        return False


# ------------------------------------------------------------------------------------------------ helpers
_INV_CACHE: dict = {}          # (data_ptr, numel, n_rows) -> (idx kept alive, rowptr, perm); gather indices belong to cached topologies


def _inverse_csr(idx: torch.Tensor, n_rows: int, cache: dict = None, drop_last: bool = False):
    """CSR over the *values* of a gather index: rows = gathered-from rows, perm = gathering positions.  Built once per
    index tensor (the sort inside csr_by_target was a third of the first training profile): the cache keeps the index
    tensor alive, so its address cannot be reused for different contents.  ``drop_last``: the last source row is a constant
    padding row (the zero row every out-of-bounds tap of a convolution points at): its CSR row is emptied - nobody wants that
    gradient, and as the one long row (a border's worth of entries walked by a single wave) it was the whole duration of the
    scatter kernel."""
    key = (idx.data_ptr(), idx.numel(), n_rows, drop_last)
    hit = _INV_CACHE.get(key)
    if hit is None:
        pos = torch.arange(idx.numel(), dtype=torch.int32, device=idx.device)
        c = ops.csr_by_target(pos, idx, n_rows)
        rowptr = c.rowptr
        if drop_last:
            rowptr = rowptr.clone()
            rowptr[n_rows] = rowptr[n_rows - 1]
        if len(_INV_CACHE) >= 64:
            _INV_CACHE.pop(next(iter(_INV_CACHE)))
        hit = _INV_CACHE[key] = (idx, rowptr, c.perm)
    return hit[1], hit[2]


def _segment_sum_cols(x, xcol, rowptr, perm, n_rows, width, scale, out, ocol):
    L.check(L.lib().magnet_segment_sum_cols_f32(ops._p(x), x.stride(0), xcol, ops._p(rowptr), ops._p(perm), n_rows, width,
                                                C.c_float(scale), ops._p(out), out.stride(0), ocol, 1, ops._stream()),
            "segment_sum_cols")


_SEG_RUNS = 9              # csrc/magnet_train_kernels.hip: MAGNET_SEG_RUNS


def _segment_sum_cols_runs(x, runs, c0, n_rows, width, scale, out, ocol):
    """out[:, ocol:ocol+width] += scale * sum over ``runs`` = [(rowptr, perm, first column of x)] of the by-row sums: one launch for up to
    nine gathers of the same rows (accumulates, like _segment_sum_cols)"""
    n = len(runs)
    rp = (C.c_void_p * n)(*[r[0].data_ptr() for r in runs])
    pm = (C.c_void_p * n)(*[(r[1].data_ptr() if r[1] is not None else None) for r in runs])
    xc = (C.c_int32 * n)(*[int(r[2]) + c0 for r in runs])
    L.check(L.lib().magnet_segment_sum_cols_runs_f32(ops._p(x), x.stride(0), rp, pm, xc, n, n_rows, width, C.c_float(scale), ops._p(out),
                                                     out.stride(0), ocol, 1, ops._stream()), "segment_sum_cols_runs")


def _pad_cols(t, width):
    if t.shape[1] == width and t.is_contiguous():
        return t
    out = torch.zeros(t.shape[0], width, device=t.device, dtype=torch.float32)
    ops.copy_cols(t.contiguous(), 0, out, 0, t.shape[1])
    return out


def _linear(rows, x, layer, out_tiles, relu):
    """one Linear (+ ReLU) of the chain on dense rows x [rows, K]"""
    y = torch.empty(rows, out_tiles * 32, device=x.device, dtype=torch.float32)
    ops.mlp(rows, ops.ktiles(ops.Term(x, x.shape[1])), [layer], nl=1, ht=out_tiles, ot=out_tiles,
            act=ops.ACT_RELU if relu else ops.ACT_NONE, last_act=relu, y=y)
    return y


def _tape_variant_ok(nl, pk):
    """is there a store-epilogue instantiation of the fused kernel for this chain without LayerNorm? (nl, ht, ot, ReLU, no
    last_act, no LN, STORE): the projector / decoder shapes (5, 4, 1), (5, 2, 1) exist; the LayerNorm chains use their own
    (5, ht, ot, LN) variants with the LayerNorm switched on and its output discarded."""
    return bool(L.query("magnet_mlp_supported", nl, pk.ht, pk.ot, ops.ACT_RELU, 0, 0, ops.EPI_STORE))


def _recompute_tape(M, kts, pk, nl, x0, X, zL, K0, x0_skip=0):
    """fill the backward's tape - x0 (assembled first-layer input), X[1:] (activated output of every hidden layer) and zL (the
    last Linear's raw output, LayerNorm chains only) - by re-running the forward:

    1. ONE launch of the chain kernel's plain store variant with its tape outputs switched on, when that instantiation exists;
    2. else, for a LayerNorm chain, ONE launch of its own (nl, ht, ot, LN) variant (the normalised output goes to a scratch array);
    3. else layer by layer: a materialisation pass for x0, then one single-Linear launch per layer (any shape the library can
       run forward can be recomputed this way)."""
    dev = X[1].device if x0 is None else x0.device          # (x0 = None: the caller's backward reads the sources instead, _x0_not_needed)
    Dp = pk.ot * 32
    lnflag = 1 if pk.ln is not None else 0
    tape_only = pk.layers[0][0].dtype in (torch.float16, torch.int16)      # y = None: the chain's own output is not needed, not written
    scratch = lambda: None if tape_only else torch.empty(max(M, 1), Dp, device=dev, dtype=torch.float32)  # noqa: E731
    if _tape_variant_ok(nl, pk):
        ops.mlp(M, kts, pk.layers, nl=nl, ht=pk.ht, ot=pk.ot, act=ops.ACT_RELU, last_act=False, ln=None, epi=ops.EPI_STORE,
                y=scratch(), ncols=Dp, x0_out=x0, hid=X[1:] + [zL], x0_skip=x0_skip)
    elif lnflag and L.query("magnet_mlp_supported", nl, pk.ht, pk.ot, ops.ACT_RELU, 0, 1, ops.EPI_STORE):
        ops.mlp(M, kts, pk.layers, nl=nl, ht=pk.ht, ot=pk.ot, act=ops.ACT_RELU, last_act=False, ln=pk.ln, epi=ops.EPI_STORE,
                y=scratch(), ncols=Dp, x0_out=x0, hid=X[1:] + [zL], x0_skip=x0_skip)
    else:
        a, keep = ops.mlp_args(M, kts)
        L.check(L.lib().magnet_mlp_materialize_input_f32(C.byref(a), ops._p(x0), K0, ops._stream()), "materialize_input")
        for l in range(nl - 1):
            X[l + 1].copy_(_linear(M, X[l], pk.layers[l], pk.ht, relu=True))
        if zL is not None:
            zL.copy_(_linear(M, X[nl - 1], pk.layers[nl - 1], pk.ot, relu=False))


_MEM = {"t": -1.0, "dev": None, "free": 0, "total": 1, "granted": 0}


def _keep_tape(nbytes, dev):
    """does the tape of one chain stay in memory from the forward to the backward?  MAGNET_KEEP_TAPE = 1 / 0 / auto (default): yes while
    it takes less than a quarter of what is still free and the device stays below 70 % full (MAGNET_TAPE_CAP) - later calls of a long rollout then fall
    back to recomputing by themselves.  "Free" is what the DEVICE reports (other processes, RCCL buffers, captured graphs' pools count)
    plus what torch's allocator holds in reserve; the figures are refreshed at most every 50 ms - a training step at the reference's
    batch sizes asks ~100 times and torch.cuda.memory_stats costs 0.1 ms a call (a fifth of such a step) - and the bytes granted since
    the last refresh are subtracted."""
    import os
    import time
    mode = os.environ.get("MAGNET_KEEP_TAPE", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    now = time.perf_counter()
    if _MEM["dev"] != dev or now - _MEM["t"] > 0.05:
        free_dev, total = torch.cuda.mem_get_info(dev)
        _MEM.update(t=now, dev=dev, total=total, granted=0,
                    free=free_dev + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
    free = _MEM["free"] - _MEM["granted"]
    used = _MEM["total"] - free
    ok = nbytes < 0.25 * free and used + nbytes < float(os.environ.get("MAGNET_TAPE_CAP", "0.7")) * _MEM["total"]
    if ok:
        _MEM["granted"] += nbytes
    return ok


def _tiles_of(width):
    """decompose a padded width into launchable output widths (128 / 64 / 32 columns)"""
    out, c = [], 0
    while c < width:
        w = 128 if width - c >= 128 else (64 if width - c >= 64 else 32)
        out.append((c, w))
        c += w
    return out


def _groups_of_ktiles(kts):
    """sorted k-tile indices -> [(first column, block width, number of blocks)]: maximal runs cut into blocks of at most four k
    tiles, equal-width neighbours merged (as _tile_groups does for a whole width)"""
    out, i = [], 0
    while i < len(kts):
        j = i
        while j + 1 < len(kts) and kts[j + 1] == kts[j] + 1:
            j += 1
        c, left = kts[i] * 32, (j - i + 1) * 32
        while left > 0:
            w = 128 if left >= 128 else (64 if left >= 64 else 32)
            if out and out[-1][1] == w and out[-1][0] + out[-1][1] * out[-1][2] == c and out[-1][2] < L.MAX_LAYERS:
                out[-1] = (out[-1][0], w, out[-1][2] + 1)
            else:
                out.append((c, w, 1))
            c, left = c + w, left - w
        i = j + 1
    return out


def _factored_first_layer(m, srcs, M):
    """k tiles of the first Linear whose input is a pure row gather (every Term covering them has an index, rdiv = 1) of far fewer
    source rows than M: there d(source) = sum over the gathering rows of dz W and dW = sum of dz^T x commute with the gather -
    dz is summed over each source row's gatherers FIRST ([M, Np] -> [n_src, Np]) and the GEMMs run at source granularity.  The
    backward twin of the forward's factored first layer.  Returns {k tile: True} for those tiles."""
    nkt = len(m.tiles)
    ok = [True] * nkt
    seen = [False] * nkt
    for (si, col, idx, scale, rdiv, kt0, width) in m.runs:
        rows = srcs[si].reshape(-1, srcs[si].shape[-1]).shape[0]
        good = idx is not None and rdiv == 1 and width % 32 == 0 and 2 * rows <= M
        for kt in range(kt0, kt0 + width // 32):
            seen[kt] = True
            ok[kt] = ok[kt] and good
    ok = [ok[kt] and seen[kt] for kt in range(nkt)]
    # the callers decide per RUN from its first k tile (source-row walk, x0 tape skip, column groups): a run is factored only when ALL
    # of its k tiles are - one tile shared with a dense Term un-factors the whole run (and every run that shares a tile with it)
    changed = True
    while changed:
        changed = False
        for (si, col, idx, scale, rdiv, kt0, width) in m.runs:
            tiles = range(kt0, kt0 + max(width // 32, 1))
            if any(ok[kt] for kt in tiles if kt < nkt) and not all(ok[kt] for kt in tiles if kt < nkt):
                for kt in tiles:
                    if kt < nkt:
                        ok[kt] = False
                changed = True
    return ok


def _x0_skip_mask(m, srcs, M, np_):
    """bit kt set: the backward walks k tile kt of the first Linear back at source-row granularity (_factored_first_layer) and never
    reads that k tile of the assembled input - the tape does not have to hold it"""
    if M < _FUSED_LAYER_ROWS or np_ not in (32, 64, 128):
        return 0
    return sum(1 << kt for kt, f in enumerate(_factored_first_layer(m, srcs, M)) if f)


def _x0_not_needed(m, srcs, M, np_, fk=None):
    """-> {k tile: (source index, source column, scale)} when the backward can do WITHOUT the assembled first-layer input x0, else None.
    Every k tile of the first Linear must be either factored (walked back at source-row granularity: _factored_first_layer) or ONE dense
    Term (no index, rdiv = 1, a source of exactly M rows): its columns of x0 are a scaled copy of columns the source tensor holds
    anyway, and dW[:, tile] = scale dz^T source[:, columns] (dW is scaled AFTER the reduction: the same bits as scaling x0 before it
    only for the power-of-two scales MAgNet's chains use - else equal to rounding).  MAgNet's edge chains are all of this kind ([x_i | x_j | e] gathered +
    dense, or the factored P / Q products + e): their x0 was a quarter (C4b) / a sixth (C5) of the whole tape (round 5)."""
    # (only where it pays: a step at the reference's batch sizes is host bound and its whole tape is a few GB - there the extra launches
    #  and the Python of this path cost more than the memory is worth.  MAGNET_X0_FREE_MIN_MB: x0 of at least this size, default 256;
    #  MAGNET_X0_FREE=0: never)
    if os.environ.get("MAGNET_X0_FREE", "1") == "0" or 4 * M * len(m.tiles) * 32 < float(os.environ.get("MAGNET_X0_FREE_MIN_MB", "256")) * 2 ** 20:
        return None
    if fk is None:
        fk = _factored_first_layer(m, srcs, M) if (M >= _FUSED_LAYER_ROWS and np_ in (32, 64, 128)) else [False] * len(m.tiles)
    dense = {}
    for kt, tile in enumerate(m.tiles):
        if fk[kt]:
            continue
        if len(tile) != 1:
            return None
        si, col, idx, scale, rdiv = tile[0]
        s2 = srcs[si]
        if idx is not None or rdiv != 1 or s2.dtype != torch.float32 or s2.dim() < 2 or s2.reshape(-1, s2.shape[-1]).shape[0] != M or not s2.is_contiguous() \
                or col + 32 > s2.shape[-1]:
            return None
        dense[kt] = (si, col, float(scale))
    return dense


def _tile_groups(width):
    """_tiles_of with runs of equal-width blocks merged: (first column, block width, number of blocks)"""
    out = []
    for c0, w in _tiles_of(width):
        if out and out[-1][1] == w and out[-1][0] + out[-1][1] * out[-1][2] == c0 and out[-1][2] < L.MAX_LAYERS:   # (one layer slot per group)
            out[-1] = (out[-1][0], w, out[-1][2] + 1)
        else:
            out.append((c0, w, 1))
    return out


def _first_layer_at_source_rows(ctx, m, pk, srcs, params, dz, fk, dWp, dbp, Np, Kp, dev):
    """the factored k tiles of the first Linear (see _factored_first_layer): per gathered run S = the sum of dz over the rows that
    gathered each source row ([n_src, Np]; by the CSR's own target: contiguous segments), then dW[:, tile] (+)= scale S^T src and
    d src (+)= scale S W[:, tile] as launches over n_src rows.  Writes dWp's factored columns (and dbp = column sums of dz when asked);
    returns the per-source gradient buffers it created (None for sources without a factored run)."""
    W = params[0]
    M = dz.shape[0]
    sums, first_of_kt, grads = {}, {}, [None] * m.n_src
    ws = None
    for (si, col, idx, scale, rdiv, kt0, width) in m.runs:
        if not fk[kt0]:
            continue
        src2 = srcs[si].reshape(-1, srcs[si].shape[-1])
        n_src = src2.shape[0]
        key = idx.data_ptr()
        if key not in sums:                                              # sum of dz over the gatherers of every source row
            S = torch.empty(n_src, Np, device=dev, dtype=torch.float32)
            if m.tgt is not None and key == m.tgt.data_ptr():
                rowptr, perm = m.rowptr, None
            else:
                rowptr, perm = _inverse_csr(idx, n_src, m.inv_cache)
            L.check(L.lib().magnet_segment_sum_cols_f32(ops._p(dz), dz.stride(0), 0, ops._p(rowptr), ops._p(perm), n_src, Np, C.c_float(1.0),
                                                        ops._p(S), Np, 0, 0, ops._stream()), "segment_sum_cols")
            sums[key] = S
            if dbp is not None and not first_of_kt:                       # db = column sums of dz = column sums of any complete S
                ops.reduce_partials(S, n_src, Np, dbp, accumulate=False)
        S = sums[key]
        Ss = S if scale == 1.0 else S * scale
        nblk = min(768, (n_src + 31) // 32)
        for (c0, w, G) in _tile_groups(width):
            need = G * nblk * Np * w + nblk * Np + G * Np * w
            if ws is None or ws.numel() < need:
                ws = torch.empty(need, device=dev, dtype=torch.float32)
            acc = 1 if first_of_kt.get(kt0 + c0 // 32) else 0             # the second Term of a differenced pair adds to the first's columns
            L.check(L.lib().magnet_wgrad_tiles_f32(ops._p(Ss), Np, Np, ops._p(src2), src2.stride(0), col + c0, w, G, C.c_long(n_src),
                                                   ops._p(dWp), Kp, kt0 * 32 + c0, None, ops._p(ws), acc, ops._stream()), "wgrad_tiles")
        for kt in range(kt0, kt0 + width // 32):
            first_of_kt[kt] = True
        if ctx.needs_input_grad[1 + si]:
            if grads[si] is None:
                grads[si] = torch.zeros(srcs[si].shape, device=dev, dtype=torch.float32)
            g2 = grads[si].view(-1, grads[si].shape[-1])
            tmp = torch.empty(n_src, width, device=dev, dtype=torch.float32)
            for (c0, w, G) in _tile_groups(width):
                tps = [_tpack(m, pk, 0, kt0 * 32 + c0 + g * w, w, W, Np, Kp) for g in range(G)]
                ops.mlp(n_src, ops.ktiles(ops.Term(Ss, Np)), tps, nl=1, ht=w // 32, ot=w // 32, act=ops.ACT_NONE, colgroups=G,
                        y=tmp[:, c0:], ldy=width, ncols=w)
            ops.copy_cols(tmp, 0, g2, col, width, accumulate=True)
    return grads


# ------------------------------------------------------------------------------------------------ fused MLP
class _Mlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, *tensors):
        m = meta
        srcs = tensors[:m.n_src]
        res = tensors[-1] if m.has_res else None
        kts = [[(ops.Term(srcs[si], 32, idx=idx, scale=scale, rdiv=rdiv), col) for (si, col, idx, scale, rdiv) in tile]
               for tile in m.tiles]
        pk = m.pk
        y = torch.empty(m.out_rows, pk.out_dim, device=srcs[0].device, dtype=torch.float32)
        seg = {}
        if m.epi == ops.EPI_SEGMEAN:
            seg = dict(tgt=m.tgt, rowptr=m.rowptr,
                       partials=torch.empty(((m.M + 31) // 32) * 2 * pk.ot * 32 + 1, device=y.device, dtype=torch.float32))
        nl = len(pk.layers)
        # the backward's tape (assembled first-layer input, every hidden activation, the last Linear's raw output under a LayerNorm):
        # written by THIS launch when it fits in memory (288 GB per GPU: a C5 training rollout keeps ~150 GB of it), else recomputed by
        # one more launch of the chain in the backward
        tape = ()
        K0 = len(m.tiles) * 32
        wants_tape = nl > 1 and not m.final_relu and pk.layers[0][0].dtype in (torch.float16, torch.int16) and m.M > 0 and \
            any(ctx.needs_input_grad[1:])
        if wants_tape:
            # (no x0 at all when every k tile is factored or a dense copy of a source: an empty placeholder keeps the tuple's layout)
            np0 = pk.ht * 32
            fk = _factored_first_layer(m, srcs, m.M) if (m.M >= _FUSED_LAYER_ROWS and np0 in (32, 64, 128)) else [False] * len(m.tiles)
            m.x0_dense = _x0_not_needed(m, srcs, m.M, np0, fk)        # the decision travels with the tape (not re-derived from the
            x0_free = m.x0_dense is not None                          # environment in the backward)
            K0t = 0 if x0_free else K0
        if wants_tape and _keep_tape(4 * m.M * (K0t + (nl - 1) * pk.ht * 32 + pk.ot * 32), y.device):
            new = lambda w: torch.empty(m.M, w, device=y.device, dtype=torch.float32)  # noqa: E731
            tape = (new(K0t), *[new(pk.ht * 32) for _ in range(nl - 1)], *([new(pk.ot * 32)] if pk.ln is not None else []))
            seg = dict(seg, x0_out=None if x0_free else tape[0], hid=list(tape[1:nl]) + [tape[nl] if pk.ln is not None else None],
                       x0_skip=sum(1 << kt for kt, f in enumerate(fk) if f))
        ops.mlp(m.M, kts, pk.layers, nl=nl, ht=pk.ht, ot=pk.ot, act=ops.ACT_RELU if (nl > 1 or m.final_relu) else ops.ACT_NONE,
                last_act=m.final_relu, ln=pk.ln, epi=m.epi, y=y, res=res, ncols=pk.out_dim, **seg)
        if m.epi == ops.EPI_SEGMEAN:
            ops.segmean_fixup(m.rowptr, m.out_rows, pk.ot * 32, seg["partials"], y)
        ctx.meta = m
        ctx.n_tape = len(tape)
        ctx.save_for_backward(*tensors, *([y] if m.final_relu else []), *tape)
        return y

    @staticmethod
    def backward(ctx, dy):
        m, tensors = ctx.meta, ctx.saved_tensors
        tape = ()
        if ctx.n_tape:
            tape, tensors = tensors[-ctx.n_tape:], tensors[:-ctx.n_tape]
        if m.final_relu:                                                 # y = ReLU(chain): mask the incoming gradient first
            y_out, tensors = tensors[-1], tensors[:-1]
            masked = torch.empty_like(y_out)
            L.check(L.lib().magnet_relu_bwd_f32(ops._p(dy.contiguous().float()), ops._p(y_out), ops._p(masked),
                                                C.c_long(y_out.numel()), ops._stream()), "relu_bwd")
            dy = masked
        pk = m.pk
        srcs = tensors[:m.n_src]
        params = tensors[m.n_src:m.n_src + m.n_par]
        dev = dy.device
        nl, M = len(pk.layers), m.M
        Dp, D = pk.ot * 32, pk.out_dim
        dy = dy.contiguous().float()
        with torch.no_grad():
            # ---- epilogue
            d_res = dy if m.has_res else None
            mean_in_ln = m.epi == ops.EPI_SEGMEAN and pk.ln is not None and dy.shape[1] == Dp      # the LayerNorm backward gathers dy[tgt] / deg itself
            if mean_in_ln:
                d_rows = None
            elif m.epi == ops.EPI_SEGMEAN:
                d_rows = torch.empty(max(M, 1), Dp, device=dev, dtype=torch.float32)
                L.check(L.lib().magnet_edge_mean_bwd_cols_f32(ops._p(dy), ops._p(m.tgt), ops._p(m.rowptr), C.c_long(M), Dp,
                                                              ops._p(d_rows), Dp, ops._stream()), "edge_mean_bwd_cols")
            else:
                d_rows = _pad_cols(dy, Dp)
            # ---- recompute the chain in ONE launch of the forward kernel with its tape outputs switched on: the assembled
            # first-layer input x0, every layer's activated output and the last Linear's raw output z_L (before LayerNorm) come
            # out of the registers the chain runs in (round 1: a materialisation pass + one single-Linear launch per layer, each
            # re-reading and re-writing an [M, width] array).  Multi-layer chains only; a single Linear needs x0 alone.
            K0 = len(m.tiles) * 32
            kts = [[(ops.Term(srcs[si], 32, idx=idx, scale=scale, rdiv=rdiv), col) for (si, col, idx, scale, rdiv) in tile]
                   for tile in m.tiles]
            grads_par = [None] * m.n_par
            zL = None
            x0_dense = None                                               # {k tile: (source, column, scale)} when x0 does not exist
            if tape:                                                      # kept by the forward launch
                x0, X = tape[0], list(tape[:nl])
                zL = tape[nl] if pk.ln is not None else None
                if x0.numel() == 0 and M > 0:
                    x0_dense = m.x0_dense                                 # what the forward decided when it left x0 out of the tape
                    assert x0_dense is not None
            elif nl > 1:
                one_launch = _tape_variant_ok(nl, pk) or (pk.ln is not None and bool(
                    L.query("magnet_mlp_supported", nl, pk.ht, pk.ot, ops.ACT_RELU, 0, 1, ops.EPI_STORE)))
                x0_dense = _x0_not_needed(m, srcs, M, pk.ht * 32) if (one_launch and M > 0) else None
                x0 = None if x0_dense is not None else torch.empty(max(M, 1), K0, device=dev, dtype=torch.float32)
                X = [x0] + [torch.empty(max(M, 1), pk.ht * 32, device=dev, dtype=torch.float32) for _ in range(nl - 1)]
                zL = torch.empty(max(M, 1), Dp, device=dev, dtype=torch.float32) if pk.ln is not None else None
                _recompute_tape(M, kts, pk, nl, x0, X, zL, K0, x0_skip=_x0_skip_mask(m, srcs, M, pk.ht * 32))
            else:
                x0 = torch.empty(max(M, 1), K0, device=dev, dtype=torch.float32)
                a, keep = ops.mlp_args(M, kts)
                L.check(L.lib().magnet_mlp_materialize_input_f32(C.byref(a), ops._p(x0), K0, ops._stream()), "materialize_input")
                X = [x0]
                if pk.ln is not None:
                    zL = _linear(M, X[-1], pk.layers[nl - 1], pk.ot, relu=False)
            if pk.ln is not None:
                nblk = min(2048, (M + 3) // 4)
                part = torch.empty(nblk, 2, D, device=dev, dtype=torch.float32)
                # (the kernel writes the D real columns of every row; padding columns, if any, must read as zero)
                dz = (torch.empty if D == Dp and M > 0 else torch.zeros)(max(M, 1), Dp, device=dev, dtype=torch.float32)
                ln_amax = torch.empty(nblk, device=dev, dtype=torch.float32) if D == Dp else None   # block maxima of |dz| for k_linear_bwd
                L.check(L.lib().magnet_layernorm_bwd_amax_f32(ops._p(zL), Dp, ops._p(dy if mean_in_ln else d_rows), Dp, ops._p(pk.ln[0]),
                                                              C.c_long(M), D, C.c_float(pk.ln[2]), ops._p(dz), Dp, ops._p(part),
                                                              ops._p(ln_amax), ops._p(m.tgt if mean_in_ln else None),
                                                              ops._p(m.rowptr if mean_in_ln else None), ops._stream()), "layernorm_bwd")
                gb = torch.zeros(2 * D, device=dev, dtype=torch.float32)
                ops.reduce_partials(part, nblk, 2 * D, gb)
                grads_par[2 * nl], grads_par[2 * nl + 1] = gb[:D], gb[D:]
            else:
                dz, ln_amax = d_rows, None
            # ---- walk the Linears back
            need_dx0 = any(ctx.needs_input_grad[1 + i] for i in range(m.n_src))
            direct = {}                                                       # source -> its gradient, written by the data GEMM itself
            ws = ops.wgrad_workspace(M, dev)
            nblk_w = min(768, (M + 31) // 32)
            dX0 = None
            amax, lb_ws = ((ln_amax, nblk) if ln_amax is not None else None), None
            tpacks_emit_amax = BWD_GEMM == "bf16x6"                       # (the split-GEMM kernels' masked store can emit block maxima)
            def fused_ok(l_, Np_):
                """layer l_ (W [Np_][Np_], input X[l_]) can be walked back by csrc/linear_bwd.hip"""
                W_ = params[2 * l_]
                return (l_ > 0 and tuple(W_.shape) == (Np_, Np_) and X[l_].shape[1] == Np_ and Np_ in (128, 64) and M >= _FUSED_LAYER_ROWS
                        and pk.layers[0][0].dtype == torch.float16 and X[l_].is_contiguous())

            def grad_dest(W_, b_, Np_):
                """(dW, db, accumulate, into): the parameter's own .grad when accumulate_param_grads lets the kernel's reduction add to it"""
                # (a frozen Linear - requires_grad=False tensors are leaves too - must not get a .grad from here: autograd drops
                #  what backward returns for it, an optimizer built over self.parameters() would not drop an assigned .grad)
                into_ = (m.accum and takes_inplace_grad(W_) and takes_inplace_grad(b_)
                         and W_.dtype == b_.dtype == torch.float32 and tuple(W_.shape) == (Np_, Np_) and tuple(b_.shape) == (Np_,))
                Wg_, bg_ = (W_.grad, b_.grad) if into_ else (None, None)
                if (Wg_ is not None and bg_ is not None and Wg_.dtype == bg_.dtype == torch.float32 and Wg_.device == bg_.device == dev
                        and Wg_.is_contiguous() and bg_.is_contiguous()):
                    return Wg_, bg_, True, into_
                return (torch.empty(Np_, Np_, device=dev, dtype=torch.float32), torch.empty(Np_, device=dev, dtype=torch.float32), False, into_)

            def grad_place(l_, W_, b_, dWp_, dbp_, accumulated, into_):
                if accumulated:
                    return
                if into_ and W_.grad is None and b_.grad is None:
                    W_.grad, b_.grad = dWp_, dbp_                       # the step's first contribution IS the gradient buffer
                else:
                    grads_par[2 * l_], grads_par[2 * l_ + 1] = dWp_, dbp_

            skip = False
            for l in range(nl - 1, -1, -1):
                if skip:                                                  # (walked back together with the layer above it)
                    skip = False
                    continue
                W, b = params[2 * l], params[2 * l + 1]
                n, k = W.shape
                Np, Kp = dz.shape[1], (K0 if (l == 0 and x0_dense is not None) else X[l].shape[1])
                # two consecutive hidden 128 -> 128 (64 -> 64) Linears in ONE kernel: the gradient between them never leaves the chip (four [M, W]
                # array passes instead of six; magnet_linear_bwd128_pair_f32)
                if (_LINEAR_BWD_PAIR and Np in (128, 64) and n == k == Kp == Np and dz.is_contiguous() and fused_ok(l, Np) and fused_ok(l - 1, Np)):
                    Wb, bb = params[2 * (l - 1)], params[2 * (l - 1) + 1]
                    dA_, dbA_, accA, intoA = grad_dest(W, b, Np)
                    dB_, dbB_, accB, intoB = grad_dest(Wb, bb, Np)
                    if accA == accB:
                        if lb_ws is None:
                            lb_ws = torch.empty(L.query("magnet_linear_bwd128_workspace"), device=dev, dtype=torch.float32)
                        dX = torch.empty(M, Np, device=dev, dtype=torch.float32)
                        amax = ops.linear_bwd128_pair(dz, X[l], X[l - 1], _tpack16(pk, l, W), _tpack16(pk, l - 1, Wb), dX, dA_, dbA_, dB_, dbB_,
                                                      lb_ws, amax_in=amax, accumulate=accA)
                        grad_place(l, W, b, dA_, dbA_, accA, intoA)
                        grad_place(l - 1, Wb, bb, dB_, dbB_, accB, intoB)
                        dz = dX
                        skip = True
                        continue
                # a hidden 128 -> 128 (or 64 -> 64) Linear over many rows: data gradient + ReLU mask + weight and bias gradients in ONE kernel
                # (csrc/linear_bwd.hip: dz and X[l] are read once instead of twice, no partial-sum passes over [M, 128] arrays)
                if (l > 0 and n == k == Np == Kp and Np in (128, 64) and M >= _FUSED_LAYER_ROWS and pk.layers[0][0].dtype == torch.float16
                        and dz.is_contiguous() and X[l].is_contiguous()):
                    if lb_ws is None:
                        lb_ws = torch.empty(L.query("magnet_linear_bwd128_workspace"), device=dev, dtype=torch.float32)
                    dX = torch.empty(M, Np, device=dev, dtype=torch.float32)
                    # (a frozen Linear - requires_grad=False tensors are leaves too - must not get a .grad from here: autograd drops
                    #  what backward returns for it, an optimizer built over self.parameters() would not drop an assigned .grad)
                    into = (m.accum and takes_inplace_grad(W) and takes_inplace_grad(b)
                            and W.dtype == b.dtype == torch.float32 and tuple(W.shape) == (Np, Np) and tuple(b.shape) == (Np,))
                    Wg, bg = (W.grad, b.grad) if into else (None, None)
                    if (Wg is not None and bg is not None and Wg.dtype == bg.dtype == torch.float32 and Wg.device == bg.device == dev
                            and Wg.is_contiguous() and bg.is_contiguous()):
                        # (accumulate_param_grads: this call's share is added to p.grad by the kernel's own reduction)
                        amax = ops.linear_bwd128(dz, X[l], _tpack16(pk, l, W), dX, Wg, bg, lb_ws, mask=True, amax_in=amax, accumulate=True)
                    else:
                        dWp, dbp = torch.empty(Np, Np, device=dev, dtype=torch.float32), torch.empty(Np, device=dev, dtype=torch.float32)
                        amax = ops.linear_bwd128(dz, X[l], _tpack16(pk, l, W), dX, dWp, dbp, lb_ws, mask=True, amax_in=amax)
                        if into and W.grad is None and b.grad is None:
                            W.grad, b.grad = dWp, dbp                         # the step's first contribution IS the gradient buffer
                        else:
                            grads_par[2 * l], grads_par[2 * l + 1] = dWp, dbp
                    dz = dX
                    continue
                amax = None
                # the column tiles cover all Kp columns and all Np rows: written, not accumulated - no zero fill needed
                alloc = torch.empty if M > 0 else torch.zeros           # (no rows: the kernel returns without writing)
                dWp = alloc(Np, Kp, device=dev, dtype=torch.float32)
                dbp = alloc(Np, device=dev, dtype=torch.float32)
                col_groups = _tile_groups(Kp)
                factored = None
                if l == 0 and M >= _FUSED_LAYER_ROWS and Np in (32, 64, 128):
                    fk = _factored_first_layer(m, srcs, M)
                    if any(fk):
                        factored = fk
                        col_groups = _groups_of_ktiles([kt for kt in range(Kp // 32) if not fk[kt]])
                        grads_src_f = _first_layer_at_source_rows(ctx, m, pk, srcs, params, dz, fk, dWp, dbp if not col_groups else None,
                                                                  Np, Kp, dev)
                for j, (c0, w, G) in enumerate(col_groups):                    # equal-width column blocks: one launch per kernel
                    need = G * nblk_w * Np * w + nblk_w * Np + G * Np * w
                    if ws.numel() < need:
                        ws = torch.empty(need, device=dev, dtype=torch.float32)
                    if l == 0 and x0_dense is not None:
                        # no assembled input: these columns are a scaled copy of a source's - read the source (one launch per run of
                        # consecutive columns of one source, else per k tile), scale the block of dW afterwards
                        # (block by block of the SAME width w as the launch over x0 had: the same kernel instantiation and the same order
                        #  of partial sums, i.e. the same bits; a block that is not w consecutive columns of one source: k tile by k tile)
                        spans = []
                        for g_ in range(G):
                            kts_b = list(range((c0 + g_ * w) // 32, (c0 + (g_ + 1) * w) // 32))
                            si0, col0, sc0 = x0_dense[kts_b[0]]
                            if all(x0_dense[kt] == (si0, col0 + 32 * i, sc0) for i, kt in enumerate(kts_b)):
                                spans.append((c0 + g_ * w, w, 1, si0, col0, sc0))
                            else:
                                spans += [(32 * kt, 32, 1, *x0_dense[kt]) for kt in kts_b]
                        for q, (cc, ww, GG, si_, col_, sc_) in enumerate(spans):
                            xs = srcs[si_].reshape(-1, srcs[si_].shape[-1])
                            L.check(L.lib().magnet_wgrad_tiles_f32(ops._p(dz), Np, Np, ops._p(xs), xs.shape[1], col_, ww, GG, C.c_long(M),
                                                                   ops._p(dWp), Kp, cc, ops._p(dbp if (j == 0 and q == 0) else None), ops._p(ws),
                                                                   0, ops._stream()), "wgrad_tiles")
                            if sc_ != 1.0:
                                dWp[:, cc:cc + ww * GG].mul_(sc_)
                        continue
                    L.check(L.lib().magnet_wgrad_tiles_f32(ops._p(dz), Np, Np, ops._p(X[l]), Kp, c0, w, G, C.c_long(M), ops._p(dWp), Kp,
                                                           c0, ops._p(dbp if j == 0 else None), ops._p(ws), 0, ops._stream()), "wgrad_tiles")
                if l == 0 and m.first_cols is not None:
                    gW = torch.zeros_like(W)
                    gW[:, m.first_cols] = dWp[:n][:, m.col_map]
                elif l == 0 and m.col_map is not None:
                    gW = dWp[:n][:, m.col_map]
                else:
                    gW = dWp[:n, :k]
                grads_par[2 * l], grads_par[2 * l + 1] = gW.contiguous(), dbp[:n].contiguous()
                if l == 0 and not need_dx0:
                    break
                # dX = dz . W   (transposed packs, column block by column block)
                dX = None                                   # (allocated by the first column block that is written to it: with every block of
                                                            #  the first Linear going straight to a source's gradient it is 3 GB nobody reads)
                new_dX = lambda: torch.empty(max(M, 1), Kp, device=dev, dtype=torch.float32)   # noqa: E731
                # column blocks of equal width go out as ONE launch (the kernel's column groups over gridDim.y): the nine 128-column
                # taps of an EDSR convolution on a 64 x 64 map were nine 32-workgroup launches of 14 us each
                # (l > 0: the launch also leaves the block maxima of what it stores - the layer below, if it takes the one-kernel path,
                # needs them for its power-of-two scale and would otherwise make a pass of its own over dX)
                mx = torch.zeros(256, device=dev, dtype=torch.float32) if l > 0 and M >= _FUSED_LAYER_ROWS and tpacks_emit_amax else None
                data_groups = col_groups
                if l == 0:                                                      # blocks the GEMM can write straight into a source's gradient: own launches
                    data_groups = []
                    for (c0, w, G) in col_groups:
                        hits = [g for g in range(G) if DIRECT_SRC_GRAD and _dense_source_of_block(m, srcs, ctx, M, c0 + g * w, w, factored) is not None]
                        g = 0
                        while g < G:
                            g1 = g + 1
                            while g not in hits and g1 < G and g1 not in hits:
                                g1 += 1
                            data_groups.append((c0 + g * w, w, g1 - g))
                            g = g1
                for (c0, w, G) in data_groups:
                    hit = _dense_source_of_block(m, srcs, ctx, M, c0, w, factored) if (l == 0 and G == 1 and DIRECT_SRC_GRAD) else None
                    if hit is not None:
                        # a source that enters as ONE dense full-width run covering exactly this column block (MAgNet's edge / node
                        # latents): its gradient is written by the GEMM itself (a power-of-two Term scale folded into the transposed
                        # pack), with what has to be added to it as the epilogue's residual:
                        #  * dy, when the same tensor is the chain's residual (x' = x + MLP([agg | x]));
                        #  * the running sum of a `grad_chain` (e_0 feeds every hop of the processor scaled by 2^k: the hops' backward
                        #    calls run in reverse order, each adds its block to the sum and only the last hands it to autograd).
                        # Before: the block went to dX0, a (scaled) column copy moved it out, and torch added the pieces - five passes
                        # over an [E, 128] array per hop (7 % of a C5 training step)
                        si, sc = hit
                        g = torch.empty(srcs[si].shape, device=dev, dtype=torch.float32)
                        g2 = g.view(-1, g.shape[-1])
                        res_t = tensors[m.n_src + m.n_par] if m.has_res else None
                        add, chain = None, None
                        if (d_res is not None and res_t is not None and res_t.data_ptr() == srcs[si].data_ptr() and res_t.shape == srcs[si].shape
                                and res_t.stride() == srcs[si].stride() and tuple(dy.shape) == tuple(g2.shape)):
                            add, d_res = dy, None                              # (autograd takes None as zero: the sum is in g)
                        elif m.grad_chain is not None and m.grad_chain[0].data_ptr() == srcs[si].data_ptr() and m.grad_chain[0].shape == srcs[si].shape:
                            chain = m.grad_chain
                            add = chain[1].get("g")
                        ops.mlp(M, ops.ktiles(ops.Term(dz, Np)), [_tpack(m, pk, l, c0, w, W, Np, Kp, scale=sc)], nl=1, ht=w // 32, ot=w // 32,
                                act=ops.ACT_NONE, colgroups=1, y=g2, ldy=g2.stride(0), ncols=w,
                                **(dict(epi=ops.EPI_STORE_RES, res=add.view(-1, add.shape[-1])) if add is not None else {}))
                        direct[si] = g
                        if chain is not None:
                            if chain[2]:
                                chain[1].pop("g", None)                        # (the last call hands the sum to autograd: nothing left to keep)
                            else:
                                chain[1]["g"] = g
                                direct[si] = None                              # (not the last hop of the chain: the sum travels on)
                        continue
                    tps = [_tpack(m, pk, l, c0 + g * w, w, W, Np, Kp) for g in range(G)]
                    if dX is None:
                        dX = new_dX()
                    if l > 0:      # ... with the ReLU mask of the layer below as the epilogue: dz_{l-1} = (X[l] > 0) ? dX : 0
                        ops.mlp(M, ops.ktiles(ops.Term(dz, Np)), tps, nl=1, ht=w // 32, ot=w // 32, act=ops.ACT_NONE, colgroups=G,
                                epi=ops.EPI_STORE_MASK, res=X[l][:, c0:], y=dX[:, c0:], ldy=Kp, ncols=w, partials=mx)
                    else:
                        ops.mlp(M, ops.ktiles(ops.Term(dz, Np)), tps, nl=1, ht=w // 32, ot=w // 32, act=ops.ACT_NONE, colgroups=G,
                                y=dX[:, c0:], ldy=Kp, ncols=w)
                if l > 0:
                    dz = dX
                    amax = (mx, 256) if mx is not None else None
                else:
                    dX0 = dX
            # ---- scatter the first-layer input gradient back through the gathers
            grads_src = [None] * m.n_src
            if need_dx0:
                # a source that enters as ONE dense full-width run (MAgNet's edge latents) is written, not accumulated: no zero fill,
                # no read of the destination
                runs_of = {}
                for r_ in m.runs:
                    runs_of.setdefault(r_[0], []).append(r_)
                written_once = {si for si, rs in runs_of.items()
                                if len(rs) == 1 and rs[0][2] is None and rs[0][4] == 1 and rs[0][1] == 0
                                and rs[0][6] == srcs[si].shape[-1] and not ((rs[0][6] | srcs[si].shape[-1]) & 3)
                                and srcs[si].reshape(-1, srcs[si].shape[-1]).shape[0] == M and (factored is None or not factored[rs[0][5]])}
                for si, s in enumerate(srcs):
                    if si in direct:
                        grads_src[si] = direct[si]                             # (None: a grad_chain's sum travels on to the next call)
                    elif ctx.needs_input_grad[1 + si]:
                        if factored is not None and grads_src_f[si] is not None:
                            grads_src[si] = grads_src_f[si]
                        else:
                            grads_src[si] = (torch.empty if si in written_once else torch.zeros)(s.shape, device=dev, dtype=torch.float32)
                pending = {}                 # (source, column, width, scale) -> gathered runs that can share launches (the taps of a convolution)
                for (si, col, idx, scale, rdiv, kt0, width) in m.runs:
                    g = grads_src[si]
                    if g is None or si in direct or (factored is not None and factored[kt0]):   # (done at source granularity / by the GEMM above)
                        continue
                    g2 = g.view(-1, g.shape[-1])
                    if idx is None and rdiv == 1 and (scale == 1.0 or not ((width | col | g2.stride(0)) & 3)):
                        ops.copy_cols(dX0, kt0 * 32, g2, col, width, accumulate=si not in written_once, scale=scale)   # (dense rows: a scaled column copy)
                        continue
                    if idx is None:                                              # dense rows (shared by rdiv consecutive rows)
                        rowptr = torch.arange(0, M + rdiv, rdiv, dtype=torch.int32, device=dev).clamp_(max=M)
                        perm = None
                    elif idx.data_ptr() == (m.tgt.data_ptr() if m.tgt is not None else -1):
                        rowptr, perm = m.rowptr, None                           # gathered by the CSR's own sorted target
                    else:
                        rowptr, perm = _inverse_csr(idx, g2.shape[0], m.inv_cache, drop_last=bool(getattr(pk, "last_row_is_padding", False)))
                        if width <= 256 and rdiv == 1:
                            pending.setdefault((si, col, width, scale), []).append((rowptr, perm, kt0 * 32))
                            continue
                    for (c0, w) in _tiles_of(width):
                        _segment_sum_cols(dX0, kt0 * 32 + c0, rowptr, perm, g2.shape[0], w, scale, g2, col + c0)
                for (si, col, width, scale), runs in pending.items():
                    g2 = grads_src[si].view(-1, grads_src[si].shape[-1])
                    for (c0, w) in _tiles_of(width):
                        if len(runs) == 1:
                            _segment_sum_cols(dX0, runs[0][2] + c0, runs[0][0], runs[0][1], g2.shape[0], w, scale, g2, col + c0)
                            continue
                        for i0 in range(0, len(runs), _SEG_RUNS):
                            _segment_sum_cols_runs(dX0, runs[i0:i0 + _SEG_RUNS], c0, g2.shape[0], w, scale, g2, col + c0)
            # (a grad_chain whose last call did not take the direct path above: hand the running sum over all the same)
            if m.grad_chain is not None and m.grad_chain[2] and m.grad_chain[1].get("g") is not None:
                for si, s_ in enumerate(srcs):
                    if si not in direct and s_.data_ptr() == m.grad_chain[0].data_ptr() and s_.shape == m.grad_chain[0].shape:
                        acc = m.grad_chain[1].pop("g")
                        grads_src[si] = acc if grads_src[si] is None else grads_src[si] + acc
        return (None, *grads_src, *grads_par, *([d_res] if m.has_res else []))


def _dense_source_of_block(m, srcs, ctx, M, c0, w, factored):
    """(source, scale) when the column block [c0, c0 + w) of the first Linear's input gradient IS that source's gradient - it enters the
    chain as one dense full-width run at exactly these columns, scaled by a power of two at most (folded into the transposed pack) -
    so that the data GEMM can write the gradient itself; else None"""
    hit = None
    for r_ in m.runs:
        kt0, width = r_[5], r_[6]
        if kt0 * 32 < c0 + w and c0 < kt0 * 32 + width:                       # a run that overlaps the block
            if hit is not None:
                return None
            hit = r_
    if hit is None:
        return None
    si, col, idx, scale, rdiv, kt0, width = hit
    s = srcs[si]
    pow2 = scale > 0 and float(scale) == 2.0 ** round(__import__("math").log2(scale))
    ok = (idx is None and rdiv == 1 and (scale == 1.0 or (pow2 and m.col_map is None)) and col == 0 and kt0 * 32 == c0
          and width == w == s.shape[-1] and not (w & 31)
          and ctx.needs_input_grad[1 + si] and sum(1 for q in m.runs if q[0] == si) == 1
          and s.reshape(-1, s.shape[-1]).shape[0] == M and M > 0 and (factored is None or not factored[kt0]))
    return (si, float(scale)) if ok else None


_LINEAR_BWD_PAIR = os.environ.get("MAGNET_LINEAR_BWD_PAIR", "1") != "0"     # A/B knob: 0 = every hidden Linear of a chain walked back by its own launch
_FUSED_LAYER_ROWS = 4096          # below this the persistent one-workgroup-per-CU kernel leaves most of the chip idle


def _tpack16(pk, l, W):
    """fp16x3 planes of W^T as a Linear, for ops.linear_bwd128 (cached on pk like the transposed bf16x6 packs)"""
    from .packing import flag_if_abs_reaches, recording
    cache = pk.__dict__.setdefault("_tpacks16", {})
    rec, key = _tpack_recorder(pk, W, cache, l)
    if key not in cache:
        with recording(rec):                            # (re-packed in place with the forward's packs after an optimizer step)
            flag_if_abs_reaches(W.detach(), 31.98)      # that kernel keeps 2^11 x the high plane (reported by the next range check)
            cache[key] = pack(W.detach().t(), None, "fp16x3")[0]
    return cache[key]


def _tpack_recorder(pk, W, cache, key):
    """(recorder to record under, cache key) of a transposed pack cached on the chain namespace ``pk``.  A namespace whose packs are
    replayed in place after an optimizer step (packing.PackRecorder, attached as pk._rec) outlives the step: a transposed pack of a
    PARAMETER is recorded and replayed with them; one made from a temporary (EDSR hands its convolution weights over as a permuted
    copy) cannot be replayed, so its key carries the recorder's generation - rebuilt after every replay, older generations dropped."""
    rec = pk.__dict__.get("_rec")
    if rec is None or rec.backed(W):
        return rec, key
    gen = rec.generation
    for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == "gen" and k[1] != gen]:
        del cache[k]
    return None, ("gen", gen, key)


def _tpack(m, pk, l, c0, w, W, Np, Kp, scale=1.0):
    """packed transpose of the padded weight block Wp[:, c0:c0+w] as a Linear [w outputs, Np inputs] (cached on pk; the pack of a
    step's weights lives as long as that step's ``pk``).  Wp is never materialised: the device packer reads W through its strides,
    W.t() is a view, and the placement of W's columns inside the padded first-layer input (col_map / first_cols) becomes the
    packer's row map."""
    from .packing import pack_device, recording, source_map
    cache = pk.__dict__.setdefault("_tpacks", {})
    rec, key = _tpack_recorder(pk, W, cache, (l, c0, w) if scale == 1.0 else (l, c0, w, float(scale)))
    if key not in cache:
        n, k = W.shape
        rows = None                                            # padded column c0 + r of Wp -> source column of W (identity: r < k)
        if l == 0 and m.col_map is not None:
            full = m.__dict__.get("_full_map")                # one map per mlp call, shared by its column tiles
            if full is None:
                full = m.__dict__["_full_map"] = source_map(m.col_map, Kp, W.device, source_cols=m.first_cols)
            rows = full[c0:c0 + w].contiguous()
            Wt = W.detach().t()
        else:
            Wt = W.detach().t()[c0:c0 + w] if c0 < k else W.detach().t()[:0]
        with recording(rec):
            if scale != 1.0:                                  # (a Term's power-of-two scale folded into the block: exact; refreshed on a pack replay)
                from .packing import derived
                assert rows is None
                Wt = derived(lambda: (W.detach().t()[c0:c0 + w] * float(scale)).contiguous())
            cache[key] = pack_device(Wt, None, BWD_GEMM, w, Np, row_src=rows)
    return cache[key]


# ------------------------------------------------------------------------------------------------ convolution (EDSR)
class _Conv(torch.autograd.Function):
    """One convolution of EDSR (models/backbones/edsr.py:32-60) as a gathered-tap Linear, HIP forward and backward, with everything
    the generic ``_Mlp`` works out per call known in advance: T taps of one feature map with a zero padding row, no LayerNorm, one
    Linear.  The argument blocks come from one C call each (ops.TapTiles), and
      * d(feature map) is ONE launch of the forward's own kernel on dz with the transposed weight blocks over the OPPOSITE taps
        (dsrc[p] = sum_t W_t^T dz[nbr_(T-1-t)(p)]: the taps of magnet_conv_tap_index are row-major offsets, tap T - 1 - t undoes tap t)
        instead of dz W (a [P][T x C] array), T inverse CSRs and their segment sums: 157 -> 66 us per convolution at the reference's
        magnet_cnn_2d batch size;
      * dW = dz^T x0 over the materialised taps (magnet_wgrad_tiles_f32), db = column sums of dz."""

    @staticmethod
    def forward(ctx, meta, src, W2, bias, res):
        m, pk = meta, meta.pk
        y = torch.empty(m.P, pk.out_dim, device=src.device, dtype=torch.float32)
        ops.mlp(m.P, ops.TapTiles(src, m.width, m.taps), pk.layers, nl=1, ht=pk.ht, ot=pk.ot,
                act=ops.ACT_RELU if m.relu else ops.ACT_NONE, last_act=m.relu, epi=ops.EPI_STORE_RES if res is not None else ops.EPI_STORE,
                y=y, res=res, ncols=pk.out_dim)
        ctx.meta = m
        ctx.save_for_backward(src, W2, *([y] if m.relu else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        m, pk = ctx.meta, ctx.meta.pk
        src, W2 = ctx.saved_tensors[:2]
        dev, P, T, w0 = dy.device, m.P, m.taps.shape[0], m.width
        Np, K0 = pk.ot * 32, m.taps.shape[0] * m.width
        n, k = W2.shape
        assert pk.out_dim == Np, "EDSR feature widths are multiples of 32"
        dy = dy.contiguous().float()
        with torch.no_grad():
            d_res = dy if ctx.needs_input_grad[4] else None
            if m.relu:                                                       # y = ReLU(conv): mask the incoming gradient first
                dz = torch.empty_like(dy)
                L.check(L.lib().magnet_relu_bwd_f32(ops._p(dy), ops._p(ctx.saved_tensors[2]), ops._p(dz), C.c_long(dy.numel()),
                                                    ops._stream()), "relu_bwd")
            else:
                dz = dy
            gW = gb = gsrc = None
            if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
                x0 = torch.empty(P, K0, device=dev, dtype=torch.float32)
                a, keep = ops.mlp_args(P, ops.TapTiles(src, w0, m.taps))
                L.check(L.lib().magnet_mlp_materialize_input_f32(C.byref(a), ops._p(x0), K0, ops._stream()), "materialize_input")
                dWp = torch.empty(Np, K0, device=dev, dtype=torch.float32)
                dbp = torch.empty(Np, device=dev, dtype=torch.float32)
                nblk_w = min(768, (P + 31) // 32)
                ws = None
                for j, (c0, w, G) in enumerate(_tile_groups(K0)):
                    need = G * nblk_w * Np * w + nblk_w * Np + G * Np * w
                    if ws is None or ws.numel() < need:
                        ws = torch.empty(need, device=dev, dtype=torch.float32)
                    L.check(L.lib().magnet_wgrad_tiles_f32(ops._p(dz), Np, Np, ops._p(x0), K0, c0, w, G, C.c_long(P), ops._p(dWp), K0, c0,
                                                           ops._p(dbp if j == 0 else None), ops._p(ws), 0, ops._stream()), "wgrad_tiles")
                gW = (dWp[:n][:, pk.col_map] if pk.col_map is not None else dWp[:n, :k]).contiguous()
                gb = dbp[:n].contiguous()
            if ctx.needs_input_grad[1]:
                cache = pk.__dict__.setdefault("_tpacks", {})
                rec, key = _tpack_recorder(pk, W2, cache, ("conv_t",))
                if key not in cache:
                    from .packing import recording
                    Wp = torch.zeros(Np, K0, device=dev, dtype=torch.float32)
                    if pk.col_map is not None:
                        Wp[:n, pk.col_map.long()] = W2.detach().float()
                    else:
                        Wp[:n, :k] = W2.detach().float()
                    Wt = Wp.reshape(Np, T, w0).permute(2, 1, 0).reshape(w0, T * Np).contiguous()     # [source channel][t][co]
                    with recording(None):                                    # (made from a temporary: never replayed)
                        cache[key] = pack(Wt, None, BWD_GEMM)
                dzp = torch.cat([dz, torch.zeros(1, Np, device=dev, dtype=torch.float32)])       # + the zero row of the out-of-image taps
                gsrc = torch.empty(P + 1, w0, device=dev, dtype=torch.float32)
                gsrc[P:].zero_()
                ops.mlp(P, ops.TapTiles(dzp, Np, m.taps_rev), [cache[key]], nl=1, ht=w0 // 32, ot=w0 // 32, act=ops.ACT_NONE, y=gsrc,
                        ldy=w0, ncols=w0)
        return None, gsrc, gW, gb, d_res


def conv(src, taps, taps_rev, pk, W2, bias, relu=False, res=None):
    """differentiable EDSR convolution: ``src`` [P + 1, width] channels-last rows (last row zero), ``taps`` [T, P] from
    ops.conv_tap_index and ``taps_rev`` = taps.flip(0) (the opposite taps, contiguous), ``pk`` the chain namespace of the packed weight
    (layers, ht, ot, out_dim, col_map), ``W2`` [Co, T * Ci] / ``bias`` the parameters behind it (views of the conv's own)"""
    assert not (relu and res is not None)
    meta = types.SimpleNamespace(P=int(taps.shape[1]), width=int(src.shape[1]), taps=taps, taps_rev=taps_rev, pk=pk, relu=bool(relu))
    return _Conv.apply(meta, ops._f32(src, "conv source"), W2, bias, res)


def mlp(M, groups, pk, params, *, epi=ops.EPI_STORE, res=None, csr=None, out_rows=None, first_cols=None, final_relu=False, grad_chain=None):
    """Differentiable fused MLP.  ``groups``: the arguments of ``ops.ktiles`` (Terms or (Term, Term) pairs); ``pk``: the
    packed namespace of the chain (layers, ln, ht, ot, out_dim, col_map); ``params``: [W0, b0, ..., (ln.weight, ln.bias)]
    the torch parameters behind ``pk``.  Returns a fresh [out_rows, out_dim] tensor attached to the autograd graph.
    ``grad_chain`` = (tensor, holder dict, last): ``tensor`` is a dense source of several mlp calls whose backward calls run one after the
    other (the processor's hops); each adds its share of d(tensor) to the sum kept in ``holder`` and only the call with ``last`` (the one
    whose backward runs last = the first hop) returns it - instead of one gradient per call and torch adding them up."""
    srcs, tiles, runs = [], [], []
    ids = {}

    def src_index(t):
        key = id(t)
        if key not in ids:
            ids[key] = len(srcs)
            srcs.append(t)
        return ids[key]

    kt = 0
    for g in groups:
        terms = g if isinstance(g, (tuple, list)) else (g,)
        w = terms[0].width
        for term in terms:
            runs.append((src_index(term.t), term.col, term.idx, float(term.scale), int(term.rdiv), kt, w))
        for k in range(w // 32):
            tiles.append([(src_index(term.t), term.col + 32 * k, term.idx, float(term.scale), int(term.rdiv)) for term in terms])
        kt += w // 32
    meta = types.SimpleNamespace(
        M=int(M), tiles=tiles, runs=runs, pk=pk, epi=epi, n_src=len(srcs), n_par=len(params), has_res=res is not None,
        tgt=csr.tgt if csr is not None else None, rowptr=csr.rowptr if csr is not None else None,
        out_rows=int(out_rows if out_rows is not None else M), first_cols=first_cols, col_map=getattr(pk, "col_map", None),
        inv_cache={}, final_relu=bool(final_relu), grad_chain=grad_chain, accum=_ACCUM["on"])
    assert not (final_relu and (pk.ln is not None or epi != ops.EPI_STORE))
    srcs = [ops._f32(s, "mlp source") for s in srcs]
    extra = [ops._f32(res, "residual")] if res is not None else []
    return _Mlp.apply(meta, *srcs, *params, *extra)


# ------------------------------------------------------------------------------------------------ small ops
class _NodeFeatures(torch.autograd.Function):
    """nf[b*N + n] = [u[b, n, 0:T] | x | t_last | 0-pad]  (magnet_node_features_f32); only u is differentiable"""

    @staticmethod
    def forward(ctx, u, flat, t_last, B, N, T, d, ld):
        nf = torch.empty(B * N, ld, device=u.device, dtype=torch.float32)
        uf = ops._f32(u.reshape(B * N, T))
        L.check(L.lib().magnet_node_features_f32(ops._p(uf), ops._p(flat), ops._p(ops._f32(t_last)), B, N, T, d, ld, ops._p(nf),
                                                 ops._stream()), "node_features")
        ctx.shape = (B, N, T)
        return nf

    @staticmethod
    def backward(ctx, dnf):
        B, N, T = ctx.shape
        du = torch.empty(B * N, T, device=dnf.device, dtype=torch.float32)
        ops.copy_cols(dnf.contiguous(), 0, du, 0, T)
        return du.view(B, N, T), None, None, None, None, None, None, None


def node_features(u, flat, t_last, ld):
    B, N, T = u.shape
    return _NodeFeatures.apply(u, flat, t_last, B, N, T, flat.shape[1], ld)


class _KnnBlend(torch.autograd.Function):
    """z = blend of the two nearest LR nodes (magnet_inr_knn_blend_f32, models/magnet_gnn.py:254-280); differentiable in
    ``base``, ``vals`` and the two rank-1 columns (value, time) of the ``proj_head`` weight"""

    @staticmethod
    def forward(ctx, base, vals, W, meta):
        m = meta
        wv, wt = W[:, m.col_v].float().contiguous(), W[:, m.col_t].float().contiguous()
        z = torch.empty(m.B * m.Nq * m.T, m.C, device=base.device, dtype=torch.float32)
        L.check(L.lib().magnet_inr_knn_blend_f32(ops._p(base), ops._p(m.w), ops._p(vals), ops._p(m.nn2), ops._p(m.t), m.t.shape[1],
                                                 ops._p(wv), ops._p(wt), m.B, m.L, m.Nq, m.T, m.C, m.mode, ops._p(z), ops._stream()),
                "inr_knn_blend")
        ctx.meta = m
        ctx.save_for_backward(vals, wv, W)
        return z

    @staticmethod
    def backward(ctx, dz):
        m = ctx.meta
        vals, wv, W = ctx.saved_tensors
        dev = dz.device
        nq = m.B * m.Nq
        dbase = torch.empty(2 * nq, m.C, device=dev, dtype=torch.float32)
        dvq = torch.empty(2 * nq, m.T, device=dev, dtype=torch.float32)
        nblk = min(1024, (nq + 3) // 4)
        part = torch.empty(nblk, 2, m.C, device=dev, dtype=torch.float32)
        L.check(L.lib().magnet_inr_knn_blend_bwd_f32(ops._p(dz.contiguous()), ops._p(m.w), ops._p(vals), ops._p(m.nn2), ops._p(m.t),
                                                     m.t.shape[1], ops._p(wv), m.B, m.L, m.Nq, m.T, m.C, m.mode, ops._p(dbase),
                                                     ops._p(dvq), ops._p(part), ops._stream()), "inr_knn_blend_bwd")
        g = torch.zeros(2 * m.C, device=dev, dtype=torch.float32)
        ops.reduce_partials(part, nblk, 2 * m.C, g)
        dW = torch.zeros_like(W)
        dW[:, m.col_v], dW[:, m.col_t] = g[:m.C], g[m.C:]
        # scatter the per-(query, neighbour) value gradients to the LR nodes: pad T to a launchable width
        Tp = 4
        while Tp < m.T:
            Tp *= 2
        dvp = _pad_cols(dvq, Tp)
        dvals_p = torch.zeros(vals.shape[0], Tp, device=dev, dtype=torch.float32)
        rowptr, perm = _inverse_csr(m.nn2, vals.shape[0], m.inv_cache)
        _segment_sum_cols(dvp, 0, rowptr, perm, vals.shape[0], Tp, 1.0, dvals_p, 0)
        return dbase, dvals_p[:, :m.T].contiguous(), dW, None


def knn_blend(base, vals, W, meta):
    return _KnnBlend.apply(base, vals, W, meta)


class _Euler(torch.autograd.Function):
    """out[b, k, node] = last + (t[b, T+k] - t[b, T-1]) * ret  (magnet_euler_update_f32, models/magnet_gnn.py:359-369)"""

    @staticmethod
    def forward(ctx, ret, last, t, B, n, T, T_out):
        out = torch.empty(B, T_out, n, 1, device=ret.device, dtype=torch.float32)
        tt = ops._f32(t)
        L.check(L.lib().magnet_euler_update_f32(ops._p(ret), ret.shape[1], ops._p(ops._f32(last.reshape(-1))), ops._p(tt), tt.shape[1],
                                                B, n, T, T_out, ops._p(out), ops._stream()), "euler_update")
        ctx.args = (tt, B, n, T, T_out, ret.shape[1], last.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        tt, B, n, T, T_out, ld, lshape = ctx.args
        dret = torch.zeros(B * n, ld, device=dout.device, dtype=torch.float32)
        dlast = torch.empty(B * n, device=dout.device, dtype=torch.float32)
        L.check(L.lib().magnet_euler_update_bwd_f32(ops._p(dout.contiguous()), ops._p(tt), tt.shape[1], B, n, T, T_out, ops._p(dret), ld,
                                                    ops._p(dlast), ops._stream()), "euler_update_bwd")
        return dret, dlast.view(lshape), None, None, None, None, None


def euler_update(ret, last, t, B, n, T, T_out):
    return _Euler.apply(ret, last, t, B, n, T, T_out)


class _GridBlend(torch.autograd.Function):
    """z = area blend of the 2^d shifted cell predictions (magnet_inr_grid_blend_f32, models/magnet_cnn_2d.py:285-290)"""

    @staticmethod
    def forward(ctx, p, area, n_rows, T, C, S):
        z = torch.empty(n_rows * T, C, device=p.device, dtype=torch.float32)
        L.check(L.lib().magnet_inr_grid_blend_f32(ops._p(p), ops._p(area), n_rows, T, C, S, ops._p(z), ops._stream()), "inr_grid_blend")
        ctx.args = (area, n_rows, T, C, S)
        return z

    @staticmethod
    def backward(ctx, dz):
        area, n_rows, T, C, S = ctx.args
        dp = torch.empty(n_rows * S * T, C, device=dz.device, dtype=torch.float32)
        L.check(L.lib().magnet_inr_grid_blend_bwd_f32(ops._p(dz.contiguous()), ops._p(area), n_rows, T, C, S, ops._p(dp), ops._stream()),
                "inr_grid_blend_bwd")
        return dp, None, None, None, None, None


def grid_blend(p, area, n_rows, T, C, S):
    return _GridBlend.apply(p, area, n_rows, T, C, S)
