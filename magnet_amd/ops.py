"""Thin Python wrappers over the C ABI (include/magnet_hip.h).  torch is used for device memory
and the current stream only; all arithmetic happens in libmagnet_hip.so.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib as L
from .packing import SinglePlanes
from ._lib import ACT_NONE, ACT_RELU, ACT_SWISH, EPI_QUADBLEND, EPI_SEGMEAN, EPI_STORE, EPI_STORE_MASK, EPI_STORE_RES  # noqa: F401


_raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream() -> C.c_void_p:
    """the current stream of the current device as the raw hipStream_t (torch.cuda.current_stream() builds a Stream object and costs
    8 us a call - 16 ms of a MAgNet training step at the reference's batch sizes, which is host bound; the C accessor is 20 x cheaper)"""
    return C.c_void_p(_raw_stream(_cur_dev()))


def _same_device(t: torch.Tensor, name: str) -> None:
    """kernels are enqueued on the current stream of the CURRENT device (one process per GPU, torch.cuda.set_device(local_rank)
    first - what Lightning's ddp strategy does for the reference): a tensor of another device would be dereferenced from the
    wrong GPU's stream, so refuse it loudly"""
    if t.device.index != _cur_dev():
        raise RuntimeError(f"magnet_amd: {name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(...) (or wrap the call in torch.cuda.device(...)) first")


def _f32(t: torch.Tensor, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"magnet_amd: {name} must live on a HIP device (got {t.device}); there is no CPU path")
    _same_device(t, name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _i32(t: torch.Tensor, name="index") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"magnet_amd: {name} must live on a HIP device")
    _same_device(t, name)
    assert t.dtype == torch.int32 and t.is_contiguous()
    return t


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ------------------------------------------------------------------------------------ MLP chain
@dataclass
class Term:
    """``width`` (multiple of 32) input columns taken from t[(idx[row] | row // rdiv), col : col + width]."""
    t: torch.Tensor
    width: int
    col: int = 0
    idx: torch.Tensor | None = None
    scale: float = 1.0
    rdiv: int = 1


def ktiles(*groups):
    """groups: Term or (Term, Term) (summed, equal widths) -> flat list of per-32-column tile specs."""
    out = []
    for g in groups:
        terms = g if isinstance(g, (tuple, list)) else (g,)
        w = terms[0].width
        assert w % 32 == 0 and all(t.width == w for t in terms) and len(terms) <= 2
        for k in range(w // 32):
            out.append([(t, t.col + 32 * k) for t in terms])
    return out


@dataclass
class TapTiles:
    """the k tiles of a convolution written as a gathered-tap Linear: tap t is src[taps[t][row], 0 : width) - what
    ``ktiles(*[Term(src, width, idx=taps[t]) for t in range(T)])`` describes, kept as ONE object so that the argument block is filled by
    one C call (magnet_mlp_args_set_taps) instead of T * width / 32 descriptors field by field from Python"""
    src: torch.Tensor           # [rows, ld] fp32, ld >= width
    width: int
    taps: torch.Tensor          # [T, M] int32, contiguous

    def __len__(self):
        return self.taps.shape[0] * (self.width // 32)


def mlp_args(M, kts):
    """argument block with the row count and the first-layer k-tile descriptors filled in -> (args, keep-alive list)"""
    a = L.MlpArgs()
    if isinstance(kts, TapTiles):
        src, taps = _f32(kts.src, "mlp source"), kts.taps
        assert taps.dtype == torch.int32 and taps.is_contiguous() and taps.shape[1] == M and src.dim() == 2
        a.M = int(M)
        L.check(L.lib().magnet_mlp_args_set_taps(C.byref(a), _p(src), src.stride(0), int(kts.width), _p(taps), C.c_long(taps.shape[1]),
                                                 int(taps.shape[0])), "mlp_args_set_taps")
        return a, [src, taps]
    a.M, a.nkt = int(M), len(kts)
    assert 1 <= len(kts) <= L.MAX_KT
    keep = []
    for i, tile in enumerate(kts):
        a.kt[i].n = len(tile)
        for j, (term, col) in enumerate(tile):
            t = _f32(term.t, "mlp source")
            keep.append((t, term.idx))
            ld = t.stride(0) if t.dim() == 2 else t.shape[-1]
            assert ld % 4 == 0 and col % 4 == 0 and col + 32 <= ld, (ld, col)
            s = a.kt[i].s[j]
            s.p, s.ld, s.col, s.scale, s.rdiv = t.data_ptr(), ld, col, float(term.scale), int(term.rdiv)
            s.idx = term.idx.data_ptr() if term.idx is not None else None
    return a, keep


def mlp_plan(M, kts, layers, *, nl, ht, ot, act, last_act=False, ln=None, epi=EPI_STORE, y=None, ldy=None, ncols=None,
             res=None, tgt=None, rowptr=None, partials=None, colgroups=1, groups_in_block=False, y_group_stride=0, pre=None,
             x0_out=None, hid=None, x0_skip=0):
    """Build the argument block of one fused-MLP launch once and return a zero-argument callable that enqueues it on the
    current stream.  For call sites that repeat the same launch on the same buffers (the EDSR convolution stack: 34
    launches of up to 36 gathered k tiles each) this removes the per-call descriptor marshalling from the host path."""
    # colgroups G > 1 (single Linear only): ``layers`` holds G packed column slices of ot*32 outputs each; the launch
    # gets G workgroup columns - more parallelism when M alone cannot fill the chip (a 64 x 64 EDSR map is 4096 rows)
    assert len(layers) == nl * colgroups and (colgroups == 1 or nl == 1)
    a, keep = mlp_args(M, kts)
    a.ncolgroups = int(colgroups)
    # groups_in_block: every workgroup runs its rows through the G groups one after the other (inputs read from HBM once)
    # instead of G workgroup columns; y_group_stride: floats between the groups' outputs (0: neighbouring column blocks)
    a.groups_in_block = int(colgroups) if groups_in_block and colgroups > 1 else 0
    a.y_group_stride = int(y_group_stride)
    keep += [layers, ln, y, res, tgt, rowptr, partials, pre, x0_out, hid]
    # tape outputs for a training backward (magnet_hip.h): assembled first-layer input, every layer's activated output, the last
    # Linear's raw output - one recomputing launch instead of a materialisation pass plus one launch per layer
    if x0_out is not None:
        assert x0_out.is_contiguous() and x0_out.shape[1] == len(kts) * 32
        a.x0_out = x0_out.data_ptr()
        a.x0_skip = int(x0_skip)            # k tiles of x0 the caller will not read (bit kt)
    for i, t in enumerate(hid or ()):
        if t is not None:
            assert nl > 1 and t.is_contiguous() and t.shape[1] == (ot if i == nl - 1 else ht) * 32
            a.hid[i] = t.data_ptr()
    # pre: up to two (rows [n, ld >= first-layer width] fp32, idx int32 or None): gathered rows added to the first Linear's
    # pre-activation (the per-node products of a factored first layer, see magnet_hip.h)
    for i, (t, idx) in enumerate(pre or ()):
        assert nl > 1 and i < 2 and t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] % 4 == 0 and t.shape[-1] >= ht * 32
        a.pre[i].p, a.pre[i].ld = t.data_ptr(), t.shape[-1]
        a.pre[i].idx = idx.data_ptr() if idx is not None else None
    for i, (w, b) in enumerate(layers):
        a.layer[i].w, a.layer[i].b = w.data_ptr(), b.data_ptr()
    if ln is not None:
        a.ln_g, a.ln_b, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    if y is None:                   # tape-only launch (fp16x3 / bf16x6 kernels, plain store epilogue): nothing but x0_out / hid[] is written
        assert epi == EPI_STORE and (x0_out is not None or hid) and layers[0][0].dtype in (torch.float16, torch.int16)
        a.y, a.ldy, a.ncols = None, int(ot * 32), int(ncols if ncols is not None else ot * 32)
    else:
        a.y, a.ldy, a.ncols = y.data_ptr(), int(ldy if ldy is not None else y.shape[-1]), int(ncols if ncols is not None else ot * 32)
    if res is not None:
        a.res, a.ldres = res.data_ptr(), (res.stride(0) if res.dim() == 2 else res.shape[-1])
    g = RangeGuard.current
    if g is not None and g.active and layers[0][0].dtype == torch.float16:                  # see RangeGuard.armed
        a.status, a.status_bit = g._word(layers[0][0].device).data_ptr(), ST_CHAIN_RANGE
        keep.append(g._status)
    if epi == EPI_SEGMEAN:
        a.tgt, a.rowptr, a.partials = tgt.data_ptr(), rowptr.data_ptr(), partials.data_ptr()
    elif epi == EPI_STORE_MASK and partials is not None:       # a zeroed float[256] that receives the block maxima of |y| (split-GEMM kernels)
        assert partials.numel() >= 256 and partials.dtype == torch.float32 and layers[0][0].dtype in (torch.float16, torch.int16)
        a.partials = partials.data_ptr()
    kind = layers[0][0].dtype                                 # float16: fp16x3, int16: bf16x6, float32: fp32 MFMA
    assert all(w.dtype == kind for w, _ in layers)
    name = {torch.float16: "magnet_mlp_fp16x3_f32", torch.int16: "magnet_mlp_bf16x6_f32", torch.float32: "magnet_mlp_f32",
            torch.bfloat16: "magnet_mlp_bf16_f32"}[kind]
    single = [isinstance(w, SinglePlanes) for w, _ in layers]
    if any(single):                                            # packing mode "fp16x3s": the one-accumulator chain kernel
        assert all(single) and colgroups == 1
        name = "magnet_mlp_fp16x3s_f32"
    fn = getattr(L.lib(), name)
    ref = C.byref(a)
    what = f"{name}(nl={nl},ht={ht},ot={ot},act={act},ln={ln is not None},epi={epi})"
    scalars = (nl, ht, ot, act, int(last_act), int(ln is not None), epi)

    def launch(_keep=keep, _a=a):
        L.check(fn(ref, *scalars, _stream()), what)
    return launch


def mlp(M, kts, layers, *, nl, ht, ot, act, last_act=False, ln=None, epi=EPI_STORE, y=None, ldy=None, ncols=None,
        res=None, tgt=None, rowptr=None, partials=None, colgroups=1, groups_in_block=False, y_group_stride=0, pre=None,
        x0_out=None, hid=None, x0_skip=0):
    """Run the fused MLP kernel.  ``layers``: list of (packed_w, padded_b).  Returns y."""
    if isinstance(layers[0][0], SinglePlanes) and not L.query("magnet_mlp_fp16x3s_supported",
                                                              nl, ht, ot, act, int(last_act), int(ln is not None), epi):
        raise NotImplementedError(f"no one-accumulator chain kernel for (nl={nl}, ht={ht}, ot={ot}, ln={ln is not None}, epi={epi}): "
                                  "set MAGNET_CHAIN_GEMM=fp16x3")
    if nl > 1 and colgroups == 1 and not L.query("magnet_mlp_supported", nl, ht, ot, act, int(last_act), int(ln is not None),
                                                  EPI_STORE if epi == EPI_QUADBLEND else epi):
        return _mlp_by_layers(M, kts, layers, nl=nl, ht=ht, ot=ot, act=act, last_act=last_act, ln=ln, epi=epi, y=y, ldy=ldy,
                              ncols=ncols, res=res, tgt=tgt, rowptr=rowptr, partials=partials, pre=pre, x0_out=x0_out, hid=hid)
    mlp_plan(M, kts, layers, nl=nl, ht=ht, ot=ot, act=act, last_act=last_act, ln=ln, epi=epi, y=y, ldy=ldy, ncols=ncols,
             res=res, tgt=tgt, rowptr=rowptr, partials=partials, colgroups=colgroups, groups_in_block=groups_in_block,
             y_group_stride=y_group_stride, pre=pre, x0_out=x0_out, hid=hid, x0_skip=x0_skip)()
    return y


def _mlp_by_layers(M, kts, layers, *, nl, ht, ot, act, last_act, ln, epi, y, ldy, ncols, res, tgt, rowptr, partials, pre, x0_out, hid):
    """A chain depth the fused kernel is not instantiated for (the reference's MLP is generic in ``mlp_layers``,
    models/backbones/mlp.py:9-28; every shipped config uses 4, i.e. the fused 5-Linear kernel): the hidden layers as one-Linear
    launches that materialise [M, hidden], the last Linear with the chain's LayerNorm / epilogue.  Correct, not fast."""
    if pre is not None:
        raise NotImplementedError("a factored first layer needs the fused chain kernel (5 Linears)")
    dev = y.device
    if x0_out is not None:                                     # tape output (training recompute): the assembled first-layer input
        a, keep = mlp_args(M, kts)
        L.check(L.lib().magnet_mlp_materialize_input_f32(C.byref(a), _p(x0_out), x0_out.shape[1], _stream()), "materialize_input")
    hid = list(hid) if hid is not None else [None] * nl
    h, src = None, kts
    for l in range(nl - 1):
        h = hid[l] if hid[l] is not None else torch.empty(max(M, 1), ht * 32, device=dev, dtype=torch.float32)
        mlp_plan(M, src, [layers[l]], nl=1, ht=ht, ot=ht, act=act, last_act=True, y=h)()
        src = ktiles(Term(h, ht * 32))
    if hid[nl - 1] is not None:                                # tape output: the last Linear's raw result (before LayerNorm)
        mlp_plan(M, src, [layers[nl - 1]], nl=1, ht=ot, ot=ot, act=ACT_NONE, last_act=False, y=hid[nl - 1])()
    mlp_plan(M, src, [layers[nl - 1]], nl=1, ht=ot, ot=ot, act=act if last_act else ACT_NONE, last_act=last_act, ln=ln, epi=epi, y=y,
             ldy=ldy, ncols=ncols, res=res, tgt=tgt, rowptr=rowptr, partials=partials)()
    return y


def segmean_fixup(rowptr, n_rows, d, partials, agg):
    L.check(L.lib().magnet_segmean_fixup_f32(_p(rowptr), n_rows, d, _p(partials), _p(agg), _stream()), "segmean_fixup")


def segment_mean(msg: torch.Tensor, rowptr: torch.Tensor, n_rows: int, out: torch.Tensor | None = None):
    msg = _f32(msg, "msg")
    h = msg.shape[1]
    out = torch.empty(n_rows, h, device=msg.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().magnet_segment_mean_f32(_p(msg), _p(_i32(rowptr)), n_rows, h, _p(out), _stream()), "segment_mean")
    return out


# ------------------------------------------------------------------------------------ fp16x3 range guard
FP16_LIMIT = 65504.0
ST_IN_RANGE, ST_IN_NONFINITE, ST_OUT_NONFINITE, ST_CHAIN_RANGE = 1, 2, 4, 8


class RangeGuard:
    """Sticky device-side status word for the fp16x3 GEMM mode (two fp16 pieces per value: magnitudes >= 65504 overflow to
    inf silently).  ``inputs(x)`` flags out-of-range / non-finite entries of an array that enters a GEMM chain,
    ``outputs(y)`` non-finite entries of a result; ``tripped()`` reads the word once (one host sync - the models call it once per
    rollout) and says whether the rollout has to be repeated in the full-range ``bf16x6`` split (the models then call
    packing.widen() and run again: the reference computes in fp32 over the whole fp32 range, configs/trainer/default.yaml:5, so data
    of that magnitude must get the reference's result, not an error).  Inactive (no launches, no sync) in the other GEMM modes and
    once the process has been widened."""

    current = None          # the guard of the rollout in progress (armed()): the chain kernels launched meanwhile report into its word

    def __init__(self, active: bool):
        self.enabled, self._status = bool(active), None

    def armed(self):
        """``with guard.armed():`` every multi-Linear fp16x3 chain launched inside also reports an activation BETWEEN two of its Linears
        that reaches 65504 (magnet_mlp_args_t.status): there the high fp16 piece saturates, the next layer sees +-inf, and a ReLU turns
        -inf into a finite - wrong - output that neither the input nor the output check sees (found on a diverging magnet_cnn_2d run
        whose masked overflow surfaced only as NaN gradients one step later)"""
        return _Armed(self)

    @property
    def active(self):
        from .packing import wide
        return self.enabled and not wide()

    def _word(self, dev):
        if self._status is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        return self._status

    def inputs(self, x):
        if self.active:
            L.check(L.lib().magnet_status_check_f32(_p(x), C.c_long(x.numel()), C.c_float(FP16_LIMIT), ST_IN_RANGE,
                                                    ST_IN_NONFINITE, ST_OUT_NONFINITE, _p(self._word(x.device)), _stream()),
                    "status_check")

    def outputs(self, y):
        if self.active:
            L.check(L.lib().magnet_status_check_f32(_p(y), C.c_long(y.numel()), C.c_float(float("inf")), ST_OUT_NONFINITE,
                                                    ST_OUT_NONFINITE, 0, _p(self._word(y.device)), _stream()), "status_check")

    def tripped(self) -> str | None:
        """None: nothing flagged (or inactive).  Otherwise the reason the fp16x3 result cannot be trusted: an input of a GEMM chain
        had magnitude >= 65504, or finite inputs gave a non-finite prediction (an intermediate left the fp16 range - or the data
        contain a genuine 0/0, e.g. duplicate LR nodes in the INR blend, which the reference turns into NaN as well: the repeat in
        bf16x6 then reproduces that NaN).  Non-finite INPUTS are not a reason: the reference propagates them too."""
        if not self.active or self._status is None:
            return None
        st = int(self._status.item())
        if st == 0:
            return None
        self._status.zero_()
        if st & ST_IN_RANGE:
            return "an input of the fp16x3 GEMM chain has magnitude >= 65504"
        if st & ST_CHAIN_RANGE:
            return "an activation inside an fp16x3 MLP chain (or handed from one fp16x3 launch to the next) reached 65504"
        if st & ST_OUT_NONFINITE and not st & ST_IN_NONFINITE:
            return "non-finite prediction from finite inputs in fp16x3 mode (an intermediate left the fp16 range)"
        return None

    def clear(self):
        if self._status is not None:
            self._status.zero_()


class _Armed:
    def __init__(self, guard):
        self.g = guard

    def __enter__(self):
        self.prev, RangeGuard.current = RangeGuard.current, self.g

    def __exit__(self, *exc):
        RangeGuard.current = self.prev
        return False


# ------------------------------------------------------------------------------------ MPNN pieces
def mpnn_node_input(u, pos, Lvec, tmax, tw, dpos, kin):
    u, pos = _f32(u, "u"), _f32(pos, "pos")
    n = u.shape[0]
    out = torch.empty(n, kin, device=u.device, dtype=torch.float32)
    L.check(L.lib().magnet_mpnn_node_input_f32(_p(u), _p(pos), pos.shape[1], _p(_f32(Lvec)), _p(_f32(tmax)), n, tw,
                                               dpos, kin, _p(out), _stream()), "mpnn_node_input")
    return out


EDGE_ENTRY = {torch.float16: "magnet_mpnn_edge_fp16x3_f32", torch.int16: "magnet_mpnn_edge_bf16x6_f32",
              torch.float32: "magnet_mpnn_edge_f32", torch.bfloat16: "magnet_mpnn_edge_bf16_f32"}


EDGE_ENTRY_L2E = {torch.float16: "magnet_mpnn_edge_fp16x3_l2e_f32", torch.bfloat16: "magnet_mpnn_edge_bf16_l2e_f32"}
# fp16x3 entry points with their own weight packs (ops.mpnn_edge16 ...), and the kernel behind every entry (bench.py names it)
EDGE_ENTRY_EXTRA = ("magnet_mpnn_edge_fp16x3_t16_f32", "magnet_mpnn_edge_stream_fp16x3_f32")      # (+ the development library's: dev_ops.py)
EDGE_KERNEL_OF_ENTRY = {"magnet_mpnn_edge_stream_fp16x3_f32": "k_mpnn_edge_stream32", "magnet_mpnn_edge_rowsp_fp16x3_f32": "k_mpnn_edge_rowsp", "magnet_mpnn_edge_rowsg_fp16x3_f32": "k_mpnn_edge_rowsg", "magnet_mpnn_edge_rowsr_fp16x3_f32": "k_mpnn_edge_rowsr", "magnet_mpnn_edge_rows_fp16x3_f32": "k_mpnn_edge_rows32", "magnet_mpnn_edge_fp16x3_t16_f32": "k_mpnn_edge16", "magnet_mpnn_edge_fp16x3_t32_f32": "k_mpnn_edge32t", "magnet_mpnn_edge_fp16x3_ws_f32": "k_mpnn_edge_ws", "magnet_mpnn_edge_fp16x3_f32": "k_mpnn_edge7",
                        "magnet_mpnn_edge_fp16x3_l2e_f32": "k_mpnn_edge7", "magnet_mpnn_edge_bf16_f32": "k_mpnn_edge7",
                        "magnet_mpnn_edge_bf16_l2e_f32": "k_mpnn_edge7", "magnet_mpnn_edge_bf16x6_f32": "k_mpnn_edge6",
                        "magnet_mpnn_edge_f32": "k_mpnn_edge2"}
LOG2E = 1.4426950408889634


def mpnn_edge(A, Bm, csr, w2, b2, agg, partials, l2e=False):
    """fused MPNN edge kernel; the dtype of the packed message_net_2 weights selects the arithmetic
    (float16 planes: fp16x3 split, int16 planes: bf16x6 split, float32 fragments: fp32 MFMA).
    l2e: A and Bm carry the factor log2(e) (EDGE_ENTRY_L2E modes only; see magnet_hip.h)."""
    name = EDGE_ENTRY_L2E[w2.dtype] if l2e else EDGE_ENTRY[w2.dtype]
    L.check(getattr(L.lib(), name)(_p(A), _p(Bm), _p(csr.src), _p(csr.tgt), _p(csr.rowptr), csr.E, csr.N, _p(w2), _p(b2),
                                   _p(agg), _p(partials), _stream()), name)
    return agg


def edge_scratch_floats(n_edges: int) -> int:
    """size of the ``partials`` scratch of the fused edge kernels (the 16-edge-tile kernel needs two 128-float rows per
    16 edges at most; the 32-edge ones half of that)"""
    return max((int(n_edges) + 15) // 16, 1) * 2 * 128


def mpnn_edge16(A, Bm, csr, w2_t16, b2, agg, partials, l2e=False, single=False):
    """the same op on 16-edge tiles with two waves per SIMD (k_mpnn_edge16, fp16x3 only); w2_t16 from
    packing.pack_edge16_fp16x2 - the fragment order differs from the 32-edge kernels' packs.  single: the planes were packed
    with scaled=True (both carry 2^11): one accumulator for the three products of the split, pieces straight out of the Swish"""
    assert w2_t16.dtype == torch.float16 and partials.numel() >= edge_scratch_floats(csr.E)
    L.check(L.lib().magnet_mpnn_edge_fp16x3_t16_f32(_p(A), _p(Bm), _p(csr.src), _p(csr.tgt), _p(csr.rowptr), csr.E, csr.N,
                                                    _p(w2_t16), _p(b2), _p(agg), _p(partials), int(l2e) | (2 if single else 0),
                                                    _stream()),
            "mpnn_edge_fp16x3_t16")
    return agg


STREAM_DEEP = int(__import__("os").environ.get("MAGNET_STREAM_DEEP", "1"))      # gather sets in flight beyond two (0, 1; development library: 2 and + 4)


@dataclass
class StreamTiles:
    """Tile plan of k_mpnn_edge_stream32 (generation 14): the rows cut into n_chunks row-aligned ranges of about equal edge count, every
    chunk's edge stream cut into 32-edge tiles that touch at most two rows."""
    tiles: torch.Tensor            # [n_chunks, tpc, 4] int32 descriptors (include/magnet_hip.h)
    ntiles: torch.Tensor           # [n_chunks] int32
    n_chunks: int
    tpc: int
    zero_fill: bool                # some rows have no edge: the kernel does not write them
    n_tiles: int = 0               # tiles of all chunks (fill = E / (32 n_tiles))


def stream_tiles(csr: "CSR"):
    """build the plan on the device (one launch + one host read of its flags); None when the graph does not fit the kernel (a row with
    65 536 edges or more, a chunk that would need more tiles than planned for: very uneven rows)"""
    n, dev = csr.N, csr.rowptr.device
    if n <= 0 or csr.E <= 0:
        return None
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    n_chunks = 8 * max(1, min((csr.E + 255) // 256, ncu))
    tpc = (csr.E + 32 * n_chunks - 1) // (32 * n_chunks) + 8          # a chunk overshoots its share by less than one row (+ cuts)
    tiles = torch.empty(n_chunks, tpc, 4, dtype=torch.int32, device=dev)
    ntiles = torch.empty(n_chunks, dtype=torch.int32, device=dev)
    flags = torch.empty(2, dtype=torch.int32, device=dev)
    L.check(L.lib().magnet_stream_tiles_plan(_p(csr.rowptr), n, n_chunks, tpc, _p(tiles), _p(ntiles), _p(flags), _stream()), "stream_tiles_plan")
    fl, total = (int(v) for v in flags.tolist())                       # one host sync per plan
    if fl & 5:
        return None
    return StreamTiles(tiles, ntiles, n_chunks, tpc, bool(fl & 2), total)


STREAM_L2E, STREAM_GATHER3, STREAM_DEV_GATHER4, STREAM_DEV_PRIO = 1, 2, 4, 8        # include/magnet_hip.h / magnet_hip_dev.h


def mpnn_edge_stream(A, Bm, csr, plan: StreamTiles, w2_t32, b2, agg, l2e=False, deep=None):
    """the fused edge op by k_mpnn_edge_stream32 (generation 14): ONE launch, no remainder pass, no straggler lists.
    ``deep``: gather sets in flight beyond two (0, 1 = default; MAGNET_STREAM_DEEP, A/B knob - bitwise the same result; the development
    library also takes 2 = four sets and + 4 = s_setprio 1 for the younger wave of every SIMD)"""
    if deep is None:
        deep = STREAM_DEEP
    assert w2_t32.dtype == torch.float16
    if plan.zero_fill:
        agg.zero_()
    flags = (STREAM_L2E if l2e else 0) | {0: 0, 1: STREAM_GATHER3, 2: STREAM_DEV_GATHER4}[int(deep) & 3] | (STREAM_DEV_PRIO if int(deep) & 4 else 0)
    L.check(L.lib().magnet_mpnn_edge_stream_fp16x3_f32(_p(A), _p(Bm), _p(csr.src), _p(plan.tiles), _p(plan.ntiles), plan.n_chunks, plan.tpc,
                                                       csr.N, _p(w2_t32), _p(b2), _p(agg), flags, _stream()), "mpnn_edge_stream_fp16x3")
    return agg


mpnn_edge_bf16x6 = mpnn_edge     # kept for scripts: the weights' dtype decides


def mpnn_readout(h, u, tw, w1, b1, k1, s1, w2, b2, k2, use_swish, dt, stats=None, rows_per_graph=0, eps=1e-5):
    """``stats`` = (mean, var): h is the raw output of the last hop and is normalised while it is loaded (returns None when no fused
    kernel covers the shape: normalise first, then call without ``stats``)"""
    n = h.shape[0]
    out = torch.empty(n, tw, device=h.device, dtype=torch.float32)
    if stats is not None:
        rc = L.lib().magnet_mpnn_readout_norm_f32(_p(h), _p(stats[0]), _p(stats[1]), C.c_float(float(eps)), stats[0].shape[0],
                                                  int(rows_per_graph), _p(u), n, tw, _p(w1), _p(b1), k1, s1, _p(w2), _p(b2), k2,
                                                  int(use_swish), _p(_f32(dt)), _p(out), _stream())
        if rc == -2:
            return None
        L.check(rc, "mpnn_readout_norm")
        return out
    L.check(L.lib().magnet_mpnn_readout_f32(_p(h), _p(u), n, tw, _p(w1), _p(b1), k1, s1, _p(w2), _p(b2), k2,
                                            int(use_swish), _p(_f32(dt)), _p(out), _stream()), "mpnn_readout")
    return out


def instance_norm(x, n_graphs, rows_per_graph, eps=1e-5, out=None, return_var=False):
    x = _f32(x, "x")
    h = x.shape[1]
    nchunks = (rows_per_graph + 63) // 64
    ws = torch.empty((2 * n_graphs * nchunks + 2 * n_graphs) * h, device=x.device, dtype=torch.float32)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().magnet_instance_norm_f32(_p(x), n_graphs, rows_per_graph, h, C.c_float(float(eps)), _p(out), _p(ws),
                                             _stream()), "instance_norm")
    if return_var:                                            # biased variance [n_graphs, h] (kept for the backward)
        o = (n_graphs * nchunks + n_graphs) * h
        return out, ws[o:o + n_graphs * h].view(n_graphs, h)
    return out


def instance_norm_ragged(x, gptr, eps=1e-5, out=None, return_var=False):
    """per-graph InstanceNorm for graphs of unequal size: graph g owns rows gptr[g] .. gptr[g + 1] (int32, device)"""
    x = _f32(x, "x")
    ng, h = gptr.numel() - 1, x.shape[1]
    out = torch.empty_like(x) if out is None else out
    stats = torch.empty(2, max(ng, 1), h, device=x.device, dtype=torch.float32)
    L.check(L.lib().magnet_instance_norm_ragged_f32(_p(x), _p(_i32(gptr)), ng, h, C.c_float(float(eps)), _p(out), _p(stats[0]),
                                                    _p(stats[1]), _stream()), "instance_norm_ragged")
    return (out, stats[1]) if return_var else out           # (biased variance [n_graphs, h]: kept for the backward)


def instance_norm_ragged_bwd(dy, yhat, var, gptr, eps=1e-5, out=None):
    out = torch.empty_like(dy) if out is None else out
    L.check(L.lib().magnet_instance_norm_ragged_bwd_f32(_p(dy), _p(yhat), _p(var), _p(_i32(gptr)), gptr.numel() - 1, dy.shape[1],
                                                        C.c_float(float(eps)), _p(out), _stream()), "instance_norm_ragged_bwd")
    return out


NORM_SMALL_ROWS = 2048          # csrc/mpnn_kernels.hip: graphs up to this size are normalised by ONE launch


def instance_norm_stats(x, n_graphs, rows_per_graph):
    """-> (mean, var) [n_graphs, H] views of a fresh workspace: the statistics of ``instance_norm`` without the apply pass"""
    x = _f32(x, "x")
    h = x.shape[1]
    nchunks = (rows_per_graph + 63) // 64
    ws = torch.empty((2 * n_graphs * nchunks + 2 * n_graphs) * h, device=x.device, dtype=torch.float32)
    L.check(L.lib().magnet_instance_norm_stats_f32(_p(x), n_graphs, rows_per_graph, h, _p(ws), _stream()), "instance_norm_stats")
    o = n_graphs * nchunks * h
    return ws[o:o + n_graphs * h].view(n_graphs, h), ws[o + n_graphs * h:o + 2 * n_graphs * h].view(n_graphs, h)


def node_halves_supported(kin, n_graphs, rows_per_graph, h=128):
    return h == 128 and kin in (32, 64) and 1 <= n_graphs <= 8 and rows_per_graph % 32 == 0


def mpnn_node_halves(h, node_in, pre_a, pre_b, A, Bm, stats=None, rows_per_graph=0, eps=1e-5, h_norm=None):
    """A = Wa [hn | node_in] + ba, Bm = Wb [hn | node_in] (+ bb) with hn = InstanceNorm-applied h when ``stats`` = (mean, var)
    is given (then hn is also written to ``h_norm``).  pre_a / pre_b: (fp16x3 planes, padded bias) of ``packing.pack``."""
    n, kin = h.shape[0], (node_in.shape[1] if node_in is not None else 0)
    mean, var = stats if stats is not None else (None, None)
    ng = mean.shape[0] if mean is not None else 0
    # one-accumulator packs (packing mode "fp16x3s"): k_node_halves2, two waves per SIMD; else the two-accumulator k_node_halves
    single = isinstance(pre_a[0], SinglePlanes) and isinstance(pre_b[0], SinglePlanes)
    entry = L.lib().magnet_mpnn_node_halves2_f32 if single else L.lib().magnet_mpnn_node_halves_f32
    L.check(entry(_p(h), _p(node_in), kin, C.c_long(n), _p(mean), _p(var), C.c_float(float(eps)), ng,
                  int(rows_per_graph), _p(pre_a[0]), _p(pre_a[1]), _p(pre_b[0]), _p(pre_b[1]),
                  _p(A), _p(Bm), _p(h_norm), _stream()), "mpnn_node_halves")


def mpnn_node_update(h, agg, node_in, upd1, upd2, y, stats=None, rows_per_graph=0, eps=1e-5, stats_out_graphs=0):
    """y = hn + Swish(W2 Swish(W1 [hn | agg | node_in] + b1) + b2), hn = InstanceNorm-applied h when ``stats`` = (mean, var);
    upd1 / upd2: (fp16x3 planes, padded bias).  ``stats_out_graphs`` = n_graphs > 0: also returns (mean, var) of y per graph of
    ``rows_per_graph`` rows (multiple of 128), computed from partials the kernel emits - no further pass over y."""
    assert y.data_ptr() != h.data_ptr()
    n = h.shape[0]
    mean, var = stats if stats is not None else (None, None)
    ng = mean.shape[0] if mean is not None else 0
    ps = pm = None
    if stats_out_graphs:
        assert rows_per_graph % 128 == 0 and n == stats_out_graphs * rows_per_graph
        part = torch.empty(2, n // 128, 128, device=h.device, dtype=torch.float32)
        ps, pm = part[0], part[1]
    # one-accumulator packs (packing mode "fp16x3s"): k_node_update2, two workgroups per CU; else the two-accumulator k_node_update
    single = isinstance(upd1[0], SinglePlanes) and isinstance(upd2[0], SinglePlanes)
    entry = L.lib().magnet_mpnn_node_update2_f32 if single else L.lib().magnet_mpnn_node_update_f32
    L.check(entry(_p(h), _p(agg), _p(node_in), node_in.shape[1], C.c_long(n), _p(mean), _p(var),
                  C.c_float(float(eps)), ng, int(rows_per_graph), _p(upd1[0]), _p(upd1[1]),
                  _p(upd2[0]), _p(upd2[1]), _p(y), _p(ps), _p(pm), _stream()), "mpnn_node_update")
    if not stats_out_graphs:
        return y
    mv = torch.empty(2, stats_out_graphs, 128, device=h.device, dtype=torch.float32)
    L.check(L.lib().magnet_colstat_final_f32(_p(ps), _p(pm), stats_out_graphs, int(rows_per_graph), 128, 128, _p(mv[0]), _p(mv[1]),
                                             _stream()), "colstat_final")
    return mv[0], mv[1]


def mpnn_node_update_norm_small_supported(kin: int, n_rows: int) -> bool:
    """csrc/mlp_bf16_kernels.hip: update + InstanceNorm in one launch when the whole batch fits one workgroup"""
    return kin == 32 and 1 <= n_rows <= 128


def mpnn_node_update_norm_small(h, agg, node_in, upd1, upd2, y_norm, rows_per_graph, eps=1e-5):
    """y_norm = InstanceNorm(h + update_net(h, agg, node_in)) over blocks of ``rows_per_graph`` rows, one launch (n <= 128 rows)"""
    assert y_norm.data_ptr() != h.data_ptr()
    L.check(L.lib().magnet_mpnn_node_update_norm_small_f32(_p(h), _p(agg), _p(node_in), node_in.shape[1], C.c_long(h.shape[0]),
                                                           int(rows_per_graph), C.c_float(float(eps)), _p(upd1[0]), _p(upd1[1]),
                                                           _p(upd2[0]), _p(upd2[1]), _p(y_norm), _stream()),
            "mpnn_node_update_norm_small")
    return y_norm


def conv_resident_plan(fmap, width, taps, packs, relu, res, y, ldy):
    """prebuilt launch of magnet_conv_resident_f32 (EDSR 3 x 3 convolution on a small map, weights resident in LDS) or None when the
    shape is outside that kernel; ``packs``: per 32-column group (fp16x3 planes, bias)"""
    ncg, ntaps, P = len(packs), taps.shape[0], taps.shape[1]
    if width not in (32, 128) or ntaps not in (3, 9) or not 1 <= ncg <= 4 or packs[0][0].dtype != torch.float16:
        return None
    wp = (C.c_void_p * ncg)(*[w.data_ptr() for w, _ in packs])
    bp = (C.c_void_p * ncg)(*[b.data_ptr() for _, b in packs])
    g = RangeGuard.current
    word = g._word(fmap.device) if g is not None and g.active else None                    # see RangeGuard.armed
    keep = (fmap, taps, packs, res, y, word)
    fn = L.lib().magnet_conv_resident_f32
    args = (_p(fmap), int(width), _p(taps), int(ntaps), int(P), wp, bp, ncg, int(bool(relu)), _p(res), _p(y), int(ldy), _p(word),
            ST_CHAIN_RANGE)

    def go(_keep=keep):
        L.check(fn(*args, _stream()), "conv_resident")
    return go


def resize_linear(x, out_hw, d):
    """x [n, Hi, Wi] (d = 2) or [n, Wi] (d = 1) -> [n, Ho, Wo] / [n, Wo], F.interpolate(..., align_corners=False)"""
    x = _f32(x, "resize source")
    n = x.shape[0]
    Hi, Wi = (x.shape[1], x.shape[2]) if d == 2 else (1, x.shape[1])
    Ho, Wo = (out_hw[0], out_hw[1]) if d == 2 else (1, out_hw[-1])
    y = torch.empty((n, Ho, Wo) if d == 2 else (n, Wo), device=x.device, dtype=torch.float32)
    L.check(L.lib().magnet_resize_linear_f32(_p(x), n, Hi, Wi, Ho, Wo, d, _p(y), _stream()), "resize_linear")
    return y


def conv_tap_index(n_images, H, W, k, d, device):
    """[k^d, P] int32 neighbour rows of every pixel of a channels-last map, P = zero-padding row"""
    ntaps = k * k if d == 2 else k
    tap = torch.empty(ntaps, n_images * H * W, dtype=torch.int32, device=device)
    L.check(L.lib().magnet_conv_tap_index(n_images, H, W, k, d, _p(tap), _stream()), "conv_tap_index")
    return tap


# ------------------------------------------------------------------------------------ backward building blocks
def swish_fwd(z, out=None):
    out = torch.empty_like(z) if out is None else out
    L.check(L.lib().magnet_swish_fwd_f32(_p(z), _p(out), C.c_long(z.numel()), _stream()), "swish_fwd")
    return out


def swish_bwd(dr, z, out=None, amax=None):
    """dz = dr * Swish'(z); ``amax``: a zeroed float[256] that receives the block maxima of |dz| (-> linear_bwd128(amax_in=(amax, 256)))"""
    out = torch.empty_like(z) if out is None else out
    if amax is not None:
        L.check(L.lib().magnet_swish_bwd_amax_f32(_p(dr), _p(z), _p(out), C.c_long(z.numel()), _p(amax), _stream()), "swish_bwd")
    else:
        L.check(L.lib().magnet_swish_bwd_f32(_p(dr), _p(z), _p(out), C.c_long(z.numel()), _stream()), "swish_bwd")
    return out


def axpy(a, b, alpha=1.0, out=None):
    """out = a + alpha * b (out may be a)"""
    out = torch.empty_like(a) if out is None else out
    assert a.numel() == b.numel()
    L.check(L.lib().magnet_axpy_f32(_p(a), _p(b), C.c_float(alpha), _p(out), C.c_long(a.numel()), _stream()), "axpy")
    return out


def copy_cols(src, scol, dst, dcol, w, accumulate=False, scale=1.0):
    """dst[:, dcol:dcol+w] (+)= scale * src[:, scol:scol+w]"""
    rows = src.shape[0]
    if not ((w | scol | dcol | src.stride(0) | dst.stride(0)) & 3) and not ((src.data_ptr() | dst.data_ptr()) & 15):
        L.check(L.lib().magnet_copy_cols_scaled_f32(_p(src), src.stride(0), scol, _p(dst), dst.stride(0), dcol, w, C.c_long(rows),
                                                    C.c_float(float(scale)), int(accumulate), _stream()), "copy_cols")
        return dst
    assert scale == 1.0, "scaled column copies need 16-byte aligned blocks"
    L.check(L.lib().magnet_copy_cols_f32(_p(src), src.stride(0), scol, _p(dst), dst.stride(0), dcol, w, C.c_long(rows),
                                         int(accumulate), _stream()), "copy_cols")
    return dst


def edge_pre(A, Bm, csr, out=None):
    out = torch.empty(max(csr.E, 1), 128, device=A.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().magnet_edge_pre_f32(_p(A), _p(Bm), _p(csr.src), _p(csr.tgt), C.c_long(csr.E), _p(out), _stream()),
            "edge_pre")
    return out


def edge_mean_bwd(dagg, csr, out=None):
    out = torch.empty(max(csr.E, 1), 128, device=dagg.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().magnet_edge_mean_bwd_f32(_p(dagg), _p(csr.tgt), _p(csr.rowptr), C.c_long(csr.E), _p(out), _stream()),
            "edge_mean_bwd")
    return out


def mpnn_edge_bwd(A, Bm, dagg, csr, w2_f16, w2t_f16, b2, s1, dq, dp, dA, partials):
    """fused backward of the edge block: fills s1, dq, dp [E,128] and dA [N,128]"""
    L.check(L.lib().magnet_mpnn_edge_bwd_f32(_p(A), _p(Bm), _p(dagg), _p(csr.src), _p(csr.tgt), _p(csr.rowptr), csr.N, csr.E,
                                             _p(w2_f16), _p(w2t_f16), _p(b2), _p(s1), _p(dq), _p(dp), _p(dA), _p(partials),
                                             _stream()), "mpnn_edge_bwd")
    return dA


_EDGE_BWD_GEN = __import__("os").environ.get("MAGNET_EDGE_BWD_GEN")      # A/B knob: 2 (default) k_mpnn_edge_bwd_co, 1 k_mpnn_edge_bwd_wp, 0 k_mpnn_edge_bwd_w


def _edge_bwd_gen_knob():
    global _EDGE_BWD_GEN
    if _EDGE_BWD_GEN is not None:
        L.lib().magnet_mpnn_edge_bwd_w_generation(int(_EDGE_BWD_GEN))
        _EDGE_BWD_GEN = None


def mpnn_edge_bwd_w_takes_l2e(n_nodes):
    """whether mpnn_edge_bwd_w would read log2(e)-scaled halves for a graph of n_nodes (the kernel generation that serves it does)"""
    _edge_bwd_gen_knob()
    return bool(L.lib().magnet_mpnn_edge_bwd_w_takes_flags(int(n_nodes)))


def mpnn_edge_bwd_w(A, Bm, dagg, csr, w2_f16, w2t_f16, b2, dp, dA, partials, dW2, db2, ws=None, accumulate=True, amax_in=None, l2e=False):
    """fused backward of the edge block with message_net_2's weight gradient formed in the kernel: fills dp [E,128] and dA [N,128],
    dW2 [128,128] (+)= dq^T s1, db2 [128] (+)= colsum(dq); s1 / dq are never materialised.  ``l2e``: A / Bm are the forward's
    log2(e)-scaled halves"""
    assert dW2.is_contiguous() and dW2.shape == (128, 128) and db2.is_contiguous()
    _edge_bwd_gen_knob()
    if ws is None:
        ws = torch.empty(L.lib().magnet_mpnn_edge_bwd_w_workspace(), device=A.device, dtype=torch.float32)
    L.check(L.lib().magnet_mpnn_edge_bwd_w_flags_f32(_p(A), _p(Bm), _p(dagg), _p(csr.src), _p(csr.tgt), _p(csr.rowptr), csr.N, csr.E,
                                                     _p(w2_f16), _p(w2t_f16), _p(b2), _p(dp), _p(dA), _p(partials), _p(dW2), _p(db2),
                                                     int(accumulate), _p(ws), _p(amax_in[0]) if amax_in else None,
                                                     int(amax_in[1]) if amax_in else 0, int(bool(l2e)), _stream()), "mpnn_edge_bwd_w")
    return dA


def linear_bwd128(dz, x, wt_f16, dx, dW, db, ws, mask=True, amax_in=None, want_amax=True, accumulate=False):
    """one hidden W -> W Linear (W = dz.shape[1]: 128 or 64) walked back in one kernel: dx = (x > 0) * (dz W), dW = dz^T x,
    db = colsum(dz) (written, or added to what is there with ``accumulate``).  ``amax_in`` / the return value: (block maxima of |.|, count) handed from one layer's call to the
    next (None: computed here)"""
    M, W = dz.shape
    assert W in (128, 64) and x.shape[1] >= W and dW.is_contiguous() and dW.shape == (W, W) and db.is_contiguous()
    n_out = L.query("magnet_linear_bwd128_grid" if W == 128 else "magnet_linear_bwd64_grid", int(M))
    amax_out = torch.empty(max(n_out, 1), device=dz.device, dtype=torch.float32) if want_amax else None
    fn = L.lib().magnet_linear_bwd128_f32 if W == 128 else L.lib().magnet_linear_bwd64_f32
    L.check(fn(_p(dz), dz.stride(0), _p(x), x.stride(0), _p(wt_f16), C.c_long(M), int(bool(mask)), _p(dx), dx.stride(0), _p(dW), _p(db),
               int(bool(accumulate)),
               _p(amax_in[0]) if amax_in else None, int(amax_in[1]) if amax_in else 0, _p(amax_out), _p(ws), _stream()), "linear_bwd")
    return (amax_out, n_out) if want_amax else None


def linear_bwd128_pair(dz, x1, x2, wtA_f16, wtB_f16, dx, dWA, dbA, dWB, dbB, ws, amax_in=None, want_amax=True, accumulate=False):
    """two consecutive hidden W -> W Linears (A above B; W = dz.shape[1]: 128 or 64) walked back in one kernel: dx = (x2 > 0) * (((x1 > 0) * (dz W_A)) W_B), the two
    weight / bias gradients written (or added to with ``accumulate``); ``amax_in`` / return value as linear_bwd128"""
    M, W = dz.shape
    assert W in (128, 64) and x1.shape[1] >= W and x2.shape[1] >= W
    for t in (dWA, dWB):
        assert t.is_contiguous() and t.shape == (W, W)
    assert dbA.is_contiguous() and dbB.is_contiguous()
    n_out = L.query("magnet_linear_bwd128_pair_grid" if W == 128 else "magnet_linear_bwd64_pair_grid", int(M))
    amax_out = torch.empty(max(n_out, 1), device=dz.device, dtype=torch.float32) if want_amax else None
    fn = L.lib().magnet_linear_bwd128_pair_f32 if W == 128 else L.lib().magnet_linear_bwd64_pair_f32
    L.check(fn(_p(dz), dz.stride(0), _p(x1), x1.stride(0), _p(x2), x2.stride(0), _p(wtA_f16), _p(wtB_f16),
                                                  C.c_long(M), _p(dx), dx.stride(0), _p(dWA), _p(dbA), _p(dWB), _p(dbB), int(bool(accumulate)),
                                                  _p(amax_in[0]) if amax_in else None, int(amax_in[1]) if amax_in else 0, _p(amax_out),
                                                  _p(ws), _stream()), "linear_bwd_pair")
    return (amax_out, n_out) if want_amax else None


def segment_sum128(x, rowptr, n_rows, perm=None, out=None):
    out = torch.empty(n_rows, 128, device=x.device, dtype=torch.float32) if out is None else out
    L.check(L.lib().magnet_segment_sum128_f32(_p(x), _p(rowptr), _p(perm), n_rows, _p(out), _stream()), "segment_sum128")
    return out


def wgrad_workspace(M, device):
    nblk = min(768, (M + 31) // 32)
    return torch.empty(nblk * (128 * 160 + 128) + 128 * 160, device=device, dtype=torch.float32)      # (160: wgrad_pair's five k tiles)


def wgrad(dY, X, xcol, kc, dW, wcol, db, ws, accumulate=True, M=None):
    """dW[:, wcol:wcol+kc] (+)= dY^T X[:, xcol:xcol+kc]; db (+)= colsum(dY) (db may be None)."""
    M = dY.shape[0] if M is None else M
    assert dY.shape[1] == 128 and dW.shape[0] == 128 and dY.is_contiguous()
    L.check(L.lib().magnet_wgrad_f32(_p(dY), dY.stride(0), _p(X), X.stride(0), xcol, kc, C.c_long(M), _p(dW), dW.stride(0),
                                     wcol, _p(db), _p(ws), int(accumulate), _stream()), "wgrad")


def wgrad_pair(dY, X, xcol, X2, x2col, dW, wcol, wcol2, db, ws, accumulate=True):
    """dW[:, wcol:wcol+128] (+)= dY^T X[:, xcol:xcol+128] and dW[:, wcol2:wcol2+32] (+)= dY^T X2[:, x2col:x2col+32] in one pass over dY;
    db (+)= colsum(dY) (db may be None).  ``ws``: wgrad_workspace."""
    assert dY.shape[1] == 128 and dW.shape[0] == 128 and dY.is_contiguous()
    L.check(L.lib().magnet_wgrad_pair_f32(_p(dY), dY.stride(0), _p(X), X.stride(0), xcol, _p(X2), X2.stride(0), x2col,
                                          C.c_long(dY.shape[0]), _p(dW), dW.stride(0), wcol, wcol2, _p(db), _p(ws), int(accumulate),
                                          _stream()), "wgrad_pair")


def reduce_partials(partial, nblk, n, out, scale=1.0, accumulate=True):
    L.check(L.lib().magnet_reduce_partials_f32(_p(partial), nblk, C.c_long(n), C.c_float(scale), _p(out), int(accumulate),
                                               _stream()), "reduce_partials")


def instance_norm_bwd(dy, yhat, var, n_graphs, rows_per_graph, eps=1e-5, out=None):
    h = dy.shape[1]
    nchunks = (rows_per_graph + 127) // 128
    ws = torch.empty((2 * n_graphs * nchunks + 2 * n_graphs) * h, device=dy.device, dtype=torch.float32)
    out = torch.empty_like(dy) if out is None else out
    L.check(L.lib().magnet_instance_norm_bwd_f32(_p(dy), _p(yhat), _p(var), n_graphs, rows_per_graph, h,
                                                 C.c_float(float(eps)), _p(out), _p(ws), _stream()), "instance_norm_bwd")
    return out


def instance_norm_bwd_swish(dy, yhat, var, n_graphs, rows_per_graph, z, dz, amax=None, eps=1e-5, out=None):
    """instance_norm_bwd and swish_bwd(., z) in one apply pass: returns (dx, dz); ``amax`` as swish_bwd's"""
    h = dy.shape[1]
    nchunks = (rows_per_graph + 127) // 128
    ws = torch.empty((2 * n_graphs * nchunks + 2 * n_graphs) * h, device=dy.device, dtype=torch.float32)
    out = torch.empty_like(dy) if out is None else out
    L.check(L.lib().magnet_instance_norm_bwd_swish_f32(_p(dy), _p(yhat), _p(var), n_graphs, rows_per_graph, h, C.c_float(float(eps)),
                                                       _p(z), _p(out), _p(dz), _p(amax), _p(ws), _stream()), "instance_norm_bwd_swish")
    return out, dz


def mpnn_readout_bwd(h, dout, tw, w1, b1, k1, s1, w2, k2, use_swish, dt):
    """-> (dh [n,128], du_last [n], gpart [min(1024, ceil(n/4)), 8*k1+8+8*k2+1])"""
    n = h.shape[0]
    dh = torch.empty(n, 128, device=h.device, dtype=torch.float32)
    du = torch.empty(n, device=h.device, dtype=torch.float32)
    gp = torch.empty(min(1024, (n + 3) // 4), 8 * k1 + 8 + 8 * k2 + 1, device=h.device, dtype=torch.float32)
    L.check(L.lib().magnet_mpnn_readout_bwd_f32(_p(h), _p(_f32(dout)), n, tw, _p(w1), _p(b1), k1, s1, _p(w2), k2,
                                                int(use_swish), _p(_f32(dt)), _p(dh), _p(du), _p(gp), _stream()),
            "mpnn_readout_bwd")
    return dh, du, gp


# ------------------------------------------------------------------------------------ graphs
@dataclass
class CSR:
    """edges sorted by target: rowptr [N+1], src/tgt [E] (int32, device)."""
    rowptr: torch.Tensor
    src: torch.Tensor
    tgt: torch.Tensor
    N: int
    E: int
    perm: torch.Tensor | None = None   # CSR position -> input edge id (None when already sorted)


def graph_ptr(n_graphs: int, nodes_per_graph: int, device) -> torch.Tensor:
    return torch.arange(0, (n_graphs + 1) * nodes_per_graph, nodes_per_graph, dtype=torch.int32, device=device)


def radius_search(x: torch.Tensor, r, gptr: torch.Tensor, cap: int, brute_force: bool = False, ragged: bool = False):
    """first ``cap`` same-graph nodes with dist^2 < r^2 per node, ascending index.  ``ragged``: graphs of different sizes
    (no reference caller produces them; costs one host sync for the chunk count)."""
    x = _f32(x.reshape(x.shape[0], -1), "coords")
    n, dim = x.shape
    r64 = float(r)
    r2 = float(torch.tensor(r64 * r64, dtype=torch.float64).to(torch.float32))   # as torch_cluster passes it
    nbr = torch.empty(n * cap, dtype=torch.int32, device=x.device)
    cnt = torch.empty(n, dtype=torch.int32, device=x.device)
    ng = gptr.numel() - 1
    if brute_force:
        L.check(L.lib().magnet_radius_search_f32(_p(x), dim, n, _p(_i32(gptr)), ng, C.c_float(r2), cap,
                                                 _p(nbr), _p(cnt), _stream()), "radius_search")
        return nbr, cnt
    if ragged:
        sizes = (gptr[1:] - gptr[:-1]).long()
        cptr = torch.cat([torch.zeros(1, dtype=torch.long, device=x.device), torch.cumsum((sizes + 63) // 64, 0)]).int()
        nchunks = int(cptr[-1].item())
    else:
        # chunk table: graphs are equal-sized blocks in every caller, so the prefix is closed-form (no sync)
        per = (n // ng + 63) // 64 if ng else 0
        assert n % max(ng, 1) == 0, "equal-sized graphs expected (pass ragged=True otherwise)"
        cptr = torch.arange(0, (ng + 1) * per, max(per, 1), dtype=torch.int32, device=x.device)[: ng + 1]
        nchunks = ng * per
    bbox = torch.empty((max(nchunks, 1) + (nchunks + 63) // 64 + 1) * 6, dtype=torch.float32, device=x.device)   # chunk boxes + one box per 64 chunks
    L.check(L.lib().magnet_radius_search_bbox_f32(_p(x), dim, n, _p(_i32(gptr)), _p(cptr), ng, nchunks, C.c_float(r2), cap,
                                                  _p(bbox), _p(nbr), _p(cnt), _stream()), "radius_search_bbox")
    return nbr, cnt


def radius_pairs(x, r, gptr, loop: bool, max_num_neighbors: int = 32, brute_force: bool = False, ragged: bool = False):
    """(found, query, qptr, E): the edge list torch_cluster.radius_graph would return, query-major."""
    cap = max_num_neighbors if loop else max_num_neighbors + 1
    nbr, cnt = radius_search(x, r, gptr, cap, brute_force, ragged)
    n = cnt.numel()
    qptr = torch.empty(n + 1, dtype=torch.int32, device=nbr.device)
    L.check(L.lib().magnet_radius_offsets(_p(nbr), _p(cnt), n, cap, int(loop), _p(qptr), _stream()), "radius_offsets")
    E = int(qptr[-1].item())                          # one host sync per graph build
    found = torch.empty(max(E, 1), dtype=torch.int32, device=nbr.device)
    query = torch.empty(max(E, 1), dtype=torch.int32, device=nbr.device)
    L.check(L.lib().magnet_radius_emit_pairs(_p(nbr), _p(cnt), _p(qptr), n, cap, int(loop), _p(found), _p(query),
                                             _stream()), "radius_emit_pairs")
    return found[:E], query[:E], qptr, E


def csr_by_target(src: torch.Tensor, tgt: torch.Tensor, n_nodes: int) -> CSR:
    E = src.numel()
    dev = src.device
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int32, device=dev)
    cursor = torch.zeros(max(n_nodes, 1), dtype=torch.int32, device=dev)
    scratch = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    perm, s_out, t_out = (torch.empty(max(E, 1), dtype=torch.int32, device=dev) for _ in range(3))
    L.check(L.lib().magnet_csr_by_target(_p(_i32(src.contiguous())), _p(_i32(tgt.contiguous())), E, n_nodes, _p(rowptr),
                                         _p(cursor), _p(scratch), _p(perm), _p(s_out), _p(t_out), _stream()), "csr_by_target")
    return CSR(rowptr, s_out[:E], t_out[:E], n_nodes, E, perm[:E])


def csr_from_edge_index(edge_index: torch.Tensor, n_nodes: int) -> CSR:
    """PyG-style int64 [2, E] (row 0 = source j, row 1 = target i) -> CSR by target."""
    if not edge_index.is_cuda:
        raise RuntimeError("magnet_amd: edge_index must live on a HIP device")
    ei = edge_index.contiguous()
    E = ei.shape[1]
    both = torch.empty(2 * max(E, 1), dtype=torch.int32, device=ei.device)
    L.check(L.lib().magnet_i64_to_i32(_p(ei), 2 * E, _p(both), _stream()), "i64_to_i32")
    return csr_by_target(both[:E], both[E:2 * E], n_nodes)


def edge_index_from_pairs(row0: torch.Tensor, row1: torch.Tensor) -> torch.Tensor:
    E = row0.numel()
    ei = torch.empty(2, E, dtype=torch.int64, device=row0.device)
    L.check(L.lib().magnet_pairs_to_edge_index(_p(row0), _p(row1), E, _p(ei), _stream()), "pairs_to_edge_index")
    return ei


def knn(x, x_ptr, y, y_ptr, k: int):
    x, y = _f32(x.reshape(x.shape[0], -1)), _f32(y.reshape(y.shape[0], -1))
    m = y.shape[0]
    out = torch.empty(m, k, dtype=torch.int32, device=x.device)
    L.check(L.lib().magnet_knn_f32(_p(x), _p(_i32(x_ptr)), _p(y), _p(_i32(y_ptr)), x_ptr.numel() - 1, x.shape[1], m, k,
                                   _p(out), _stream()), "knn")
    return out


_DEV_NAMES = ("mpnn_edge32t", "mpnn_edge_ws", "RowTiles", "row_tiles", "mpnn_edge_rows", "ROW_TILE_PMIN")


def __getattr__(name):
    """the development library's wrappers under their old names (scripts/dev, scripts/bench_edge*.py): magnet_amd/dev_ops.py"""
    if name in _DEV_NAMES:
        try:
            from . import dev_ops
        except ImportError as e:                    # (hasattr(ops, ...) must answer False with the product library, not raise)
            raise AttributeError(str(e)) from None
        return getattr(dev_ops, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
