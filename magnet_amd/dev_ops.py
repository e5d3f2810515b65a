"""Wrappers of the DEVELOPMENT library's extra entry points (include/magnet_hip_dev.h; `python -m magnet_amd.build --dev`,
MAGNET_DEV_LIB=1): the earlier generations of the fused MPNN edge op - k_mpnn_edge32t / 32p (9), k_mpnn_edge_ws (10), k_mpnn_edge_rows32
(11), k_mpnn_edge_rowsr / rowsg (12, 12b), k_mpnn_edge_rowsp (13) - kept for A/Bs and ablations.  The product
library contains none of them; importing this module without the development library raises."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib as L
from .ops import CSR, _p, _stream, edge_scratch_floats, mpnn_edge16

if not L.DEV:
    raise ImportError("magnet_amd.dev_ops needs the development library: build it with `python -m magnet_amd.build --dev` and set "
                      "MAGNET_DEV_LIB=1 (the product library has k_mpnn_edge16 and k_mpnn_edge_stream32 only)")


def mpnn_edge32t(A, Bm, csr, w2_t32, b2, agg, partials, l2e=False, paired=False):
    """the same op by the transposed 32-edge-tile kernel (k_mpnn_edge32t, generation 9, fp16x3 only: a lane owns an output
    feature, the mean is a running sum over accumulator registers, one accumulator per output block); w2_t32 from
    packing.pack_edge32t_fp16x2.  paired: the two-waves-per-SIMD variant k_mpnn_edge32p (the tile as two pairs of 32-feature blocks)"""
    assert w2_t32.dtype == torch.float16 and partials.numel() >= edge_scratch_floats(csr.E)
    L.check(L.lib().magnet_mpnn_edge_fp16x3_t32_f32(_p(A), _p(Bm), _p(csr.src), _p(csr.tgt), _p(csr.rowptr), csr.E, csr.N,
                                                    _p(w2_t32), _p(b2), _p(agg), _p(partials), int(l2e) | (2 if paired else 0),
                                                    _stream()),
            "mpnn_edge_fp16x3_t32")
    return agg


def mpnn_edge_ws(A, Bm, csr, w2_t32, b2, agg, partials, l2e=False):
    """the same op by the wave-specialised kernel (k_mpnn_edge_ws, generation 10: gather / GEMM / epilogue waves pipelined through
    LDS, fp16x3 only); w2_t32 from packing.pack_edge32t_fp16x2"""
    assert w2_t32.dtype == torch.float16 and partials.numel() >= edge_scratch_floats(csr.E)
    L.check(L.lib().magnet_mpnn_edge_fp16x3_ws_f32(_p(A), _p(Bm), _p(csr.src), _p(csr.tgt), _p(csr.rowptr), csr.E, csr.N,
                                                   _p(w2_t32), _p(b2), _p(agg), _p(partials), int(l2e), _stream()),
            "mpnn_edge_fp16x3_ws")
    return agg


@dataclass
class RowTiles:
    """Row-aligned tile plan of a CSR graph (k_mpnn_edge_rows32): every row with >= pmin edges owns one tile of its first
    min(deg, 32) edges.  A row with exactly one edge beyond its tile (the 33rd edge of an interior MPNN row) hands it to the
    straggler list; ``rem`` is the CSR of all other edges (rows shorter than pmin, everything beyond the 33rd edge)."""
    tiles: torch.Tensor            # [n_tiles, 4] int32 descriptors (include/magnet_hip.h)
    n_tiles: int
    strag_ptr: torch.Tensor        # [N + 1] int32: stragglers of the rows before row r
    strag_row: torch.Tensor        # [n_strag] int32
    strag_src: torch.Tensor
    n_strag: int
    rem: "CSR"                     # remainder edges, sorted by target
    pmin: int
    rem_pass: bool                 # the remainder pass must run (it also zero-fills the rows without any edge)


ROW_TILE_PMIN = 20                 # a partial tile costs a full one: shorter rows are cheaper in the remainder pass


def row_tiles(csr: "CSR", pmin: int = ROW_TILE_PMIN) -> RowTiles:
    """build the tile plan on the device (three launches - counts, their three prefix sums, the fill - and one host read of the counts)"""
    n, dev = csr.N, csr.rowptr.device
    ptr = torch.empty(3, n + 1, dtype=torch.int32, device=dev)
    L.check(L.lib().magnet_row_tiles_ptrs(_p(csr.rowptr), n, int(pmin), _p(ptr), _stream()), "row_tiles_ptrs")   # counts + their scans
    n_tiles, n_strag, n_rem = (int(v) for v in ptr[:, -1].tolist())         # one host sync per plan
    tiles = torch.empty(max(n_tiles, 1), 4, dtype=torch.int32, device=dev)
    strag_row, strag_src = (torch.empty(max(n_strag, 1), dtype=torch.int32, device=dev) for _ in range(2))
    src_rem = torch.empty(max(n_rem, 1), dtype=torch.int32, device=dev)
    tgt_rem = torch.empty(max(n_rem, 1), dtype=torch.int32, device=dev)
    L.check(L.lib().magnet_row_tiles_fill(_p(csr.rowptr), _p(csr.src), n, _p(ptr[0]), _p(ptr[1]), _p(ptr[2]), _p(tiles), _p(strag_row),
                                          _p(strag_src), _p(src_rem), _p(tgt_rem), _stream()), "row_tiles_fill")
    return RowTiles(tiles, n_tiles, ptr[1], strag_row, strag_src, n_strag, CSR(ptr[2], src_rem[:n_rem], tgt_rem[:n_rem], n, n_rem), int(pmin),
                    n_rem > 0 or n_tiles < n)


def mpnn_edge_rows(A, Bm, csr, plan: RowTiles, w2_t16, w2_t32, b2, agg, partials, l2e=False, single16=False, wreg=False):
    """the fused edge op by the row-aligned kernel (k_mpnn_edge_rows32).  Up to three launches: the remainder edges through
    k_mpnn_edge16 (leaves their mean in agg and zeros in the rows it has no edge for; skipped when every row has a tile and nothing
    is left over - every 2-D MPNN mesh), then ONE launch in which every wave first sends the stragglers of its rows through the
    pipeline (their messages go to agg[row]) and then its tiles (one per row: reads what the row holds, writes the mean over all
    of its edges).  w2_t16 / w2_t32: the two packs of message_net_2
    (packing.pack_edge16_fp16x2 / pack_edge32t_fp16x2)."""
    assert w2_t32.dtype == torch.float16
    if plan.rem_pass:
        mpnn_edge16(A, Bm, plan.rem, w2_t16, b2, agg, partials, l2e=l2e, single=single16)
    if wreg >= 2:     # generation 12b (k_mpnn_edge_rowsg): W2 in registers AND the stragglers grouped with their rows (no straggler lists);
        #               3 = generation 13 (k_mpnn_edge_rowsp): that as one software pipeline per wave (epilogue / production between the MFMAs)
        entry = L.lib().magnet_mpnn_edge_rowsp_fp16x3_f32 if wreg == 3 else L.lib().magnet_mpnn_edge_rowsg_fp16x3_f32
        L.check(entry(_p(A), _p(Bm), _p(csr.src), _p(plan.tiles), plan.n_tiles, int(plan.n_strag > 0),
                                                          _p(w2_t32), _p(b2), _p(agg), int(l2e), _stream()), "mpnn_edge_rowsg_fp16x3")
        return agg
    # wreg: generation 12 (k_mpnn_edge_rowsr: W2's planes in the accumulation registers, one wave per SIMD) - same arguments
    entry = L.lib().magnet_mpnn_edge_rowsr_fp16x3_f32 if wreg else L.lib().magnet_mpnn_edge_rows_fp16x3_f32
    L.check(entry(_p(A), _p(Bm), _p(csr.src), _p(plan.tiles), plan.n_tiles,
                                                     _p(plan.strag_ptr) if plan.n_strag else None, _p(plan.strag_row), _p(plan.strag_src),
                                                     _p(w2_t32), _p(b2), _p(agg), int(l2e), _stream()),
            "mpnn_edge_rows_fp16x3")
    return agg
