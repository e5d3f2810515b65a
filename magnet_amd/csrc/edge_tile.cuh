// Per-tile bookkeeping shared by the fused edge kernels (forward: mpnn_edge_bf16.hip, backward: mpnn_edge_bwd.hip).
// A wave owns 32 consecutive CSR edges (sorted by target); lane (j = lane & 31, half = lane >> 5) holds edge j.
#pragma once
#include "common.cuh"

struct TileMeta {            // what the epilogue of a tile needs, carried one iteration
    int tile, ti, tnext;
    bool e_ok;
    float m1, m2, m4, m8, mb;
};

__device__ __forceinline__ TileMeta tile_meta(int tile, int E, const int32_t* __restrict__ tgt, int j) {
    TileMeta md;
    md.tile = tile;
    const int e_raw = tile * 32 + j;
    md.e_ok = e_raw < E;
    md.ti = tgt[md.e_ok ? e_raw : E - 1];
    md.tnext = (e_raw + 1 < E && j < 31) ? tgt[e_raw + 1] : -2;
    const int t = md.e_ok ? md.ti : (-16 - j);
    const int jr = j & 15;
    const int t1 = dpp_i<DPP_SHR1, 0xF>(t), t2 = dpp_i<DPP_SHR2, 0xF>(t), t4 = dpp_i<DPP_SHR4, 0xF>(t);
    const int t8 = dpp_i<DPP_SHR8, 0xF>(t), tb = dpp_i<DPP_BCAST15, 0xA>(t);
    md.m1 = ((jr >= 1) & (t1 == t)) ? 1.f : 0.f;
    md.m2 = ((jr >= 2) & (t2 == t)) ? 1.f : 0.f;
    md.m4 = ((jr >= 4) & (t4 == t)) ? 1.f : 0.f;
    md.m8 = ((jr >= 8) & (t8 == t)) ? 1.f : 0.f;
    md.mb = ((j >= 16) & (tb == t)) ? 1.f : 0.f;
    return md;
}

