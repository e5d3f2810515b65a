// Per-tile bookkeeping shared by the fused edge kernels (forward: mpnn_edge_bf16.hip, backward: mpnn_edge_bwd.hip).
// A wave owns 32 consecutive CSR edges (sorted by target); lane (j = lane & 31, half = lane >> 5) holds edge j.
#pragma once
#include "common.cuh"

struct TileMeta {            // what the epilogue of a tile needs, carried one iteration
    int tile, ti, tnext;
    int rs, re;              // rowptr[ti], rowptr[ti + 1]: requested here, long before the epilogue needs them
    bool e_ok;
    float m1, m2, m4, m8, mb;
};

// the part of tile_meta that does not load anything: ti = tgt[min(e, E-1)], tnext = tgt[e+1] or -2, rs / re = rowptr[ti], rowptr[ti+1]
__device__ __forceinline__ TileMeta tile_meta_from(int tile, int E, int j, int ti, int tnext, int rs, int re) {
    TileMeta md;
    md.tile = tile;
    const int e_raw = tile * 32 + j;
    md.e_ok = e_raw < E;
    md.ti = ti;
    md.tnext = tnext;
    md.rs = rs;
    md.re = re;
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
__device__ __forceinline__ TileMeta tile_meta(int tile, int E, const int32_t* __restrict__ tgt,
                                              const int32_t* __restrict__ rowptr, int j) {
    const int e_raw = tile * 32 + j;
    const int ti = tgt[e_raw < E ? e_raw : E - 1];
    const int tnext = (e_raw + 1 < E && j < 31) ? tgt[e_raw + 1] : -2;
    return tile_meta_from(tile, E, j, ti, tnext, rowptr[ti], rowptr[ti + 1]);
}


// segmented inclusive sum of one register across the 32 row lanes (runs of equal targets)
__device__ __forceinline__ float seg_scan_sum(float x, const TileMeta& md) {
    x += md.m1 * dpp_f<DPP_SHR1, 0xF>(x);
    x += md.m2 * dpp_f<DPP_SHR2, 0xF>(x);
    x += md.m4 * dpp_f<DPP_SHR4, 0xF>(x);
    x += md.m8 * dpp_f<DPP_SHR8, 0xF>(x);
    x += md.mb * dpp_f<DPP_BCAST15, 0xA>(x);
    return x;
}

// After seg_scan_sum the last lane of every run holds the run's sum: a finished row of `out` (divided by the row's
// degree when MEAN) or a tile-boundary partial (partials[tile][0 = run entered from the previous tile | 1 = run continues
// into the next tile][D]); k_segmean_fixup / k_segsum_fixup combine the partials in tile order (deterministic).
template <int NT, bool MEAN>
__device__ __forceinline__ void store_run_results(const f32x16 (&v)[NT], const TileMeta& md, float* __restrict__ out,
                                                  float* __restrict__ partials, int half) {
    constexpr int D = NT * 32;
    if (md.e_ok && md.tnext != md.ti) {
        const int e0 = md.tile * 32;
        const int s = md.rs, en = md.re;
        const bool complete = (s >= e0) && (en <= e0 + 32);
        // one reciprocal per run instead of 32 * NT IEEE divisions (each a ~10-instruction sequence that every lane of the
        // wave pays although only the run-end lanes are active): mean = sum * RN(1 / degree), within 1 ulp of sum / degree
        const float inv = (MEAN && complete) ? 1.f / (float)(en - s) : 1.f;
        float* dst = complete ? out + (size_t)md.ti * D : partials + ((size_t)md.tile * 2 + (s < e0 ? 0 : 1)) * D;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o = {v[nt][4 * g], v[nt][4 * g + 1], v[nt][4 * g + 2], v[nt][4 * g + 3]};
                if (MEAN) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] *= inv;
                }
                *reinterpret_cast<f32x4*>(dst + nt * 32 + 8 * g + 4 * half) = o;
            }
    }
}

// CSR segment-mean epilogue of the MLP kernels: the wave's 32 rows are consecutive target-sorted edges of `tile`
// (md = tile_meta(...) computed at the top of the kernel so that its loads are hidden behind the GEMMs)
template <int NT>
__device__ __forceinline__ void segmean_epilogue_dpp(f32x16 (&v)[NT], const TileMeta& md, int M, float* __restrict__ agg,
                                                     float* __restrict__ partials, int lane) {
    if (md.tile * 32 >= M) return;                                // wave-uniform: no rows in this tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[nt][i] = seg_scan_sum(v[nt][i], md);
    store_run_results<NT, true>(v, md, agg, partials, lane >> 5);
}

// The same CSR segment mean for a 128-wide tile, through LDS instead of the DPP scan (which costs ten instructions plus
// hazard nops for each of the 64 registers of every lane): the tile goes to a per-wave LDS buffer 64 columns at a time
// (rows padded to 68 floats: conflict-free 16-byte writes and 4-byte column reads), every lane then walks ONE column down
// the 32 rows with a running sum, and run ends - wave-uniform, a ballot - leave under a scalar branch with a coalesced
// 256-byte store.  Same partials protocol as store_run_results (per tile), sums in edge order.
// buf: 32 * 68 floats owned by this wave; the caller guarantees nobody else is using that memory any more.
__device__ __forceinline__ void segmean_epilogue_lds128(const f32x16 (&v)[4], const TileMeta& md, int M, float* __restrict__ agg,
                                                        float* __restrict__ partials, float* __restrict__ buf, int lane) {
    if (md.tile * 32 >= M) return;                                // wave-uniform: no rows in this tile
    const int j = lane & 31, half = lane >> 5;
    const unsigned ends = (unsigned)__builtin_amdgcn_ballot_w64(md.e_ok && md.tnext != md.ti);   // both halves agree
    const int e0 = md.tile * 32;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& t = v[2 * p + q];
                const f32x4 o = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
                *reinterpret_cast<f32x4*>(buf + j * 68 + q * 32 + 8 * g + 4 * half) = o;
            }
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            acc += buf[e * 68 + lane];
            if (ends & (1u << e)) {
                const int ti = __builtin_amdgcn_readlane(md.ti, e);
                const int rs = __builtin_amdgcn_readlane(md.rs, e), re = __builtin_amdgcn_readlane(md.re, e);
                if (rs >= e0 && re <= e0 + 32) agg[(size_t)ti * 128 + p * 64 + lane] = acc / (float)(re - rs);
                else partials[((size_t)md.tile * 2 + (rs < e0 ? 0 : 1)) * 128 + p * 64 + lane] = acc;
                acc = 0.f;
            }
        }
    }
}

// acc[nt] (+ cross accumulators) += W(planes in LDS) * x, x = the wave's [128 x 32] register tile.
// While the matrix pipe works, x * store_scale is also written to HBM (row `dst`, may be null), one 16-byte chunk every
// other MFMA group: HBM absorbs ~9 bytes per cycle and CU, so a burst of 16 stores would stall the wave on the full
// memory queue; spread out, the stores drain behind the GEMM.
template <int MODE>
__device__ __forceinline__ void gemm128(const u32x4* __restrict__ wl, const f32x16 (&x)[4], f32x16 (&acc)[4], f32x16 (&accc)[4],
                                        int lane, float* __restrict__ dst, float store_scale) {
    constexpr int NP = Split<MODE>::NP;
    u32x4 xs[2][NP], wr[2][NP];
    split_slot<MODE>(x[0], 0, xs[0]);
#pragma unroll
    for (int q = 0; q < NP; ++q) wr[0][q] = wl[64 * q + lane];
#pragma unroll
    for (int gq = 0; gq < 32; ++gq) {             // gq = slot * 4 + nt, slot = kt * 2 + m
        const int slot = gq >> 2, nt = gq & 3;
        if (gq + 1 < 32) {
            const int s1 = (gq + 1) >> 2, n1 = (gq + 1) & 3;
            const u32x4* wp = wl + ((((s1 >> 1) * 4 + n1) * 2 + (s1 & 1)) * NP) * 64 + lane;
#pragma unroll
            for (int q = 0; q < NP; ++q) wr[(gq + 1) & 1][q] = wp[64 * q];
        }
        if (slot + 1 < 8) Split<MODE>::quarter(x[(slot + 1) >> 1], (slot + 1) & 1, nt, xs[(slot + 1) & 1]);
        Split<MODE>::mma(wr[gq & 1], xs[slot & 1], acc[nt], accc[nt]);
        if (dst && !(gq & 1)) {
            const int c = gq >> 1, t = c >> 2, g = c & 3;
            const f32x4 o = {x[t][4 * g] * store_scale, x[t][4 * g + 1] * store_scale, x[t][4 * g + 2] * store_scale,
                             x[t][4 * g + 3] * store_scale};
            *reinterpret_cast<f32x4*>(dst + t * 32 + 8 * g) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) Split<MODE>::fold(acc[nt], accc[nt]);
}

// ------------------------------------------------------------------------------------------------
// Bookkeeping of the edge kernels that carry their running sums across the tiles of a wave (generations 7-9:
// mpnn_edge_bf16.hip, mpnn_edge32t.hip)
struct RunMeta {             // per-lane bookkeeping of one tile (lane j <-> edge j), read back with readlane at run ends
    int ti, rs, re;
    unsigned ends;           // wave-uniform: bit e set <=> edge e is the last edge of its run (or of the wave's range)
};

// indices of one tile, requested two iterations before the tile is computed (nothing on the path waits for them)
struct TileIdx {
    int ti, sj, tnext;       // tgt[e], src[e] (clamped to the last edge), tgt[e + 1] or -2 at the end of the wave's range
    bool e_ok;
};
template <int TILE = 32>
__device__ __forceinline__ TileIdx tile_idx(int tile, int e_end, const int32_t* __restrict__ src,
                                            const int32_t* __restrict__ tgt, int j) {
    TileIdx x;
    const int e_raw = tile * TILE + j;
    x.e_ok = e_raw < e_end;
    const int e = x.e_ok ? e_raw : e_end - 1;
    x.ti = tgt[e];
    x.sj = src[e];
    x.tnext = (e_raw + 1 < e_end) ? tgt[e_raw + 1] : -2;      // runs continue across the tiles of one wave
    return x;
}

// The run that ends at edge e of the current tile.  A wave walks a contiguous range of tiles and its running sum simply
// continues across tile boundaries, so only the two runs that cross the ends of the wave's RANGE [e_lo, e_hi) leave as
// partial sums (partials[range][0 = entered from the previous range | 1 = continues into the next][128]); every other row
// leaves finished, divided by its degree.  k_edge_range_fixup combines the few partials in range order (deterministic).
__device__ __forceinline__ void emit_run(const f32x2 acc, const RunMeta& md, int e, int e_lo, int e_hi, int range,
                                         float deg_scale, float* __restrict__ agg, float* __restrict__ partials, int lane) {
    const int ti = __builtin_amdgcn_readlane(md.ti, e);
    const int rs = __builtin_amdgcn_readlane(md.rs, e), re = __builtin_amdgcn_readlane(md.re, e);
    f32x2 o = acc;
    float* dst;
    if (rs >= e_lo && re <= e_hi) {
        const float dn = (float)(re - rs) * deg_scale;
        o[0] = o[0] / dn;
        o[1] = o[1] / dn;
        dst = agg + (size_t)ti * 128;
    } else {
        dst = partials + ((size_t)range * 2 + (rs < e_lo ? 0 : 1)) * 128;
    }
    *reinterpret_cast<f32x2*>(dst + 2 * lane) = o;
}


// k_edge_range_fixup (mpnn_edge_bf16.hip): rows without edges become 0, the rows crossing range boundaries are assembled from
// `partials`.  tshift = log2(edges per tile), tpw / nranges exactly as the edge kernel computed them.
int magnet_launch_edge_range_fixup(const int32_t* tgt, const int32_t* rowptr, int n_rows, int n_edges, int tpw, int tshift,
                                   int nranges, float deg_scale, const float* partials, float* agg, void* stream);
