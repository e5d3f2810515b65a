// Fused MPNN edge kernel, generation 11: tiles aligned to CSR rows.
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// Generations 6-10 cut the target-sorted edge list into tiles of 16 / 32 CONSECUTIVE edges, so a tile straddles CSR rows: every
// tile carries run bookkeeping (targets of three tiles in flight, rowptr pairs, a ballot of run ends), the per-target mean is a
// segmented reduction (DPP scan, an LDS transposition tile or running sums with run-end branches), rows that cross a wave's range
// leave as partial sums for a fix-up kernel, and A[i] is gathered per EDGE although an MPNN row has 32 or 33 edges with one i.
//
// Here a tile is (up to) 32 edges of ONE row (magnet_amd/ops.py: row_tiles - every row with >= pmin edges gives one tile of its first
// min(deg, 32) edges; whatever is left - the 33rd edge of an interior row, rows shorter than pmin - stays an ordinary CSR edge list
// for k_mpnn_edge16, which runs first and leaves the mean of those edges in agg).  With that:
//   * the GEMM is transposed, D = X W2^T with v_mfma_f32_32x32x16_f16: the tile's edges are the ROWS of D, a lane owns an output
//     feature, and the mean over the tile is a plain sum over the lane's 16 accumulator registers plus one v_permlane32_swap -
//     no tile meta, no segmented scan, no run-end branch, no partials, no transposition through LDS;
//   * A[i] is one 512-byte row per tile: fetched once (8 bytes per lane), parked in LDS, read back as broadcast ds_read_b128 -
//     the per-lane gathers are Bm[j] only (half the vector-memory return traffic of generations 6-10);
//   * ONE fp32 accumulator for the three products of the fp16x3 split (weight planes pre-scaled by 2^11, unscaled low activation
//     piece: packing.pack_edge32t_fp16x2, the arithmetic of k_mpnn_edge32t), pieces straight out of the Swish (v_fma_mixlo/hi_f16);
//   * two waves per SIMD (512-thread workgroups, <= 256 registers): 32x32x16 MFMAs hide ~5 plain VALU instructions each
//     (scripts/dev/overlap_probe.cpp), a tile's 96 MFMAs take 36 instructions of next-tile input production between them, and
//     the epilogue (64 Swish per lane) runs as a phase of its own beside the other wave's MFMAs.
//
// Per tile and wave: 96 MFMA, 64 ds_read_b128 (W2 planes) + 16 (A), 16 gather loads, ~700 VALU.
#include "common.cuh"
#include <type_traits>

typedef unsigned int u32;

namespace {

constexpr float R_SCALE = 2048.0f;         // every MFMA product carries 2^11 (host-scaled weight planes)

__device__ __forceinline__ f32x16 mfma_xw(const u32x4& x, const u32x4& w, const f32x16& c) {
    // A operand = activations (row = edge = lane & 31), B operand = weights (column = feature = lane & 31)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, w), c, 0, 0, 0);
}

// Swish of the pre-activation pair (v0, v1) -> one word of the high plane and one of the low plane:
//   h = fp16(v r), l = fp16(v r - h), r = 1 / (1 + 2^-v')   (v' = v when the inputs carry log2 e, else v log2 e)
// (the wait states inside the asm: common.cuh, MAGNET_TRANS_HEAD / MAGNET_MIX_TAIL)
template <bool L2E>
__device__ __forceinline__ void swish_pieces(float v0, float v1, u32& hw, u32& lw) {
    const float e0 = __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E), e1 = __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E);
    const float r0 = __builtin_amdgcn_rcpf(1.0f + e0), r1 = __builtin_amdgcn_rcpf(1.0f + e1);
    u32 h, l;
    // one block for the four pieces: the leading wait state covers "transcendental result read in the next issue slot" for r0 / r1;
    // the plane words are read by MFMAs a whole tile later (no VALU -> MFMA wait states here; magnet_amd/hazard_check.py verifies)
    asm("s_nop 0\n\t"
        "v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %5, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l)
        : "v"(v0), "v"(v1), "v"(r0), "v"(r1));
    hw = h;
    lw = l;
}

// upper 32 lanes of a <-> lower 32 lanes of b (wait states on both sides: the operands come out of VALU instructions and go into them)
__device__ __forceinline__ void half_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

}  // namespace

// One tile descriptor (ops.row_tiles): e0 = first edge, row = its target, info = cnt | rem << 8 (cnt edges in the tile, 1..32;
// rem = edges of the row that are NOT in the tile - their mean is in agg[row] when the tile's epilogue runs), deg = cnt + rem.
//
// STRAGGLERS - the one edge a 33-edge row has beyond its tile (3 % of the edges; as CSR rows of one edge each they cost k_mpnn_edge16
// a run end per edge: 0.11 ms for 256 k edges, a tenth of the whole op; as a launch or a pipeline of their own 0.08 ms: four tiles per
// wave do not amortise a pipeline start).  They ride in the SAME pipeline: a wave's sequence of tiles is
//     S0 S1 .. (the stragglers of its own rows, 32 per tile)   R0 R1 ... (its row tiles)
// (strag_ptr[row] = stragglers of the rows before `row`; strag_row / strag_src sorted by row).  Every edge of an S tile has its own
// target, so its A side is gathered per lane like Bm, and its epilogue stores every edge's message to agg[its row] instead of
// summing; a row tile later finds the "mean of the edges outside the tile" there - written by this very wave whole tiles earlier
// (the stores are drained before the first row tile's epilogue), and nobody else touches those lines.  The iteration body is compiled
// for the three (current, next) type pairs that occur: (S, S), (S, R), (R, R).
//
// ABL (development builds only, results INVALID): 1 no MFMAs, 2 no epilogue Swish, 4 no input Swish / pieces, 8 no Bm gathers,
// 16 no LDS weight reads - the skeleton is what is left with all of them (kill criterion of the row-aligned kernel)
constexpr int S_FLAG = 1 << 30;            // descriptor word z of a straggler tile

template <bool L2E, int ABL = 0>
__global__ void __launch_bounds__(512) k_mpnn_edge_rows32(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          const int32_t* __restrict__ csr_src, const int4* __restrict__ tiles, int ntiles,
                                                          const int32_t* __restrict__ strag_ptr /* [rows + 1] or nullptr */,
                                                          const int32_t* __restrict__ strag_row, const int32_t* __restrict__ strag_src,
                                                          const u32x4* __restrict__ w2p /* [8 ks][4 nt][2 planes][64] */,
                                                          const float* __restrict__ b2, float* __restrict__ agg) {
    constexpr float l2e = L2E ? MAGNET_LOG2E : 1.0f;                       // the Swish outputs carry log2 e when the inputs do
    __shared__ __attribute__((aligned(16))) u32x4 wl[8 * 4 * 2 * 64];      // 64 KiB: both planes of W2, resident
    __shared__ __attribute__((aligned(16))) float arow[8][128];            // the A row of the row tile whose inputs are in production
    for (int i = threadIdx.x; i < 8 * 4 * 2 * 64; i += 512) wl[i] = w2p[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges (k_mpnn_edge6): consecutive tiles = neighbouring targets
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 8 - 1) / (gridDim.x * 8);
    const int t_lo = (vb * 8 + wave) * tpw, t_hi = min(t_lo + tpw, ntiles);
    if (t_lo >= t_hi) return;
    const int n_r = t_hi - t_lo;
    int s_lo = 0, n_s = 0;                 // this wave's stragglers: those of the rows of its tiles
    if (strag_ptr != nullptr) {
        s_lo = strag_ptr[tiles[t_lo].y];
        n_s = strag_ptr[tiles[t_hi - 1].y + 1] - s_lo;
    }
    const int ks_tiles = (n_s + 31) / 32;  // S tiles, then n_r row tiles
    const int K = ks_tiles + n_r;          // elements of this wave's sequence
    const u32x4* const wlane = wl + lane;
    float* const ar = arow[wave];
    const float* const arl = ar + 8 * half;            // this lane's columns of the parked row: 16 ks + 8 half + 0..7
    float bl[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl[nt] = b2[32 * nt + j] * l2e;

    // element k of the sequence (clamped to the last one, which is always a row tile)
    auto is_s = [&](int k) { return min(k, K - 1) < ks_tiles; };
    auto desc_at = [&](int k) {
        k = min(k, K - 1);
        if (k >= ks_tiles) return tiles[t_lo + k - ks_tiles];
        return make_int4(s_lo + 32 * k, 0, min(32, n_s - 32 * k) | S_FLAG, 1);
    };
    // per lane: the source index of the lane's edge and the pointer to its A data - the lane's own target row (S: columns 8 half ..)
    // or the tile's row (R: 8 bytes per lane, parked in LDS by the next iteration)
    auto edge_of = [&](const int4& d, bool s, int& idx, const float*& pa) {
        const int e = d.x + max(min(j, (d.z & 0xFF) - 1), 0);
        if (s) {
            idx = strag_src[e];
            pa = A + (size_t)strag_row[e] * 128 + 8 * half;
        } else {
            idx = csr_src[e];
            pa = A + (size_t)d.y * 128 + 2 * lane;
        }
    };

    // ---- pipeline state: the current tile's pieces; descriptors of the current and the next tile; the next tile's gather state
    u32x4 xh[8], xl[8];                    // fp16 pieces of the current tile's activations, one k step (16 columns) each
    f32x16 out[4];
    int4 d0, d1;
    int idx1;
    const float* pa1;

    auto swish2 = [&](float o, float bias) {
        const float y = fmaf(o, 1.0f / R_SCALE, bias);
        if (ABL & 2) return y;
        return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
    };

    // four values of a tile's input: columns 16 ks + 8 half + 4 part + 0..3  ->  words 2 part, 2 part + 1 of both planes
    auto pieces4 = [&](int ks, int part, const f32x4& a, const f32x4& b) {
        u32 h0, l0, h1, l1;
        if (ABL & 4) {
            h0 = __builtin_bit_cast(u32, a[0] + b[0]), l0 = __builtin_bit_cast(u32, a[1] + b[1]);
            h1 = __builtin_bit_cast(u32, a[2] + b[2]), l1 = __builtin_bit_cast(u32, a[3] + b[3]);
        } else {
            swish_pieces<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
            swish_pieces<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
        }
        xh[ks][2 * part] = h0, xh[ks][2 * part + 1] = h1;
        xl[ks][2 * part] = l0, xl[ks][2 * part + 1] = l1;
    };

    // ---- one iteration: GEMM + epilogue of the current tile (type CUR_S), input production of the next one (type NXT_S),
    // index / A-row prefetch of the one after (type known at run time)
    auto step = [&](auto cur_tag, auto nxt_tag, int k) __attribute__((always_inline)) {
        constexpr bool CUR_S = decltype(cur_tag)::value, NXT_S = decltype(nxt_tag)::value;
        f32x4 lb[2][2];                        // gathered Bm columns 16 ks + 8 half + 0..7 of one k step per set
        f32x4 lag[2][2];                       // NXT_S: the A columns likewise
        auto request = [&](int set, int ks, const float* pa, const float* pb) {
            if (ABL & 8) return;
            lb[set][0] = *reinterpret_cast<const f32x4*>(pb + 16 * ks);
            lb[set][1] = *reinterpret_cast<const f32x4*>(pb + 16 * ks + 4);
            if (NXT_S) {
                lag[set][0] = *reinterpret_cast<const f32x4*>(pa + 16 * ks);
                lag[set][1] = *reinterpret_cast<const f32x4*>(pa + 16 * ks + 4);
            }
        };
        auto produce = [&](int set, int ks, int part, const f32x4& a_lds) {
            const f32x4 a = NXT_S ? lag[set][part] : a_lds;
            pieces4(ks, part, a, (ABL & 8) ? a : lb[set][part]);
        };
        auto lds_a = [&](int off) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            return NXT_S ? z : *reinterpret_cast<const f32x4*>(arl + off);
        };
        // A row of the next tile -> LDS (the previous row was last read by the previous iteration's epilogue phase)
        if (!NXT_S) *reinterpret_cast<f32x2*>(ar + 2 * lane) = *reinterpret_cast<const f32x2*>(pa1);
        const bool s2 = is_s(k + 2);
        const int4 d2 = desc_at(k + 2);
        const float* pb = Bm + (size_t)idx1 * 128 + 8 * half;
        const float* pa = pa1;
        const int cnt = d0.z & 0xFF, rem = (d0.z & ~S_FLAG) >> 8;
        // what the row already holds (the mean of its edges outside the tile): consumed at the end of the iteration
        float old0 = 0.f, old1 = 0.f;
        if (!CUR_S && rem) old0 = agg[(size_t)d0.y * 128 + lane], old1 = agg[(size_t)d0.y * 128 + 64 + lane];
        int idx2 = 0;
        const float* pa2 = pa1;
#ifndef ROWS_WBUF
#define ROWS_WBUF 2                          // weight-fragment buffers: the fragments of double group dg + ROWS_WBUF - 1 are requested in dg
#endif
        u32x4 wr[ROWS_WBUF][2][2];            // weight fragments: [buffer][block of the pair][plane]
        f32x4 la = {0.f, 0.f, 0.f, 0.f};      // A columns of the four values the next double group produces
        auto wload = [&](int d) {
            if (ABL & 16) return;
            const int k1 = d >> 1, q1 = d & 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                wr[d % ROWS_WBUF][u][0] = wlane[((k1 * 4 + 2 * q1 + u) * 2 + 0) * 64];
                wr[d % ROWS_WBUF][u][1] = wlane[((k1 * 4 + 2 * q1 + u) * 2 + 1) * 64];
            }
        };
        if (ABL & 16) {
#pragma unroll
            for (int d = 0; d < ROWS_WBUF; ++d)
#pragma unroll
                for (int u = 0; u < 2; ++u) wr[d][u][0] = wlane[u * 64], wr[d][u][1] = wlane[(u + 2) * 64];
        }
#pragma unroll
        for (int d = 0; d < ROWS_WBUF - 1; ++d) wload(d);
        // ---- GEMM phase: double group dg = 2 ks + q handles k step ks for the output blocks 2 q and 2 q + 1; the six MFMAs
        // alternate between the two accumulators.  Beside them: the weight fragments of a later double group, four input values of the
        // next tile (k step (dg - 2) / 2 - its registers were last read by double group dg - 1 ...), the gather request of k step dg / 2.
#pragma unroll
        for (int dg = 0; dg < 16; ++dg) {
            const int ks = dg >> 1, q = dg & 1, n0 = 2 * q, n1 = 2 * q + 1, bsel = dg % ROWS_WBUF;
            if (dg + ROWS_WBUF - 1 < 16) wload(dg + ROWS_WBUF - 1);
            const f32x4 la_now = la;
            if (dg >= 1) la = lds_a(16 * ((dg - 1) >> 1) + 4 * ((dg - 1) & 1));   // for double group dg + 1
            __builtin_amdgcn_sched_barrier(0);    // LDS requests first
            if (!(ABL & 1)) {
                if (ks == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    out[n0] = mfma_xw(xh[0], wr[bsel][0][1], z);
                    out[n1] = mfma_xw(xh[0], wr[bsel][1][1], z);
                } else {
                    out[n0] = mfma_xw(xh[ks], wr[bsel][0][1], out[n0]);
                    out[n1] = mfma_xw(xh[ks], wr[bsel][1][1], out[n1]);
                }
                out[n0] = mfma_xw(xl[ks], wr[bsel][0][0], out[n0]);
                out[n1] = mfma_xw(xl[ks], wr[bsel][1][0], out[n1]);
                out[n0] = mfma_xw(xh[ks], wr[bsel][0][0], out[n0]);
                out[n1] = mfma_xw(xh[ks], wr[bsel][1][0], out[n1]);
            } else {                                // development: the operands stay alive, the matrix pipe stays idle
                asm volatile("" ::"v"(xh[ks]), "v"(xl[ks]), "v"(wr[bsel][0][0]), "v"(wr[bsel][0][1]), "v"(wr[bsel][1][0]), "v"(wr[bsel][1][1]));
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) out[n0][i] = 0.f, out[n1][i] = 0.f;
                }
            }
            if (dg >= 2) produce(((dg - 2) >> 1) & 1, (dg - 2) >> 1, dg & 1, la_now);
            if (q == 0) request(ks & 1, ks, pa, pb);       // set ks & 1 was drained by double groups 2 ks - 2, 2 ks - 1
            if (dg == 6) edge_of(d2, s2, idx2, pa2);        // the tile after next (d2 has arrived by now)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue phase: k step 7 of the next tile's inputs, then Swish of this tile's 64 outputs per lane
        produce(1, 7, 0, la);
        produce(1, 7, 1, lds_a(16 * 7 + 4));
        if (CUR_S) {                           // every edge's message to its own row; register i <-> edge 8 (i / 4) + 4 half + i % 4
            const int lim = cnt - 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int erow[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) erow[c] = (ABL & 64) ? d0.x + 8 * g + 4 * half + c : strag_row[d0.x + max(min(8 * g + 4 * half + c, cnt - 1), 0)];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float* dst = agg + (size_t)erow[c] * 128 + j;
                    float m[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) m[nt] = swish2(out[nt][4 * g + c], bl[nt]) * (1.0f / l2e);
                    if (ABL & 32) {
                        asm volatile("" ::"v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(dst));
                    } else if (8 * g + c < lim) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) dst[32 * nt] = m[nt];
                    }
                }
            }
        } else {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (cnt == 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float y = fmaf(out[nt][i], 1.0f / R_SCALE, bl[nt]);
                        if (ABL & 2) acc[nt] += y;
                        else acc[nt] = fmaf(y, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E)), acc[nt]);
                    }
            } else {                           // a short row: register i of lane half h is edge 8 (i / 4) + 4 h + i % 4
                const int lim = cnt - 4 * half;
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[nt] += (8 * (i >> 2) + (i & 3) < lim) ? swish2(out[nt][i], bl[nt]) : 0.f;
            }
            half_swap(acc[0], acc[1]);         // -> lanes 0-31: both halves of block 0, lanes 32-63: both halves of block 1
            half_swap(acc[2], acc[3]);
            const float invd = __builtin_amdgcn_rcpf((float)d0.w);
            const float wsum = invd * (1.0f / l2e), wold = (float)rem * invd;
            agg[(size_t)d0.y * 128 + lane] = fmaf(acc[0] + acc[1], wsum, old0 * wold);
            agg[(size_t)d0.y * 128 + 64 + lane] = fmaf(acc[2] + acc[3], wsum, old1 * wold);
        }
        d0 = d1, d1 = d2, idx1 = idx2, pa1 = pa2;
    };

    // ---- prologue: the first tile's inputs - every load requested before the first value is used (the accumulator registers are
    // free); the second tile's indices on their way
    d0 = desc_at(0);
    d1 = desc_at(1);
    {
        int i0;
        const float* pa0;
        const bool s0 = is_s(0);
        edge_of(d0, s0, i0, pa0);
        const float* pb = Bm + (size_t)i0 * 128 + 8 * half;
        if (!s0) *reinterpret_cast<f32x2*>(ar + 2 * lane) = *reinterpret_cast<const f32x2*>(pa0);
#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {           // two halves of four k steps: 32 (64 for an S tile) registers of loads in flight
            f32x4 gb[4][2], ga[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int off = 16 * (4 * hk + q) + 4 * part;
                    gb[q][part] = *reinterpret_cast<const f32x4*>(pb + off);
                    ga[q][part] = s0 ? *reinterpret_cast<const f32x4*>(pa0 + off) : *reinterpret_cast<const f32x4*>(arl + off);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int part = 0; part < 2; ++part) pieces4(4 * hk + q, part, ga[q][part], (ABL & 8) ? ga[q][part] : gb[q][part]);
        }
    }
    edge_of(d1, is_s(1), idx1, pa1);

    int k = 0;
    for (; k + 1 < ks_tiles; ++k) step(std::true_type{}, std::true_type{}, k);
    if (k < ks_tiles) {
        step(std::true_type{}, std::false_type{}, k++);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the messages have left before a row tile's epilogue asks for them
    }
    for (; k < K; ++k) step(std::false_type{}, std::false_type{}, k);
}

#ifdef MAGNET_DEV_VARIANTS
// ------------------------------------------------------------------------------------------------
// The same tile in two PURE phases (development builds only: measured 2 % slower than the interleaved kernel).  Measured on the interleaved kernel above (profiles/r4/edge_rows.txt): its time is the SUM of
// its matrix-pipe time, its LDS time, its gather time and its two Swish - nothing overlaps.  A wave issues in order, so the VALU /
// transcendental work placed between its MFMAs is stuck behind the wave's own next MFMA whenever the partner wave owns the matrix
// pipe, and a transcendental never runs beside an MFMA of either wave (scripts/dev/overlap_probe.cpp).  Here a tile is
//   M phase: the 96 MFMAs and their 64 weight-fragment reads, nothing else - the two waves of a SIMD alternate on the matrix pipe;
//   T phase: every transcendental - Swish of the tile's 64 outputs per lane, then gather + Swish + fp16 pieces of the NEXT tile's
//            inputs (all 16 Bm loads are requested at the start of the phase and land behind the output Swish).
// The waves of a workgroup run the same stream and stay in step, so matrix time meets matrix time and transcendental time meets
// transcendental time.
template <bool L2E, int ABL = 0>
__global__ void __launch_bounds__(512) k_mpnn_edge_rows32p(const float* __restrict__ A, const float* __restrict__ Bm,
                                                           const int32_t* __restrict__ src, const int4* __restrict__ tiles, int ntiles,
                                                           const u32x4* __restrict__ w2p /* [8 ks][4 nt][2 planes][64] */,
                                                           const float* __restrict__ b2, float* __restrict__ agg) {
    constexpr float l2e = L2E ? MAGNET_LOG2E : 1.0f;
    __shared__ __attribute__((aligned(16))) u32x4 wl[8 * 4 * 2 * 64];      // 64 KiB: both planes of W2, resident
    __shared__ __attribute__((aligned(16))) float arow[8][128];            // the A row of the tile whose inputs are produced next
    for (int i = threadIdx.x; i < 8 * 4 * 2 * 64; i += 512) wl[i] = w2p[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb8 = gridDim.x / 8;
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 8 - 1) / (gridDim.x * 8);
    int t = (vb * 8 + wave) * tpw;
    const int tend = min(t + tpw, ntiles);
    if (t >= tend) return;
    const u32x4* const wlane = wl + lane;
    float* const ar = arow[wave];
    const float* const arl = ar + 8 * half;
    float bl[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl[nt] = b2[32 * nt + j] * l2e;

    u32x4 xh[8], xl[8];
    f32x4 stg[8][2];                       // the next tile's gathered Bm columns, 16 ks + 8 half + 0..7 per k step
    f32x16 out[4];

    auto idx_of = [&](const int4& d) { return src[d.x + min(j, (d.z & 0xFF) - 1)]; };
    auto gather_all = [&](const float* pb) {
        if (ABL & 8) return;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            stg[ks][0] = *reinterpret_cast<const f32x4*>(pb + 16 * ks);
            stg[ks][1] = *reinterpret_cast<const f32x4*>(pb + 16 * ks + 4);
        }
    };
    auto produce_all = [&]() {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(arl + 16 * ks + 4 * part);
                const f32x4 b = (ABL & 8) ? a : stg[ks][part];
                u32 h0, l0, h1, l1;
                if (ABL & 4) {
                    h0 = __builtin_bit_cast(u32, a[0] + b[0]), l0 = __builtin_bit_cast(u32, a[1] + b[1]);
                    h1 = __builtin_bit_cast(u32, a[2] + b[2]), l1 = __builtin_bit_cast(u32, a[3] + b[3]);
                } else {
                    swish_pieces<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
                    swish_pieces<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
                }
                xh[ks][2 * part] = h0, xh[ks][2 * part + 1] = h1;
                xl[ks][2 * part] = l0, xl[ks][2 * part + 1] = l1;
            }
    };

    int4 d0 = tiles[t];
    int4 d1 = tiles[min(t + 1, tend - 1)];
    {
        const int i0 = idx_of(d0);
        const f32x2 a0 = *reinterpret_cast<const f32x2*>(A + (size_t)d0.y * 128 + 2 * lane);
        *reinterpret_cast<f32x2*>(ar + 2 * lane) = a0;
        gather_all(Bm + (size_t)i0 * 128 + 8 * half);
        produce_all();
    }
    int idx1 = idx_of(d1);
    f32x2 ar1 = *reinterpret_cast<const f32x2*>(A + (size_t)d1.y * 128 + 2 * lane);

    for (; t < tend; ++t) {
        *reinterpret_cast<f32x2*>(ar + 2 * lane) = ar1;     // A row of tile t + 1 (tile t's row was consumed by the previous T phase)
        const int4 d2 = tiles[min(t + 2, tend - 1)];
        // ---- M phase
        {
            u32x4 wr[2][2][2];                    // weight fragments: [buffer][block of the pair][plane]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                wr[0][u][0] = wlane[(u * 2 + 0) * 64];
                wr[0][u][1] = wlane[(u * 2 + 1) * 64];
            }
#pragma unroll
            for (int dg = 0; dg < 16; ++dg) {
                const int ks = dg >> 1, q = dg & 1, n0 = 2 * q, n1 = 2 * q + 1, bsel = dg & 1;
                if (dg + 1 < 16 && !(ABL & 16)) {
                    const int k1 = (dg + 1) >> 1, q1 = (dg + 1) & 1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        wr[(dg + 1) & 1][u][0] = wlane[((k1 * 4 + 2 * q1 + u) * 2 + 0) * 64];
                        wr[(dg + 1) & 1][u][1] = wlane[((k1 * 4 + 2 * q1 + u) * 2 + 1) * 64];
                    }
                }
                if (ABL & 1) {                          // development: the operands stay alive, the matrix pipe stays idle
                    asm volatile("" ::"v"(xh[ks]), "v"(xl[ks]), "v"(wr[bsel][0][0]), "v"(wr[bsel][0][1]), "v"(wr[bsel][1][0]), "v"(wr[bsel][1][1]));
                    if (ks == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) out[n0][i] = 0.f, out[n1][i] = 0.f;
                    }
                } else {
                    if (ks == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        out[n0] = mfma_xw(xh[0], wr[bsel][0][1], z);
                        out[n1] = mfma_xw(xh[0], wr[bsel][1][1], z);
                    } else {
                        out[n0] = mfma_xw(xh[ks], wr[bsel][0][1], out[n0]);
                        out[n1] = mfma_xw(xh[ks], wr[bsel][1][1], out[n1]);
                    }
                    out[n0] = mfma_xw(xl[ks], wr[bsel][0][0], out[n0]);
                    out[n1] = mfma_xw(xl[ks], wr[bsel][1][0], out[n1]);
                    out[n0] = mfma_xw(xh[ks], wr[bsel][0][0], out[n0]);
                    out[n1] = mfma_xw(xh[ks], wr[bsel][1][0], out[n1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- T phase
        gather_all(Bm + (size_t)idx1 * 128 + 8 * half);
        const int cnt = d0.z & 0xFF, rem = d0.z >> 8;
        float old0 = 0.f, old1 = 0.f;
        if (rem) old0 = agg[(size_t)d0.y * 128 + lane], old1 = agg[(size_t)d0.y * 128 + 64 + lane];
        const int idx2 = idx_of(d2);
        const f32x2 ar2 = *reinterpret_cast<const f32x2*>(A + (size_t)d2.y * 128 + 2 * lane);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (cnt == 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float y = fmaf(out[nt][i], 1.0f / R_SCALE, bl[nt]);
                    if (ABL & 2) acc[nt] += y;
                    else acc[nt] = fmaf(y, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E)), acc[nt]);
                }
        } else {                               // a short row: register i of lane half h is edge 8 (i / 4) + 4 h + i % 4
            const int lim = cnt - 4 * half;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float y = fmaf(out[nt][i], 1.0f / R_SCALE, bl[nt]);
                    const float s = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
                    acc[nt] += (8 * (i >> 2) + (i & 3) < lim) ? s : 0.f;
                }
        }
        half_swap(acc[0], acc[1]);             // -> lanes 0-31: both halves of block 0, lanes 32-63: both halves of block 1
        half_swap(acc[2], acc[3]);
        const float invd = __builtin_amdgcn_rcpf((float)d0.w);
        const float wsum = invd * (1.0f / l2e), wold = (float)rem * invd;
        agg[(size_t)d0.y * 128 + lane] = fmaf(acc[0] + acc[1], wsum, old0 * wold);
        agg[(size_t)d0.y * 128 + 64 + lane] = fmaf(acc[2] + acc[3], wsum, old1 * wold);
        __builtin_amdgcn_sched_barrier(0);
        produce_all();
        d0 = d1, d1 = d2, idx1 = idx2, ar1 = ar2;
    }
}

int g_edge_rows_ablation = 0;          // development builds only: ABL bits of k_mpnn_edge_rows32 (results invalid)
#endif

// A, Bm [n_rows, 128] fp32 (times log2 e when a_bm_log2e_scaled & 1); src [n_edges] the CSR's sources; tiles [n_tiles] int4 descriptors;
// strag_ptr [n_rows + 1] / strag_row / strag_src [n_strag]: the stragglers (magnet_row_tiles_fill; strag_ptr may be null when there are
// none); w2_t32 / b2 from packing.pack_edge32t_fp16x2.  agg [n_rows, 128]: rows whose tile leaves more than one edge outside must hold
// the mean over those edges on entry (k_mpnn_edge16 on the remainder list); rows with a tile receive the mean over all of their edges.
extern "C" int magnet_mpnn_edge_rows_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                                const int32_t* strag_ptr, const int32_t* strag_row, const int32_t* strag_src,
                                                const void* w2_t32, const float* b2, float* agg, int a_bm_log2e_scaled, void* stream) {
    (void)hipGetLastError();
    if (n_tiles <= 0) return 0;
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const dim3 g(min((n_tiles + 7) / 8, ncu)), b(512);
    const int4* td = reinterpret_cast<const int4*>(tiles);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
    hipStream_t s = (hipStream_t)stream;
    const bool l2e = (a_bm_log2e_scaled & 1) != 0;
#ifdef MAGNET_DEV_VARIANTS
#define ROWS_ABL(n) case n: hipLaunchKernelGGL((k_mpnn_edge_rows32<true, n>), g, b, 0, s, A, Bm, src, td, n_tiles, strag_ptr, strag_row, strag_src, ws, b2, agg); return (int)hipGetLastError();
#define ROWSP_ABL(n) case 100 + n: hipLaunchKernelGGL((k_mpnn_edge_rows32p<true, n>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg); return (int)hipGetLastError();
    switch (g_edge_rows_ablation) {        // development only: ablations produce invalid results; 100 + n: the phased kernel
        ROWS_ABL(1) ROWS_ABL(2) ROWS_ABL(4) ROWS_ABL(6) ROWS_ABL(16) ROWS_ABL(23) ROWS_ABL(32) ROWS_ABL(64) ROWS_ABL(96)
        ROWSP_ABL(0) ROWSP_ABL(1) ROWSP_ABL(2) ROWSP_ABL(4) ROWSP_ABL(6) ROWSP_ABL(8) ROWSP_ABL(16) ROWSP_ABL(31)
        default: break;
    }
#endif
    if (l2e) hipLaunchKernelGGL((k_mpnn_edge_rows32<true>), g, b, 0, s, A, Bm, src, td, n_tiles, strag_ptr, strag_row, strag_src, ws, b2, agg);
    else hipLaunchKernelGGL((k_mpnn_edge_rows32<false>), g, b, 0, s, A, Bm, src, td, n_tiles, strag_ptr, strag_row, strag_src, ws, b2, agg);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the tile plan (ops.row_tiles)
// per row: has a tile (deg >= pmin), is a straggler row (exactly one edge beyond the tile), edges for the remainder pass (everything
// outside the tile unless it is that one straggler; the whole row without a tile)
__global__ void __launch_bounds__(256) k_row_tiles_count(const int32_t* __restrict__ rowptr, int n_rows, int pmin,
                                                         int32_t* __restrict__ has_tile, int32_t* __restrict__ has_strag,
                                                         int32_t* __restrict__ n_rem, int write_tail = 0) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == n_rows && write_tail) has_tile[r] = has_strag[r] = n_rem[r] = 0;      // element n_rows of [n_rows + 1] arrays (scanned next)
    if (r >= n_rows) return;
    const int d = rowptr[r + 1] - rowptr[r];
    const int has = d >= pmin && d > 0;
    const int out = has ? d - min(d, 32) : d;
    has_tile[r] = has;
    has_strag[r] = has && out == 1;
    n_rem[r] = (has && out == 1) ? 0 : out;
}

// tile_ptr / strag_ptr / rem_ptr: exclusive prefix sums of the three arrays above ([n_rows + 1]); one thread per row writes its
// descriptor, its straggler and its remainder edges (CSR order is kept: the remainder list is again sorted by target).
__global__ void __launch_bounds__(256) k_row_tiles_fill(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, int n_rows,
                                                        const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ strag_ptr,
                                                        const int32_t* __restrict__ rem_ptr, int4* __restrict__ tiles,
                                                        int32_t* __restrict__ strag_row, int32_t* __restrict__ strag_src,
                                                        int32_t* __restrict__ src_rem, int32_t* __restrict__ tgt_rem) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int e0 = rowptr[r], d = rowptr[r + 1] - e0;
    const int has = tile_ptr[r + 1] - tile_ptr[r];
    const int take = has ? min(d, 32) : 0;
    if (has) tiles[tile_ptr[r]] = make_int4(e0, r, take | ((d - take) << 8), d);
    if (strag_ptr[r + 1] > strag_ptr[r]) {
        const int sp = strag_ptr[r];
        strag_row[sp] = r;
        strag_src[sp] = src[e0 + 32];
    } else {
        const int o = rem_ptr[r];
        for (int k = take; k < d; ++k) {
            src_rem[o + k - take] = src[e0 + k];
            tgt_rem[o + k - take] = r;
        }
    }
}

extern "C" int magnet_row_tiles_count(const int32_t* rowptr, int n_rows, int pmin, int32_t* has_tile, int32_t* has_strag, int32_t* n_rem,
                                      void* stream) {
    (void)hipGetLastError();
    if (n_rows > 0)
        hipLaunchKernelGGL(k_row_tiles_count, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, n_rows, pmin, has_tile,
                           has_strag, n_rem);
    return (int)hipGetLastError();
}

int magnet_launch_exclusive_scans(int32_t* v, int n, int narrays, long array_stride, void* stream);      // graph_kernels.hip

// The three prefix arrays of the plan in two launches: ptrs [3][n_rows + 1] int32 receives the exclusive prefix sums of has_tile /
// has_strag / n_rem (ptrs[k][n_rows] = the totals: tiles, stragglers, remainder edges) - what magnet_row_tiles_count + three
// host-side cumulative sums produced (0.66 ms per plan through torch; the bench rebuilds the plan with the graph in every rollout).
extern "C" int magnet_row_tiles_ptrs(const int32_t* rowptr, int n_rows, int pmin, int32_t* ptrs, void* stream) {
    (void)hipGetLastError();
    if (n_rows < 0) return -2;
    const long ld = (long)n_rows + 1;
    hipLaunchKernelGGL(k_row_tiles_count, dim3((n_rows + 256) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, n_rows, pmin, ptrs,
                       ptrs + ld, ptrs + 2 * ld, 1);
    return magnet_launch_exclusive_scans(ptrs, n_rows + 1, 3, ld, stream);
}

extern "C" int magnet_row_tiles_fill(const int32_t* rowptr, const int32_t* src, int n_rows, const int32_t* tile_ptr,
                                     const int32_t* strag_ptr, const int32_t* rem_ptr, void* tiles, int32_t* strag_row, int32_t* strag_src,
                                     int32_t* src_rem, int32_t* tgt_rem, void* stream) {
    (void)hipGetLastError();
    if (n_rows > 0)
        hipLaunchKernelGGL(k_row_tiles_fill, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, src, n_rows, tile_ptr,
                           strag_ptr, rem_ptr, reinterpret_cast<int4*>(tiles), strag_row, strag_src, src_rem, tgt_rem);
    return (int)hipGetLastError();
}
