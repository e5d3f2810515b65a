// Fused MPNN edge kernel, generation 14: 32-edge tiles cut from the CSR edge STREAM of a row-aligned chunk, two rows per tile.
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// Generation 11 (mpnn_edge_rows.hip) gives every CSR row ONE tile of its first 32 edges: tiles alone it beats k_mpnn_edge16 by 8-9 %
// (0.95 against 1.04 ms on the metric graph), but an interior row of the 2-D MPNN meshes has 33 edges, and the 33rd edges - as
// straggler tiles in the same pipeline, as a grouped 16-edge tile (generation 12b) or as a remainder list - give the whole gain back
// (DESIGN.md 11.1, 14.2).  Here no edge is left over: every wave owns a CHUNK of whole rows (magnet_stream_tiles_plan: the rows are
// cut into n_chunks ranges of about equal edge count) and walks the chunk's edges 32 at a time, whatever the row boundaries are.  A
// tile then holds the last `split` edges of row r and the first cnt - split edges of row r + 1 (the plan never lets a tile touch a
// third row: it cuts the tile short instead, which the 30-33-edge rows of the meshes never need), and what generation 11 got from
// row alignment survives:
//   * transposed GEMM, lane = output feature: the tile's sum is a sum over the lane's 16 accumulator registers.  The lane <-> edge
//     map inside a tile is chosen so that register i of the two half-waves holds edges 2 i and 2 i + 1: the first `split` edges are
//     then registers 0 .. split / 2 - 1 (plus half a register when split is odd), i.e. the two-segment sum is the running sum with
//     ONE snapshot in front of register split / 2 - a uniform branch per register, nine VALU instructions once or twice per tile;
//   * A[i]: two 512-byte rows per tile parked in LDS (a tile's second row is the next tile's first), the lane picks its row by
//     comparing its edge with `split` - still one A row fetched per tile;
//   * a row's sum is finished by the wave that started it (carry registers between consecutive tiles; chunks begin and end at row
//     boundaries): no partial sums, no fix-up kernel, no read-modify-write of agg, no straggler lists, one code path.
// Everything else is generation 11: ONE fp32 accumulator for the three products of the fp16x3 split (weight planes pre-scaled by
// 2^11: packing.pack_edge32t_fp16x2), W2's planes resident in LDS, two waves per SIMD, next-tile input production between the MFMAs.
#include "common.cuh"
#include <type_traits>

typedef unsigned int u32;

#ifndef ES_PACKED
#define ES_PACKED 0                     // 1: packed fp32 adds in the input production (1 + e, A + Bm; same bits) - measured SLOWER beside
#endif                                  //    the MFMAs: 1.001 ms against 0.989 (round 6, DESIGN.md 10 rule 4 again)

namespace {

constexpr float R_SCALE = 2048.0f;         // every MFMA product carries 2^11 (host-scaled weight planes)
constexpr int AR_STRIDE = 132;             // floats between the two parked A rows (16-byte reads of the two rows hit different banks)

__device__ __forceinline__ f32x16 mfma_xw(const u32x4& x, const u32x4& w, const f32x16& c) {
    // A operand = activations (row = edge slot = lane & 31), B operand = weights (column = feature = lane & 31)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, w), c, 0, 0, 0);
}

// Swish of the pre-activation pair (v0, v1) -> one word of the high plane and one of the low plane (mpnn_edge_rows.hip: swish_pieces)
template <bool L2E>
__device__ __forceinline__ void swish_pieces(float v0, float v1, u32& hw, u32& lw) {
    const float e0 = __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E), e1 = __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E);
#if ES_PACKED
    const f32x2 d = f32x2{e0, e1} + 1.0f;                                     // (one v_pk_add_f32 for the pair: same bits, one issue slot less)
    const float r0 = __builtin_amdgcn_rcpf(d[0]), r1 = __builtin_amdgcn_rcpf(d[1]);
#else
    const float r0 = __builtin_amdgcn_rcpf(1.0f + e0), r1 = __builtin_amdgcn_rcpf(1.0f + e1);
#endif
    u32 h, l;
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

// upper 32 lanes of a <-> lower 32 lanes of b
__device__ __forceinline__ void half_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

}  // namespace

// Tile descriptor (k_stream_tiles_plan): x = first edge, y = row r of that edge, z = cnt | split << 8 | ends0 << 16 | ends1 << 17
// (cnt edges in the tile, the first `split` of them belong to row r, the others to row r + 1; ends0 / ends1: row r / row r + 1 has its
// last edge in this tile), w = deg(r) | deg(r + 1) << 16.
constexpr int ST_ENDS0 = 1 << 16, ST_ENDS1 = 1 << 17;

// LBD: gather sets in flight - 2: generation 11's schedule (the Bm columns of k step ks are requested two double groups before they are
// turned into pieces); 3 / 4: four / six double groups ahead (the first k steps of a tile already in the previous tile's epilogue),
// eight registers more per set
template <bool L2E, int LBD = 2>
__global__ void __launch_bounds__(512) k_mpnn_edge_stream32(const float* __restrict__ A, const float* __restrict__ Bm,
                                                            const int32_t* __restrict__ csr_src, const int4* __restrict__ tiles,
                                                            const int32_t* __restrict__ chunk_ntiles, int tpc, int n_chunks, int n_rows,
                                                            const u32x4* __restrict__ w2p /* [8 ks][4 nt][2 planes][64] */,
                                                            const float* __restrict__ b2, float* __restrict__ agg, int young_prio) {
    constexpr float l2e = L2E ? MAGNET_LOG2E : 1.0f;                       // the Swish outputs carry log2 e when the inputs do
    __shared__ __attribute__((aligned(16))) u32x4 wl[8 * 4 * 2 * 64];      // 64 KiB: both planes of W2, resident
    __shared__ __attribute__((aligned(16))) float arow[8][2 * AR_STRIDE];  // per wave: the two A rows of the tile whose inputs are in production
    for (int i = threadIdx.x; i < 8 * 4 * 2 * 64; i += 512) wl[i] = w2p[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges (k_mpnn_edge6): consecutive chunks = neighbouring targets
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int chunk = vb * 8 + wave;
    if (chunk >= n_chunks) return;
    const int K = __builtin_amdgcn_readfirstlane(chunk_ntiles[chunk]);
    if (K <= 0) return;
    const int4* const tb = tiles + (size_t)chunk * tpc;
    const u32x4* const wlane = wl + lane;
    float* const ar = arow[wave];
    // the edge slot of this lane's GEMM row: row m = lane & 31 of the A operand comes back as register 4 (m >> 3) + (m & 3) of half-wave
    // (m >> 2) & 1, and register i of half h is to hold edge 2 i + h
    const int pj = 2 * (4 * (j >> 3) + (j & 3)) + ((j >> 2) & 1);
    float bl[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl[nt] = b2[32 * nt + j] * l2e;

    auto desc_at = [&](int k) { return tb[min(k, K - 1)]; };
    auto cnt_of = [](const int4& d) { return d.z & 0xFF; };
    auto split_of = [](const int4& d) { return (d.z >> 8) & 0xFF; };
    // per lane: the source index of the lane's edge (a slot beyond a short tile repeats its last edge: masked in the epilogue)
    auto src_of = [&](const int4& d) { return csr_src[d.x + min(pj, cnt_of(d) - 1)]; };
    // this lane's 8 bytes of the tile's two A rows (row r + 1 clamped: the last row's tile has no second row)
    auto a_rows = [&](const int4& d, f32x2& a0, f32x2& a1) {
        a0 = *reinterpret_cast<const f32x2*>(A + (size_t)d.y * 128 + 2 * lane);
        a1 = *reinterpret_cast<const f32x2*>(A + (size_t)min(d.y + 1, n_rows - 1) * 128 + 2 * lane);
    };
    auto park = [&](const int4& d, const f32x2& a0, const f32x2& a1) {        // row r -> slot r & 1, row r + 1 -> the other one
        *reinterpret_cast<f32x2*>(ar + (d.y & 1) * AR_STRIDE + 2 * lane) = a0;
        *reinterpret_cast<f32x2*>(ar + ((d.y + 1) & 1) * AR_STRIDE + 2 * lane) = a1;
    };
    // where this lane reads its A columns of tile d: the slot of its edge's row, columns 8 half ..
    auto arl_of = [&](const int4& d) { return ar + ((d.y + (pj >= split_of(d) ? 1 : 0)) & 1) * AR_STRIDE + 8 * half; };

    // ---- pipeline state
    u32x4 xh[8], xl[8];                    // fp16 pieces of the current tile's activations, one k step (16 columns) each
    f32x16 out[4];
    int4 d0, d1;
    int idx1;
    f32x2 an0, an1;                        // A rows of the next tile (on their way to LDS)
    float carry[4] = {0.f, 0.f, 0.f, 0.f}; // this half-wave's part of the sum over the edges of the row that is still open
    f32x4 lb[LBD][2];                      // gathered Bm columns 16 ks + 8 half + 0..7 of one k step per set (set = ks % LBD)
    auto request = [&](int ks, const float* pb) {
        lb[ks % LBD][0] = *reinterpret_cast<const f32x4*>(pb + 16 * ks);
        lb[ks % LBD][1] = *reinterpret_cast<const f32x4*>(pb + 16 * ks + 4);
    };

    auto swish2 = [&](float o, float bias) {
        const float y = fmaf(o, 1.0f / R_SCALE, bias);
        return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
    };

    // four values of a tile's input: columns 16 ks + 8 half + 4 part + 0..3  ->  words 2 part, 2 part + 1 of both planes
    auto pieces4 = [&](int ks, int part, const f32x4& a, const f32x4& b) {
        u32 h0, l0, h1, l1;
#if ES_PACKED
        const f32x4 v = a + b;                                                // two v_pk_add_f32
        swish_pieces<L2E>(v[0], v[1], h0, l0);
        swish_pieces<L2E>(v[2], v[3], h1, l1);
#else
        swish_pieces<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
        swish_pieces<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
#endif
        xh[ks][2 * part] = h0, xh[ks][2 * part + 1] = h1;
        xl[ks][2 * part] = l0, xl[ks][2 * part + 1] = l1;
    };

    // ---- one iteration: GEMM + epilogue of the current tile, input production of the next one, index / A-row prefetch of the one after
    // ES_ROLES (development ablation, results invalid; profiles/r6/edge_stream_roles.txt): DO_M = this wave issues the tile's MFMAs (and their
    // weight-fragment reads), DO_V = everything else (gathers, input production, epilogue)
    auto step = [&](int k, auto do_m, auto do_v) __attribute__((always_inline)) {
        constexpr bool DO_M = decltype(do_m)::value, DO_V = decltype(do_v)::value;
        // the next tile's A rows -> LDS (the previous pair was last read by the previous iteration's epilogue phase)
        if (DO_V) park(d1, an0, an1);
        const float* const arl = arl_of(d1);
        auto lds_a = [&](int off) { return *reinterpret_cast<const f32x4*>(arl + off); };
        const int4 d2 = desc_at(k + 2);
        const float* pb = Bm + (size_t)idx1 * 128 + 8 * half;
        int idx2 = 0;
        u32x4 wr[2][2][2];                    // weight fragments: [buffer][block of the pair][plane]
        f32x4 la = {0.f, 0.f, 0.f, 0.f};      // A columns of the four values the next double group produces
        auto wload = [&](int d) {
            const int k1 = d >> 1, q1 = d & 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                wr[d % 2][u][0] = wlane[((k1 * 4 + 2 * q1 + u) * 2 + 0) * 64];
                wr[d % 2][u][1] = wlane[((k1 * 4 + 2 * q1 + u) * 2 + 1) * 64];
            }
        };
        if (DO_M) wload(0);
        // ---- GEMM phase (generation 11's): double group dg = 2 ks + q handles k step ks for the output blocks 2 q and 2 q + 1
#pragma unroll
        for (int dg = 0; dg < 16; ++dg) {
            const int ks = dg >> 1, q = dg & 1, n0 = 2 * q, n1 = 2 * q + 1, bsel = dg % 2;
            if (DO_M && dg + 1 < 16) wload(dg + 1);
            const f32x4 la_now = la;
            if (DO_V && dg >= 1) la = lds_a(16 * ((dg - 1) >> 1) + 4 * ((dg - 1) & 1));   // for double group dg + 1
            __builtin_amdgcn_sched_barrier(0);    // LDS requests first
            if (DO_M) {
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
            if (!DO_V) {
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if (dg >= 2) pieces4((dg - 2) >> 1, dg & 1, la_now, lb[((dg - 2) >> 1) % LBD][dg & 1]);
            // k step ks + LBD - 2: its set was drained by double groups 2 ks - 2, 2 ks - 1 (the first LBD - 2 k steps of a tile are
            // requested in the previous tile's epilogue)
            if (q == 0 && ks + LBD - 2 < 8) request(ks + LBD - 2, pb);
            if (dg == 6) {                                  // the tile after next (d2 has arrived by now)
                idx2 = src_of(d2);
                a_rows(d2, an0, an1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!DO_V) {                           // (ablation: the matrix stream alone - its results stay alive, nothing reads them)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(out[nt]));
            d0 = d1, d1 = d2;
            return;
        }
        // ---- epilogue phase: k step 7 of the next tile's inputs, then Swish of this tile's 64 outputs per lane
        pieces4(7, 0, la, lb[7 % LBD][0]);
        pieces4(7, 1, lds_a(16 * 7 + 4), lb[7 % LBD][1]);
        if (!DO_M) {                           // (ablation: everything but the matrix stream - the pieces nobody multiplies stay alive, the
#pragma unroll                                 //  accumulators the epilogue reads are "new" every tile)
            for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(xh[ks]), "+v"(xl[ks]));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(out[nt]));
        }
        if (LBD > 2) {                         // the first k steps of the tile after next: in flight through this epilogue
            const float* pb2 = Bm + (size_t)idx2 * 128 + 8 * half;
#pragma unroll
            for (int ks = 0; ks < LBD - 2; ++ks) request(ks, pb2);
        }
        const int cnt = __builtin_amdgcn_readfirstlane(cnt_of(d0)), split = __builtin_amdgcn_readfirstlane(split_of(d0));
        const int zf = __builtin_amdgcn_readfirstlane(d0.z);
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, snap[4] = {0.f, 0.f, 0.f, 0.f};
        if (cnt == 32) {
            // register i = edges 2 i (lower half-wave) and 2 i + 1 (upper): a lane's values belong to row r up to register p - 1 and to
            // row r + 1 from register p on, p = split / 2 (+ 1 in the lower half-wave when split is odd).  The running sum is
            // snapshotted (snap = acc, acc = 0) by the lanes with p == i right before register i is added: a uniform branch around
            // nine VALU instructions, taken once (twice: odd split) per tile - nothing else in the sum knows about rows.
            const int n0 = split >> 1, odd = split & 1;
            const unsigned at = (1u << n0) | ((unsigned)odd << (n0 + 1));       // the registers in front of which some lanes snapshot (bit 16: after the last)
            const int p = n0 + ((odd && half == 0) ? 1 : 0);
#pragma unroll
            for (int i = 0; i <= 16; ++i) {
                if (__builtin_expect((at & (1u << i)) != 0, 0)) {
                    asm volatile("" ::: "memory");      // (a real branch: hipcc otherwise turns the 17 blocks into 136 unconditional selects)
                    const bool me = p == i;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) snap[nt] = me ? acc[nt] : snap[nt], acc[nt] = me ? 0.f : acc[nt];
                }
                if (i < 16) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float y = fmaf(out[nt][i], 1.0f / R_SCALE, bl[nt]);
                        acc[nt] = fmaf(y, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E)), acc[nt]);
                    }
                }
            }
        } else {                               // a short tile (the end of a chunk, a cut in front of a third row): every value masked
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = 2 * i + half;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float s = swish2(out[nt][i], bl[nt]);
                    snap[nt] += e < split ? s : 0.f;
                    acc[nt] += (e >= split && e < cnt) ? s : 0.f;
                }
            }
        }
        // row r: what the previous tiles left + this tile's first segment
        float tot[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) tot[nt] = carry[nt] + snap[nt];
        if (zf & ST_ENDS0) {
            half_swap(tot[0], tot[1]);         // -> lanes 0-31: both halves of block 0, lanes 32-63: both halves of block 1
            half_swap(tot[2], tot[3]);
            const float w0 = __builtin_amdgcn_rcpf((float)(d0.w & 0xFFFF)) * (1.0f / l2e);
            agg[(size_t)d0.y * 128 + lane] = (tot[0] + tot[1]) * w0;
            agg[(size_t)d0.y * 128 + 64 + lane] = (tot[2] + tot[3]) * w0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) carry[nt] = acc[nt];
        } else {                               // row r goes on in the next tile (split == cnt: there is no second segment)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) carry[nt] = tot[nt];
        }
        if (zf & ST_ENDS1) {                   // row r + 1 ends here as well (chunk end, a cut)
            half_swap(carry[0], carry[1]);
            half_swap(carry[2], carry[3]);
            const float w1 = __builtin_amdgcn_rcpf((float)((d0.w >> 16) & 0xFFFF)) * (1.0f / l2e);
            agg[(size_t)(d0.y + 1) * 128 + lane] = (carry[0] + carry[1]) * w1;
            agg[(size_t)(d0.y + 1) * 128 + 64 + lane] = (carry[2] + carry[3]) * w1;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) carry[nt] = 0.f;
        }
        d0 = d1, d1 = d2, idx1 = idx2;
    };

    // ---- prologue: the first tile's inputs - every load requested before the first value is used; the second tile's indices and
    // A rows on their way
    d0 = desc_at(0);
    d1 = desc_at(1);
    {
        const int i0 = src_of(d0);
        f32x2 a0, a1;
        a_rows(d0, a0, a1);
        const float* pb = Bm + (size_t)i0 * 128 + 8 * half;
        park(d0, a0, a1);
        const float* const arl = arl_of(d0);
#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {           // two halves of four k steps: 32 registers of loads in flight
            f32x4 gb[4][2], ga[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int off = 16 * (4 * hk + q) + 4 * part;
                    gb[q][part] = *reinterpret_cast<const f32x4*>(pb + off);
                    ga[q][part] = *reinterpret_cast<const f32x4*>(arl + off);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int part = 0; part < 2; ++part) pieces4(4 * hk + q, part, ga[q][part], gb[q][part]);
        }
    }
    idx1 = src_of(d1);
    a_rows(d1, an0, an1);
    // static priority for the second-dispatched wave of every SIMD (waves 4-7): at equal priority the issue arbiter serves the OLDER
    // wave first, and the younger one is the loser of every phase (MI355X_MICROARCH.md, "static priority for the younger half")
    if (young_prio && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (LBD > 2) {
        const float* pb1 = Bm + (size_t)idx1 * 128 + 8 * half;
#pragma unroll
        for (int ks = 0; ks < LBD - 2; ++ks) request(ks, pb1);
    }
#ifndef ES_ROLES
#define ES_ROLES 0
#endif
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (ES_ROLES == 0) {
        for (int k = 0; k < K; ++k) step(k, T_{}, T_{});
    } else if ((ES_ROLES == 4 && wave >= 4) || (ES_ROLES == 5 && wave < 4)) {
        return;                                                          // 4 / 5: the matrix stream / the rest with the SIMD partner absent
    } else if (ES_ROLES == 2 || ((ES_ROLES == 1 || ES_ROLES == 4) && wave < 4)) {   // 1: waves 0 .. 3 the matrix stream, their SIMD partners 4 .. 7 the rest
        for (int k = 0; k < K; ++k) step(k, T_{}, F_{});
    } else {
        for (int k = 0; k < K; ++k) step(k, F_{}, T_{});
    }
}

// A, Bm [n_rows, 128] fp32 (times log2 e when flags & MAGNET_STREAM_L2E); src [n_edges] the CSR's sources; tiles [n_chunks][tpc] int4 and
// chunk_ntiles [n_chunks] from magnet_stream_tiles_plan (n_chunks a multiple of 8); w2_t32 / b2 from packing.pack_edge32t_fp16x2.
// agg [n_rows, 128] receives the mean over the edges of every row that has edges; rows without edges are NOT written (the plan reports
// them: the caller zero-fills agg first).  flags: include/magnet_hip.h (bit 0 log2(e)-scaled inputs, bit 1 three gather sets in flight);
// the development library adds bit 2 (four gather sets) and bit 3 (s_setprio 1 for the younger wave of every SIMD), include/magnet_hip_dev.h.
extern "C" int magnet_mpnn_edge_stream_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles,
                                                  const int32_t* chunk_ntiles, int n_chunks, int tpc, int n_rows, const void* w2_t32,
                                                  const float* b2, float* agg, int flags, void* stream) {
    (void)hipGetLastError();
#ifdef MAGNET_DEV_VARIANTS
    if (flags & ~15) return -2;
#else
    if (flags & ~3) return -2;
#endif
    if (n_chunks <= 0 || n_rows <= 0) return 0;
    if (n_chunks % 8 || tpc <= 0) return -2;
    const dim3 g(n_chunks / 8), b(512);
    const int4* td = reinterpret_cast<const int4*>(tiles);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
    hipStream_t s = (hipStream_t)stream;
    const int depth = (flags & 4) ? 4 : (flags & 2) ? 3 : 2;      // gather sets in flight
    const int young_prio = (flags >> 3) & 1;
#define STREAM_LAUNCH(L2E_, LBD_) hipLaunchKernelGGL((k_mpnn_edge_stream32<L2E_, LBD_>), g, b, 0, s, A, Bm, src, td, chunk_ntiles, tpc, n_chunks, n_rows, ws, b2, agg, young_prio)
    if (flags & 1) {
        if (depth == 2) STREAM_LAUNCH(true, 2);
        else if (depth == 3) STREAM_LAUNCH(true, 3);
#ifdef MAGNET_DEV_VARIANTS
        else STREAM_LAUNCH(true, 4);
#endif
    } else {
        if (depth == 2) STREAM_LAUNCH(false, 2);
        else if (depth == 3) STREAM_LAUNCH(false, 3);
#ifdef MAGNET_DEV_VARIANTS
        else STREAM_LAUNCH(false, 4);
#endif
    }
#undef STREAM_LAUNCH
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the plan (ops.stream_tiles)
// One wave per chunk.  Chunk c = rows [R_c, R_c+1) with R_c = the first row whose edges begin at or after c E / n_chunks.  Fast path:
// lane l takes the tiles l, l + 64, .. at their nominal places (chunk start + 32 k) and looks its row up by bisection; a tile that
// would touch a third row (or has an empty row behind its first one) makes the wave redo the chunk sequentially on lane 0, cutting
// such tiles short.  flags[0]: bit 0 = a chunk needs more than tpc tiles, bit 1 = there are rows without edges (the edge kernel does not
// write them), bit 2 = a row has 65 536 edges or more; flags[1] = the number of tiles of all chunks.
__global__ void __launch_bounds__(256) k_stream_tiles_plan(const int32_t* __restrict__ rowptr, int n_rows, int n_chunks, int tpc,
                                                           int4* __restrict__ tiles, int32_t* __restrict__ chunk_ntiles,
                                                           int32_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= n_chunks) return;
    const long E = rowptr[n_rows];
    auto first_row_from = [&](long target) {              // smallest R in [0, n_rows] with rowptr[R] >= target
        int lo = 0, hi = n_rows;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rowptr[mid] >= target) hi = mid;
            else lo = mid + 1;
        }
        return lo;
    };
    const int R0 = chunk == 0 ? 0 : first_row_from(chunk * E / n_chunks);
    const int R1 = chunk == n_chunks - 1 ? n_rows : first_row_from((chunk + 1) * E / n_chunks);
    const int start = rowptr[R0], end = rowptr[R1];
    int4* const tb = tiles + (size_t)chunk * tpc;
    int bad = 0, fl = 0;
    for (int r = R0 + lane; r < R1; r += 64) {
        const int d = rowptr[r + 1] - rowptr[r];
        if (d == 0) fl |= 2;
        if (d >= 65536) fl |= 4;
    }
    auto emit = [&](int k, int e0, int r, int cnt) {      // the descriptor of a tile that touches rows r and (at most) r + 1
        const int row_end = rowptr[r + 1];
        const int split = min(row_end - e0, cnt);
        const int deg0 = row_end - rowptr[r];
        const int next_end = r + 2 <= n_rows ? rowptr[r + 2] : row_end;
        const int deg1 = next_end - row_end;
        const int ends0 = row_end <= e0 + cnt, ends1 = split < cnt && next_end == e0 + cnt;
        if (k < tpc) tb[k] = make_int4(e0, r, cnt | (split << 8) | (ends0 ? ST_ENDS0 : 0) | (ends1 ? ST_ENDS1 : 0), deg0 | (deg1 << 16));
    };
    const int K_nominal = (end - start + 31) / 32;
    for (int k = lane; k < K_nominal; k += 64) {
        const int e0 = start + 32 * k, cnt = min(32, end - e0);
        int lo = R0, hi = R1;                             // the row of e0: the last r with rowptr[r] <= e0 < rowptr[r + 1]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rowptr[mid + 1] > e0) hi = mid;
            else lo = mid + 1;
        }
        const int r = lo, row_end = rowptr[r + 1];
        if (row_end - e0 < cnt) {                          // a second row: it must reach the end of the tile (or end with it)
            const int next_end = rowptr[r + 2];
            if (next_end == row_end || next_end - e0 < cnt) bad = 1;
        }
        if (!bad) emit(k, e0, r, cnt);
    }
    int K = K_nominal;
    if (__ballot(bad) != 0) {                              // sequential walk with cuts
        K = 0;
        if (lane == 0) {
            int e = start, r = R0;
            while (e < end) {
                while (rowptr[r + 1] <= e) ++r;
                int cnt = min(32, end - e);
                const int row_end = rowptr[r + 1];
                if (row_end - e < cnt) {
                    const int next_end = rowptr[r + 2];
                    if (next_end == row_end) cnt = row_end - e;               // an empty row follows: the tile ends with row r
                    else if (next_end - e < cnt) cnt = next_end - e;          // a third row would begin: the tile ends with row r + 1
                }
                emit(K, e, r, cnt);
                e += cnt;
                ++K;
            }
        }
        K = __shfl(K, 0);
    }
    if (K > tpc) fl |= 1, K = 0;
    for (int o = 32; o; o >>= 1) fl |= __shfl_xor(fl, o);
    if (lane == 0) {
        chunk_ntiles[chunk] = K;
        if (fl) atomicOr(flags, fl);
        if (K) atomicAdd(flags + 1, K);
    }
}

// tiles [n_chunks][tpc] int4, chunk_ntiles [n_chunks], flags [2] (zeroed here): see k_stream_tiles_plan.  n_chunks: a multiple of 8
// (the edge kernel runs 8 chunks per workgroup).
extern "C" int magnet_stream_tiles_plan(const int32_t* rowptr, int n_rows, int n_chunks, int tpc, void* tiles, int32_t* chunk_ntiles,
                                        int32_t* flags, void* stream) {
    (void)hipGetLastError();
    if (n_rows <= 0 || n_chunks <= 0 || n_chunks % 8 || tpc <= 0) return -2;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_stream_tiles_plan, dim3((n_chunks + 3) / 4), dim3(256), 0, s, rowptr, n_rows, n_chunks, tpc,
                       reinterpret_cast<int4*>(tiles), chunk_ntiles, flags);
    return (int)hipGetLastError();
}
