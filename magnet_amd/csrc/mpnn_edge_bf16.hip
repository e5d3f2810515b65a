// Fused MPNN edge kernel on the bf16 matrix cores - generation 6 (see DESIGN.md 3.2 for the measurements
// that led here).
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// * message_net_2 is evaluated with the exact hi/mid/lo bf16 split (common.cuh, "bf16x6"): 192
//   v_mfma_f32_32x32x16_bf16 per 32-edge tile, fp32-class accuracy, weight planes (96 KiB) resident in LDS.
// * ONE wave per SIMD (4 waves per CU, up to 512 registers each).  Measured on gfx950: MFMA and VALU work of
//   two different waves on one SIMD do not overlap, but VALU instructions placed between the bf16 MFMAs of
//   the same wave are hidden.  So all overlap is built inside the wave, as an explicit software pipeline over
//   the 32 MFMA groups (6 MFMAs each) of tile t:
//       - weight fragments of the next group        (LDS reads)
//       - a quarter of the next k-slot's split      (VALU)
//       - the A[i] / Bm[j] rows of tile t+1, one k tile at a time as soon as tile t has consumed the
//         matching registers, and Swish(A + Bm) two k tiles later           (global loads + VALU)
//       - the Swish + segmented DPP scan of two of tile t-1's 64 output registers          (VALU)
//   sched_barriers after every group keep hipcc from undoing the placement.
#include "common.cuh"
#ifndef MAGNET_EDGE16_FENCE
#define MAGNET_EDGE16_FENCE 0
#endif
#include "edge_tile.cuh"

#ifndef EDGE6_CONTIGUOUS
#define EDGE6_CONTIGUOUS 1
#endif

// Swish + segmented inclusive scan of one output register across the 32 edge lanes
__device__ __forceinline__ float scan_reg(float v, const TileMeta& md) { return seg_scan_sum(act_swish(v), md); }

template <int MODE, int ABL>
__global__ void __launch_bounds__(256, 1) k_mpnn_edge6(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                       const int32_t* __restrict__ rowptr, int E,
                                                       const u32x4* __restrict__ w2s /* [32 slots][NP planes][64] */,
                                                       const float* __restrict__ b2, float* __restrict__ agg,
                                                       float* __restrict__ partials) {
    constexpr int NP = Split<MODE>::NP;
    __shared__ __attribute__((aligned(16))) u32x4 wl[32 * NP * 64];          // 96 KiB (bf16x6) / 64 KiB (fp16x3)
    for (int i = threadIdx.x; i < 32 * NP * 64; i += 256) wl[i] = w2s[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int ntiles = (E + 31) / 32;
#if EDGE6_CONTIGUOUS
    // XCD-aware contiguous ranges: workgroups are dispatched round-robin over the 8 XCDs (b % 8), each XCD has its own L2;
    // virtual block vb groups the workgroups of one XCD and every wave walks a contiguous run of tiles, so the A / Bm rows a
    // wave gathers (edges sorted by target = spatially local) stay in its XCD's L2 instead of being fetched by all eight
    const int nb8 = gridDim.x / 8;
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    int tile = (vb * 4 + wave) * tpw;
    const int tend = min(tile + tpw, ntiles);
    constexpr int stride = 1;
#else
    const int stride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    const int tend = ntiles;
#endif
    if (tile >= tend) return;
    f32x16 in[4], out[4], outc[4], prev[4];
    f32x4 la[4], lb[4];                       // one k tile (32 columns) of A[i] / Bm[j] in flight
    {   // prologue: gather + Swish for the first tile
        const int e = min(tile * 32 + j, E - 1);
        const float* pa = A + (size_t)tgt[e] * 128 + 4 * half;
        const float* pb = Bm + (size_t)src[e] * 128 + 4 * half;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pa + kt * 32 + 8 * g);
                const f32x4 b = *reinterpret_cast<const f32x4*>(pb + kt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) in[kt][4 * g + c] = act_swish(a[c] + b[c]);
            }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) prev[nt][i] = 0.f;
    }
    TileMeta pmd = tile_meta(tile, E, tgt, rowptr, j);      // placeholder until a real previous tile exists
    bool have_prev = false;
    for (; tile < tend; tile += stride) {
        const TileMeta md = tile_meta(tile, E, tgt, rowptr, j);
        const int tn = tile + stride < tend ? tile + stride : tile;       // clamped: last iteration prefetches unused data
        const int en = min(tn * 32 + j, E - 1);
        const float* pan = A + (size_t)tgt[en] * 128 + 4 * half;
        const float* pbn = Bm + (size_t)src[en] * 128 + 4 * half;
        init_bias<4>(out, b2, half);
        zero_tiles<4>(outc);
        {
            u32x4 xs[2][NP], wr[2][NP];
            split_slot<MODE>(in[0], 0, xs[0]);
            {
                const u32x4* wp = wl + lane;
#pragma unroll
                for (int q = 0; q < NP; ++q) wr[0][q] = wp[64 * q];
            }
#pragma unroll
            for (int gq = 0; gq < 32; ++gq) {     // gq = slot * 4 + nt, slot = kt * 2 + m
                const int slot = gq >> 2, nt = gq & 3;
                if (gq + 1 < 32) {
                    const int s1 = (gq + 1) >> 2, n1 = (gq + 1) & 3;
                    const u32x4* wp = wl + ((((s1 >> 1) * 4 + n1) * 2 + (s1 & 1)) * NP) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < NP; ++q) wr[(gq + 1) & 1][q] = wp[64 * q];
                }
                if (!(ABL & 1) && slot + 1 < 8) Split<MODE>::quarter(in[(slot + 1) >> 1], (slot + 1) & 1, nt, xs[(slot + 1) & 1]);
                Split<MODE>::mma(wr[gq & 1], xs[slot & 1], out[nt], outc[nt]);
                // epilogue of the previous tile, two registers per group
                if (!(ABL & 2)) {
                    prev[(2 * gq) >> 4][(2 * gq) & 15] = scan_reg(prev[(2 * gq) >> 4][(2 * gq) & 15], pmd);
                    prev[(2 * gq + 1) >> 4][(2 * gq + 1) & 15] = scan_reg(prev[(2 * gq + 1) >> 4][(2 * gq + 1) & 15], pmd);
                }
                // rolling gather prefetch for the next tile (k tile kt of `in` is dead after group 8kt+3)
                if (!(ABL & 4) && (gq & 7) == 4) {
                    const int kt = gq >> 3;
                    if (kt > 0) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int c = 0; c < 4; ++c) in[kt - 1][4 * g + c] = act_swish(la[g][c] + lb[g][c]);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        la[g] = *reinterpret_cast<const f32x4*>(pan + kt * 32 + 8 * g);
                        lb[g] = *reinterpret_cast<const f32x4*>(pbn + kt * 32 + 8 * g);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (have_prev) store_run_results<4, true>(prev, pmd, agg, partials, half);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            Split<MODE>::fold(out[nt], outc[nt]);
            prev[nt] = out[nt];
        }
        pmd = md;
        have_prev = true;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) in[3][4 * g + c] = act_swish(la[g][c] + lb[g][c]);
    }
    // epilogue of the last tile
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[nt][i] = scan_reg(prev[nt][i], pmd);
    store_run_results<4, true>(prev, pmd, agg, partials, half);
}

// ------------------------------------------------------------------------------------------------
// Generation 7: the same pipeline, but the segmented mean no longer runs as a DPP scan over the 64 output registers
// of every lane (5 dpp moves + 5 fmas + hazard nops per register: ~1 250 of the ~4 400 instructions a tile cost, and
// the kernel is instruction-issue bound).  The folded GEMM output of tile t-1 is written once to a per-wave LDS tile
// (32 edges x 128 features, rows padded to 132 floats: the 16-byte writes of 16 consecutive lanes cover all 64 banks)
// and read back TRANSPOSED while tile t runs on the matrix cores: in MFMA group e every lane reads features
// (2 lane, 2 lane + 1) of edge e (one conflict-free ds_read_b64), adds the bias, applies Swish and adds into a
// two-register running sum; run ends are wave-uniform (a ballot of tgt[e] != tgt[e+1]), so the finished row leaves
// with one coalesced 512-byte store under a scalar branch.  Same number of Swish evaluations, ~3x fewer instructions.
// The sum runs in edge order (deterministic); tile-boundary runs go through `partials` exactly as before.
constexpr int TB_STRIDE = 132;

// L2E: A and Bm arrive multiplied by log2(e) (folded into the packed message_net_1 halves by the caller).  Swish then costs
// one multiply less on both sides of the GEMM: Swish(A + Bm) comes out with the factor, W2 passes it on, the bias registers
// are scaled once per kernel, the second Swish keeps it, and the division by the degree divides it out.
template <int MODE, int ABL = 0, bool L2E = false>
__global__ void __launch_bounds__(256, 1) k_mpnn_edge7(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                       const int32_t* __restrict__ rowptr, int E,
                                                       const u32x4* __restrict__ w2s /* [32 slots][NP planes][64] */,
                                                       const float* __restrict__ b2, float* __restrict__ agg,
                                                       float* __restrict__ partials) {
    constexpr int NP = Split<MODE>::NP;
    static_assert(NP <= 2, "the transposition tiles and three weight planes do not fit the 160 KiB of LDS together");
    auto sw = [](float x) { return L2E ? act_swish_l2e(x) : act_swish(x); };
    constexpr float deg_scale = L2E ? MAGNET_LOG2E : 1.0f;           // folded into the divisor of finished rows
    __shared__ __attribute__((aligned(16))) u32x4 wl[32 * NP * 64];          // 64 KiB (fp16x3) / 32 KiB (bf16)
    __shared__ __attribute__((aligned(16))) float tb[4][32 * TB_STRIDE];     // 66 KiB
    for (int i = threadIdx.x; i < 32 * NP * 64; i += 256) wl[i] = w2s[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int ntiles = (E + 31) / 32;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges, see k_mpnn_edge6
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    const int range = vb * 4 + wave;
    int tile = range * tpw;
    const int tend = min(tile + tpw, ntiles);
    if (tile >= tend) return;
    const int e_lo = tile * 32, e_hi = min(tend * 32, E);           // this wave's edges
    float* const tw = tb[wave] + j * TB_STRIDE + 4 * half;          // this lane's row of the tile (write side)
    const float* const tr = tb[wave] + 2 * lane;                    // this lane's feature pair (read side)
    f32x16 in[4], out[4], outc[4], bias[4];
    init_bias<4>(bias, b2, half);             // loop-invariant C operand of every tile's first main product
    if (L2E) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) bias[nt][i] *= MAGNET_LOG2E;
    }
    // Gather staging: two register sets, k tiles alternate between them.  The rows of k tile kt of the NEXT tile are
    // requested in group 8kt and become Swish(A + Bm) twelve groups (~4 000 cycles) later, in groups 8kt+12 .. 8kt+15
    // (k tile 3: groups 4-7 of the next iteration) - `in[kt]` of the current tile is dead after group 8kt+3.  With a
    // single set the distance was four groups and a third of the kernel's cycles were s_waitcnt on these loads
    // (SQ_WAIT_ANY 32 %; the gathered rows take ~2 000 cycles under load).
    f32x4 la[2][4], lb[2][4];
    TileIdx ix = tile_idx(tile, e_hi, src, tgt, j);
    RunMeta md;
#pragma unroll
    for (int c = 0; c < 16; ++c) {            // the first iteration reads the LDS tile before any GEMM has filled it: Swish(0) = 0
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(tw + (c >> 2) * 32 + 8 * (c & 3)) = z;
    }
    {   // prologue: gather + Swish for the first tile
        const float* pa = A + (size_t)ix.ti * 128 + 4 * half;
        const float* pb = Bm + (size_t)ix.sj * 128 + 4 * half;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pa + kt * 32 + 8 * g);
                const f32x4 b = *reinterpret_cast<const f32x4*>(pb + kt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) in[kt][4 * g + c] = sw(a[c] + b[c]);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) {          // k tile 3 stays raw: groups 4-7 of the first iteration turn it into in[3]
            la[1][g] = *reinterpret_cast<const f32x4*>(pa + 96 + 8 * g);
            lb[1][g] = *reinterpret_cast<const f32x4*>(pb + 96 + 8 * g);
        }
        md.ti = ix.ti;
        md.rs = rowptr[ix.ti];
        md.re = rowptr[ix.ti + 1];
        md.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix.e_ok && ix.tnext != ix.ti);   // both halves agree: low word
    }
    RunMeta pmd = md;
    pmd.ends = 0;                              // no previous tile yet: nothing leaves in the first iteration
    TileIdx ix1 = tile_idx(min(tile + 1, tend - 1), e_hi, src, tgt, j);    // indices one tile ahead ...
    // (a use before the loop: otherwise these loads are still pending at the loop header on the entry path, and hipcc's
    // merged wait state puts an s_waitcnt on everything but the newest three loads at the top of EVERY iteration)
    asm volatile("" ::"v"(ix1.ti), "v"(ix1.sj), "v"(ix1.tnext), "v"(md.rs), "v"(md.re));
    f32x2 acc = {0.f, 0.f};                   // running sum of the current run; carried across the wave's tiles
    for (; tile < tend; ++tile) {
        const TileIdx ix2 = tile_idx(min(tile + 2, tend - 1), e_hi, src, tgt, j);   // ... and two: requested here, used next iteration
        RunMeta md1;                                                      // bookkeeping of tile + 1: its targets arrived long ago
        md1.ti = ix1.ti;
        md1.rs = rowptr[ix1.ti];
        md1.re = rowptr[ix1.ti + 1];
        const float* pan = A + (size_t)(ABL & 8 ? j : ix1.ti) * 128 + 4 * half;   // clamped: the last iteration prefetches unused data
        const float* pbn = Bm + (size_t)(ABL & 8 ? j : ix1.sj) * 128 + 4 * half;  // (ABL 8, development: rows that stay in L1)
        f32x2 rv = *reinterpret_cast<const f32x2*>(tr);
        {
            u32x4 xs[2][NP], wr[2][NP];
            split_slot<MODE>(in[0], 0, xs[0]);
            {
                const u32x4* wp = wl + lane;
#pragma unroll
                for (int q = 0; q < NP; ++q) wr[0][q] = wp[64 * q];
            }
#pragma unroll
            for (int gq = 0; gq < 32; ++gq) {     // gq = slot * 4 + nt, slot = kt * 2 + m
                const int slot = gq >> 2, nt = gq & 3;
                if (gq + 1 < 32) {
                    const int s1 = (gq + 1) >> 2, n1 = (gq + 1) & 3;
                    const u32x4* wp = wl + ((((s1 >> 1) * 4 + n1) * 2 + (s1 & 1)) * NP) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < NP; ++q) wr[(gq + 1) & 1][q] = wp[64 * q];
                }
                const f32x2 cur = rv;             // edge gq of the previous tile; the next row is requested now as well
                if (gq + 1 < 32) rv = *reinterpret_cast<const f32x2*>(tr + (gq + 1) * TB_STRIDE);
                __builtin_amdgcn_sched_barrier(0);    // LDS requests first: hipcc otherwise sinks them behind the second MFMA
                                                      // and the next group opens with an exposed LDS round trip
                if (slot + 1 < 8) Split<MODE>::quarter(in[(slot + 1) >> 1], (slot + 1) & 1, nt, xs[(slot + 1) & 1]);
                if (slot == 0) {   // first k slot: the accumulators start from the bias registers / the inline constant 0
                    out[nt] = bias[nt];
#pragma unroll
                    for (int i = 0; i < 16; ++i) outc[nt][i] = 0.f;
                }
                Split<MODE>::mma(wr[gq & 1], xs[slot & 1], out[nt], outc[nt]);
                // edge gq of the previous tile: transposed read, Swish, running sum; a run end leaves here
                {
                    acc[0] += sw(cur[0]);
                    acc[1] += sw(cur[1]);
                    if (pmd.ends & (1u << gq)) {
                        emit_run(acc, pmd, gq, e_lo, e_hi, range, deg_scale, agg, partials, lane);
                        acc[0] = 0.f;
                        acc[1] = 0.f;
                    }
                }
                // staged rows -> Swish(A + Bm), four values per group.  The empty asm pins each value to its group: hipcc
                // otherwise sinks the whole Swish block to the end of the loop.
                if ((gq & 7) >= 4) {
                    const int ktc = ((gq >> 3) + 3) & 3, g = gq & 3;      // groups 4-7: k tile 3, 12-15: 0, 20-23: 1, 28-31: 2
                    const int set = ktc & 1;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = sw(la[set][g][c] + lb[set][g][c]);
                        asm volatile("" : "+v"(v));
                        in[ktc][4 * g + c] = v;
                    }
                }
                if ((gq & 7) == 0) {
                    const int kt = gq >> 3, set = kt & 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        la[set][g] = *reinterpret_cast<const f32x4*>(pan + kt * 32 + 8 * g);
                        lb[set][g] = *reinterpret_cast<const f32x4*>(pbn + kt * 32 + 8 * g);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // this tile's raw GEMM output -> the wave's LDS tile (row = edge), consumed during the next iteration
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            Split<MODE>::fold(out[nt], outc[nt]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {out[nt][4 * g], out[nt][4 * g + 1], out[nt][4 * g + 2], out[nt][4 * g + 3]};
                *reinterpret_cast<f32x4*>(tw + nt * 32 + 8 * g) = o;
            }
        }
        pmd = md;
        md1.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix1.e_ok && ix1.tnext != ix1.ti);
        md = md1;
        ix1 = ix2;
    }
    // the last tile's rows
    for (int e = 0; e < 32; ++e) {
        const f32x2 cur = *reinterpret_cast<const f32x2*>(tr + e * TB_STRIDE);
        acc[0] += sw(cur[0]);
        acc[1] += sw(cur[1]);
        if (pmd.ends & (1u << e)) {
            const int ti = __shfl(pmd.ti, e), rs = __shfl(pmd.rs, e), re = __shfl(pmd.re, e);
            const bool complete = rs >= e_lo && re <= e_hi;
            const float dn = complete ? (float)(re - rs) * deg_scale : 1.f;
            float* dst = complete ? agg + (size_t)ti * 128 : partials + ((size_t)range * 2 + (rs < e_lo ? 0 : 1)) * 128;
            const f32x2 o = {acc[0] / dn, acc[1] / dn};
            *reinterpret_cast<f32x2*>(dst + 2 * lane) = o;
            acc[0] = 0.f;
            acc[1] = 0.f;
        }
    }
}

// Completes what k_mpnn_edge7 left open: rows without edges become 0, and the row that crosses the start of range w (if it
// ends inside range w) is assembled from the partial sums of the ranges it touches, in range order.
__global__ void __launch_bounds__(256) k_edge_range_fixup(const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr,
                                                          int N, int E, int tpw, int tshift /* log2(edges per tile) */,
                                                          int nranges, float deg_scale,
                                                          const float* __restrict__ partials, float* __restrict__ agg) {
    const int r0 = blockIdx.x * 256 + threadIdx.x;
    if (r0 < N && rowptr[r0] == rowptr[r0 + 1]) {
        f32x4* o = reinterpret_cast<f32x4*>(agg + (size_t)r0 * 128);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < 32; ++c) o[c] = z;
    }
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w < 1 || w >= nranges) return;
    const long es = ((long)w * tpw) << tshift;
    if (es >= E) return;
    const int r = tgt[es];
    const int s = rowptr[r], e = rowptr[r + 1];
    if (s >= es) return;                                   // the row starts here: nothing entered from the previous range
    if (((e - 1) >> tshift) / tpw != w) return;            // the row goes on: the range it ends in assembles it
    const int wa = (s >> tshift) / tpw;
    f32x2 acc = *reinterpret_cast<const f32x2*>(partials + ((size_t)wa * 2 + 1) * 128 + 2 * lane);
    for (int t = wa + 1; t <= w; ++t) {
        const f32x2 p = *reinterpret_cast<const f32x2*>(partials + ((size_t)t * 2 + 0) * 128 + 2 * lane);
        acc[0] += p[0];
        acc[1] += p[1];
    }
    const float dn = (float)(e - s) * deg_scale;
    const f32x2 o = {acc[0] / dn, acc[1] / dn};
    *reinterpret_cast<f32x2*>(agg + (size_t)r * 128 + 2 * lane) = o;
}

// ------------------------------------------------------------------------------------------------
// Generation 8: the generation-7 pipeline on 16-EDGE tiles (v_mfma_f32_16x16x32_f16), two waves per SIMD.
// scripts/dev/issue_probe.cpp: a single wave issues one instruction every ~4.5 cycles and nothing of its MFMA / VALU /
// transcendental work overlaps; with two waves on a SIMD the transcendentals of one wave hide completely behind the other
// wave's MFMAs and plain VALU issues at ~2.8 cycles ("3 MFMA + 20 fma + 4 exp + 4 rcp": 111 ns -> 76 ns per group).
// The 32-edge tile needs 440 registers; halving the tile halves every per-tile array (accumulators 64, inputs 32, staging
// 32, fragments 32: ~230 registers), so eight waves fit a CU.  Price: every wave still reads all of W2 from LDS per tile,
// i.e. twice the LDS weight traffic per edge (ds_read_b128 runs at 256 B/clk on gfx950: ~40 % of the LDS pipe).
// Operand layouts (checked by scripts/dev/mfma16_probe.cpp): A = W2 rows, lane l: row 16t + (l & 15), k = 32s + 8(l >> 4) + 0..7;
// B = activations, lane l: edge l & 15, the same k; D: lane l: edge l & 15, rows 16t + 4(l >> 4) + 0..3.
typedef unsigned int u32;
constexpr int T16_STRIDE = 132;
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <bool L2E, bool SINGLE>
__global__ void __launch_bounds__(512) k_mpnn_edge16(const float* __restrict__ A, const float* __restrict__ Bm,
                                                     const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                     const int32_t* __restrict__ rowptr, int E,
                                                     const u32x4* __restrict__ w2t /* [4 s][8 t][2 planes][64] */,
                                                     const float* __restrict__ b2, float* __restrict__ agg,
                                                     float* __restrict__ partials) {
    auto sw = [](float x) { return L2E ? act_swish_l2e(x) : act_swish(x); };
    // SINGLE (round 3): the weight planes arrive pre-scaled by 2^11 (plane 0 = 2^11 hw, plane 1 = 2^11 (w - hw)) and the low piece of
    // an activation stays UNSCALED (l = x - h, an fp16 subnormal for small x - the MFMA keeps them, scripts/dev/edge9_probe.cpp), so
    // all three products of the split carry the factor 2^11 and share ONE accumulator: no cross accumulator (32 registers), no
    // fold; and the two pieces come straight out of the Swish - h = fp16(v r), l = fp16(v r - h), one v_fma_mix each, r = 1 / (1 + 2^-v)
    // - instead of two multiplies and a five-instruction split per value pair.
    constexpr float acc_scale = SINGLE ? 2048.0f : 1.0f;
    auto pieces = [](float v0, float v1, u32& hw, u32& lw) {       // SINGLE: Swish of the pair -> one word of each plane
        const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E));
        const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E));
        u32 h, l;
        asm(MAGNET_TRANS_HEAD "v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(r0));
        asm(MAGNET_TRANS_HEAD "v_fma_mixhi_f16 %0, %1, %2, 0" MAGNET_MIX_TAIL : "+v"(h) : "v"(v1), "v"(r1));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(r0), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" MAGNET_MIX_TAIL : "+v"(l) : "v"(v1), "v"(r1), "v"(h));
        hw = h, lw = l;
    };
    constexpr float deg_scale = L2E ? MAGNET_LOG2E : 1.0f;
    __shared__ __attribute__((aligned(16))) u32x4 wl[4 * 8 * 2 * 64];          // 64 KiB
    __shared__ __attribute__((aligned(16))) float tb[8][16 * T16_STRIDE];      // 66 KiB
    for (int i = threadIdx.x; i < 4 * 8 * 2 * 64; i += 512) wl[i] = w2t[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int ntiles = (E + 15) / 16;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges, see k_mpnn_edge6
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 8 - 1) / (gridDim.x * 8);
    const int range = vb * 8 + wave;
    int tile = range * tpw;
    const int tend = min(tile + tpw, ntiles);
    if (tile >= tend) return;
    const int e_lo = tile * 16, e_hi = min(tend * 16, E);           // this wave's edges
    float* const tw = tb[wave] + j * T16_STRIDE + 4 * q;            // this lane's row of the tile (write side), + 16 t
    const float* const tr = tb[wave] + 2 * lane;                    // this lane's feature pair (read side), + 132 e
    f32x4 out[8], outc[8], bias[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {             // loop-invariant C operand of every tile's first main product
        bias[t] = *reinterpret_cast<const f32x4*>(b2 + 16 * t + 4 * q);
        bias[t] *= (L2E ? MAGNET_LOG2E : 1.0f) * acc_scale;
    }
    float in[4][8];                           // !SINGLE: Swish(A + Bm) of the four k steps, split when the step comes up
    u32 ph[4][4], pl[4][4];                   // SINGLE: the two fp16 planes of the same values (word p = values 2 p, 2 p + 1)
    f32x4 la[2][2], lb[2][2];                 // gather staging: two sets of one k step (8 columns of A[i], 8 of Bm[j])
    TileIdx ix = tile_idx<16>(tile, e_hi, src, tgt, j);
    RunMeta md;
#pragma unroll
    for (int t = 0; t < 8; ++t) {             // the first iteration reads the LDS tile before any GEMM has filled it: Swish(0) = 0
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(tw + 16 * t) = z;
    }
    {   // prologue: gather + Swish for the first tile
        const float* pa = A + (size_t)ix.ti * 128 + 8 * q;
        const float* pb = Bm + (size_t)ix.sj * 128 + 8 * q;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 32 * s + 4 * g);
                const f32x4 b = *reinterpret_cast<const f32x4*>(pb + 32 * s + 4 * g);
                if (SINGLE) {
                    pieces(a[0] + b[0], a[1] + b[1], ph[s][2 * g], pl[s][2 * g]);
                    pieces(a[2] + b[2], a[3] + b[3], ph[s][2 * g + 1], pl[s][2 * g + 1]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) in[s][4 * g + c] = sw(a[c] + b[c]);
                }
            }
#pragma unroll
        for (int g = 0; g < 2; ++g) {          // k step 3 stays raw: groups 4-7 of the first iteration turn it into in[3]
            la[1][g] = *reinterpret_cast<const f32x4*>(pa + 96 + 4 * g);
            lb[1][g] = *reinterpret_cast<const f32x4*>(pb + 96 + 4 * g);
        }
        md.ti = ix.ti;
        md.rs = rowptr[ix.ti];
        md.re = rowptr[ix.ti + 1];
        md.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix.e_ok && ix.tnext != ix.ti) & 0xFFFFu;   // lanes 0-15
    }
    RunMeta pmd = md;
    pmd.ends = 0;                              // no previous tile yet: nothing leaves in the first iteration
    TileIdx ix1 = tile_idx<16>(min(tile + 1, tend - 1), e_hi, src, tgt, j);
    asm volatile("" ::"v"(ix1.ti), "v"(ix1.sj), "v"(ix1.tnext), "v"(md.rs), "v"(md.re));   // see k_mpnn_edge7
    f32x2 acc = {0.f, 0.f};                   // running sum of the current run; carried across the wave's tiles
    for (; tile < tend; ++tile) {
        const TileIdx ix2 = tile_idx<16>(min(tile + 2, tend - 1), e_hi, src, tgt, j);
        RunMeta md1;
        md1.ti = ix1.ti;
        md1.rs = rowptr[ix1.ti];
        md1.re = rowptr[ix1.ti + 1];
        const float* pan = A + (size_t)ix1.ti * 128 + 8 * q;             // clamped: the last iteration prefetches unused data
        const float* pbn = Bm + (size_t)ix1.sj * 128 + 8 * q;
        f32x2 rv = *reinterpret_cast<const f32x2*>(tr);
        {
            u32x4 xs[2][2], wr[2][2][2];          // weight fragments: [buffer][tile of the pair][plane]
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (SINGLE) {
                    xs[0][0][p] = ph[0][p];
                    xs[0][1][p] = pl[0][p];
                } else {
                    const u32x2 w = Split<GEMM_FP16X3>::pair<false>(in[0][2 * p], in[0][2 * p + 1]);
                    xs[0][0][p] = w[0];
                    xs[0][1][p] = w[1];
                }
            }
            // the plane words come out of inline asm, which hipcc's hazard recognizer does not see (common.cuh, MAGNET_MIX_TAIL).  This
            // kernel's software pipeline puts a whole double group (LDS requests, the next pair's split) between the last word of an
            // operand and the first MFMA that reads it, and every build is checked for that on the linked ISA
            // (magnet_amd/hazard_check.py); -DMAGNET_EDGE16_FENCE=1 adds an explicit fence per finished operand instead (four per
            // tile: measured +5 % on the kernel, the volatile asm breaks hipcc's interleaving of the double groups)
#if MAGNET_EDGE16_FENCE
            if (!SINGLE) mfma_operand_fence(xs[0][0], xs[0][1]);
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                wr[0][u][0] = wl[(u * 2 + 0) * 64 + lane];
                wr[0][u][1] = wl[(u * 2 + 1) * 64 + lane];
            }
            // 16 double groups: dg = s * 4 + tp handles k step s for the output tiles 2 tp and 2 tp + 1.  The six MFMAs
            // alternate between the two tiles, so an accumulator is read again only two MFMAs after it was written
            // (back to back, the dependent cross product stalled the issue for an MFMA latency).  One edge of the
            // previous tile per double group.
#pragma unroll
            for (int dg = 0; dg < 16; ++dg) {
                const int s = dg >> 2, tp = dg & 3, t0 = 2 * tp, t1 = 2 * tp + 1;
                if (dg + 1 < 16) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        wr[(dg + 1) & 1][u][0] = wl[(((dg + 1) * 2 + u) * 2 + 0) * 64 + lane];
                        wr[(dg + 1) & 1][u][1] = wl[(((dg + 1) * 2 + u) * 2 + 1) * 64 + lane];
                    }
                }
                const f32x2 cur = rv;             // edge dg of the previous tile; the next row is requested now
                if (dg + 1 < 16) rv = *reinterpret_cast<const f32x2*>(tr + (dg + 1) * T16_STRIDE);
                __builtin_amdgcn_sched_barrier(0);    // LDS requests first (see k_mpnn_edge7)
                if (s + 1 < 4) {
                    if (SINGLE) {
                        xs[(s + 1) & 1][0][tp] = ph[s + 1][tp];
                        xs[(s + 1) & 1][1][tp] = pl[s + 1][tp];
                    } else {
                        const u32x2 w = Split<GEMM_FP16X3>::pair<false>(in[s + 1][2 * tp], in[s + 1][2 * tp + 1]);
                        xs[(s + 1) & 1][0][tp] = w[0];
                        xs[(s + 1) & 1][1][tp] = w[1];
#if MAGNET_EDGE16_FENCE
                        if (tp == 3) mfma_operand_fence(xs[(s + 1) & 1][0], xs[(s + 1) & 1][1]);
#endif
                    }
                }
                if (s == 0) {   // first k step: the accumulators start from the bias registers / the inline constant 0
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    out[t0] = bias[t0], out[t1] = bias[t1];
                    outc[t0] = z, outc[t1] = z;
                }
                const int bsel = dg & 1;
                if (SINGLE) {
                    out[t0] = mfma16(wr[bsel][0][1], xs[s & 1][0], out[t0]);
                    out[t1] = mfma16(wr[bsel][1][1], xs[s & 1][0], out[t1]);
                    out[t0] = mfma16(wr[bsel][0][0], xs[s & 1][0], out[t0]);
                    out[t1] = mfma16(wr[bsel][1][0], xs[s & 1][0], out[t1]);
                    out[t0] = mfma16(wr[bsel][0][0], xs[s & 1][1], out[t0]);
                    out[t1] = mfma16(wr[bsel][1][0], xs[s & 1][1], out[t1]);
                } else {
                    outc[t0] = mfma16(wr[bsel][0][1], xs[s & 1][0], outc[t0]);
                    outc[t1] = mfma16(wr[bsel][1][1], xs[s & 1][0], outc[t1]);
                    out[t0] = mfma16(wr[bsel][0][0], xs[s & 1][0], out[t0]);
                    out[t1] = mfma16(wr[bsel][1][0], xs[s & 1][0], out[t1]);
                    outc[t0] = mfma16(wr[bsel][0][0], xs[s & 1][1], outc[t0]);
                    outc[t1] = mfma16(wr[bsel][1][0], xs[s & 1][1], outc[t1]);
                }
                {                                 // transposed read of one edge of the previous tile, Swish, running sum
                    acc[0] += sw(cur[0]);
                    acc[1] += sw(cur[1]);
                    if (pmd.ends & (1u << dg)) {
                        emit_run(acc, pmd, dg, e_lo, e_hi, range, deg_scale, agg, partials, lane);
                        acc[0] = 0.f;
                        acc[1] = 0.f;
                    }
                }
                // staged rows -> Swish(A + Bm), four values per double group, pinned to it (see k_mpnn_edge7)
                if ((dg & 3) >= 2) {
                    const int sc = ((dg >> 2) + 3) & 3, idx = dg & 1, set = sc & 1;   // dg 2-3: k step 3, 6-7: 0, 10-11: 1, 14-15: 2
                    if (SINGLE) {
                        u32 h0, l0, h1, l1;
                        pieces(la[set][idx][0] + lb[set][idx][0], la[set][idx][1] + lb[set][idx][1], h0, l0);
                        pieces(la[set][idx][2] + lb[set][idx][2], la[set][idx][3] + lb[set][idx][3], h1, l1);
                        asm volatile("" : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1));
                        ph[sc][2 * idx] = h0, pl[sc][2 * idx] = l0, ph[sc][2 * idx + 1] = h1, pl[sc][2 * idx + 1] = l1;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = sw(la[set][idx][c] + lb[set][idx][c]);
                            asm volatile("" : "+v"(v));
                            in[sc][4 * idx + c] = v;
                        }
                    }
                }
                if ((dg & 3) == 0) {
                    const int set = s & 1;
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        la[set][h2] = *reinterpret_cast<const f32x4*>(pan + 32 * s + 4 * h2);
                        lb[set][h2] = *reinterpret_cast<const f32x4*>(pbn + 32 * s + 4 * h2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // this tile's raw GEMM output -> the wave's LDS tile (row = edge), consumed during the next iteration
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = SINGLE ? out[t][r] * (1.0f / 2048.0f) : fmaf(outc[t][r], 1.0f / 2048.0f, out[t][r]);
            *reinterpret_cast<f32x4*>(tw + 16 * t) = o;
        }
        pmd = md;
        md1.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix1.e_ok && ix1.tnext != ix1.ti) & 0xFFFFu;
        md = md1;
        ix1 = ix2;
    }
    // the last tile's rows
    for (int e = 0; e < 16; ++e) {
        const f32x2 cur = *reinterpret_cast<const f32x2*>(tr + e * T16_STRIDE);
        acc[0] += sw(cur[0]);
        acc[1] += sw(cur[1]);
        if (pmd.ends & (1u << e)) {
            const int ti = __shfl(pmd.ti, e), rs = __shfl(pmd.rs, e), re = __shfl(pmd.re, e);
            const bool complete = rs >= e_lo && re <= e_hi;
            const float dn = complete ? (float)(re - rs) * deg_scale : 1.f;
            float* dst = complete ? agg + (size_t)ti * 128 : partials + ((size_t)range * 2 + (rs < e_lo ? 0 : 1)) * 128;
            const f32x2 o = {acc[0] / dn, acc[1] / dn};
            *reinterpret_cast<f32x2*>(dst + 2 * lane) = o;
            acc[0] = 0.f;
            acc[1] = 0.f;
        }
    }
}

// launches k_edge_range_fixup for the edge kernels of other translation units (mpnn_edge32t.hip); C++ linkage, not part of the C ABI
int magnet_launch_edge_range_fixup(const int32_t* tgt, const int32_t* rowptr, int n_rows, int n_edges, int tpw, int tshift,
                                   int nranges, float deg_scale, const float* partials, float* agg, void* stream) {
    const int fb = max((n_rows + 255) / 256, (nranges + 3) / 4);
    if (fb > 0)
        hipLaunchKernelGGL(k_edge_range_fixup, dim3(fb), dim3(256), 0, (hipStream_t)stream, tgt, rowptr, n_rows, n_edges, tpw, tshift,
                           nranges, deg_scale, partials, agg);
    return (int)hipGetLastError();
}

#ifdef MAGNET_DEV_VARIANTS
int g_edge6_ablation = 0;          // development builds only (see mpnn_kernels.hip): 6 = generation-6 kernel, others = ablations
#else
static constexpr int g_edge6_ablation = 0;
#endif

extern "C" int magnet_segmean_fixup_f32(const int32_t* rowptr, int n_rows, int d, const float* partials, float* agg, void* stream);

template <int MODE, bool L2E = false>
static int launch_edge6(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt, const int32_t* rowptr,
                        int n_edges, int n_rows, const void* w2_split, const float* b2, float* agg, float* partials,
                        void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_edges <= 0) return magnet_segmean_fixup_f32(rowptr, n_rows, 128, partials, agg, stream);   // all rows empty: zeros
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const int tiles = (n_edges + 31) / 32;
    const dim3 g(min((tiles + 3) / 4, ncu)), b(256);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_split);
    hipStream_t s = (hipStream_t)stream;
    if constexpr (Split<MODE>::NP <= 2) {
        if (g_edge6_ablation == 0 || g_edge6_ablation == 8) {      // 6 = A/B knob: the generation-6 kernel (DPP scan epilogue)
#ifdef MAGNET_DEV_VARIANTS
            if (g_edge6_ablation == 8)                              // development only (invalid results): gather from 32 fixed rows
                hipLaunchKernelGGL((k_mpnn_edge7<MODE, 8>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
            else
#endif
                hipLaunchKernelGGL((k_mpnn_edge7<MODE, 0, L2E>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
            const int nranges = (int)g.x * 4, tpw = (tiles + nranges - 1) / nranges;      // as computed inside the kernel
            const int fb = max((n_rows + 255) / 256, (nranges + 3) / 4);
            hipLaunchKernelGGL(k_edge_range_fixup, dim3(fb), dim3(256), 0, s, tgt, rowptr, n_rows, n_edges, tpw, 5, nranges,
                               L2E ? MAGNET_LOG2E : 1.0f, partials, agg);
            return (int)hipGetLastError();
        }
    }
    if (L2E) return -2;           // the scaled-input contract exists for the generation-7 kernel only
#ifdef MAGNET_DEV_VARIANTS
    switch (g_edge6_ablation) {   // development only: ablations produce invalid results
        case 1: hipLaunchKernelGGL((k_mpnn_edge6<MODE, 1>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
        case 2: hipLaunchKernelGGL((k_mpnn_edge6<MODE, 2>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
        case 4: hipLaunchKernelGGL((k_mpnn_edge6<MODE, 4>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
        case 7: hipLaunchKernelGGL((k_mpnn_edge6<MODE, 7>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
        default: hipLaunchKernelGGL((k_mpnn_edge6<MODE, 0>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
    }
#else
    hipLaunchKernelGGL((k_mpnn_edge6<MODE, 0>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
#endif
    const int rc = (int)hipGetLastError();
    return rc ? rc : magnet_segmean_fixup_f32(rowptr, n_rows, 128, partials, agg, stream);
}

extern "C" int magnet_mpnn_edge_bf16x6_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                           const int32_t* rowptr, int n_edges, int n_rows, const void* w2_split,
                                           const float* b2, float* agg, float* partials, void* stream) {
    return launch_edge6<GEMM_BF16X6>(A, Bm, src, tgt, rowptr, n_edges, n_rows, w2_split, b2, agg, partials, stream);
}

extern "C" int magnet_mpnn_edge_bf16_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                         const int32_t* rowptr, int n_edges, int n_rows, const void* w2_bf16,
                                         const float* b2, float* agg, float* partials, void* stream) {
    return launch_edge6<GEMM_BF16>(A, Bm, src, tgt, rowptr, n_edges, n_rows, w2_bf16, b2, agg, partials, stream);
}

extern "C" int magnet_mpnn_edge_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                           const int32_t* rowptr, int n_edges, int n_rows, const void* w2_split,
                                           const float* b2, float* agg, float* partials, void* stream) {
    return launch_edge6<GEMM_FP16X3>(A, Bm, src, tgt, rowptr, n_edges, n_rows, w2_split, b2, agg, partials, stream);
}

extern "C" int magnet_mpnn_edge_fp16x3_l2e_f32(const float* A_l2e, const float* Bm_l2e, const int32_t* src, const int32_t* tgt,
                                               const int32_t* rowptr, int n_edges, int n_rows, const void* w2_split,
                                               const float* b2, float* agg, float* partials, void* stream) {
    return launch_edge6<GEMM_FP16X3, true>(A_l2e, Bm_l2e, src, tgt, rowptr, n_edges, n_rows, w2_split, b2, agg, partials, stream);
}

extern "C" int magnet_mpnn_edge_bf16_l2e_f32(const float* A_l2e, const float* Bm_l2e, const int32_t* src, const int32_t* tgt,
                                             const int32_t* rowptr, int n_edges, int n_rows, const void* w2_bf16,
                                             const float* b2, float* agg, float* partials, void* stream) {
    return launch_edge6<GEMM_BF16, true>(A_l2e, Bm_l2e, src, tgt, rowptr, n_edges, n_rows, w2_bf16, b2, agg, partials, stream);
}

// 16-edge tiles, two waves per SIMD (k_mpnn_edge16).  w2_t16: fp16 planes of W2 in the 16x16x32 fragment order
// [4 k steps][8 output tiles][2 planes][64 lanes][8 fp16] (magnet_amd/packing.py:pack_edge16_fp16x2).
extern "C" int magnet_mpnn_edge_fp16x3_t16_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                               const int32_t* rowptr, int n_edges, int n_rows, const void* w2_t16,
                                               const float* b2, float* agg, float* partials, int a_bm_log2e_scaled,
                                               void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_edges <= 0) return magnet_segmean_fixup_f32(rowptr, n_rows, 128, partials, agg, stream);   // all rows empty: zeros
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const int tiles = (n_edges + 15) / 16;
    const dim3 g(min((tiles + 7) / 8, ncu)), b(512);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t16);
    hipStream_t s = (hipStream_t)stream;
    const bool l2e = (a_bm_log2e_scaled & 1) != 0, single = (a_bm_log2e_scaled & 2) != 0;   // bit 1: planes pre-scaled by 2^11 (one accumulator)
    if (single) {
        if (l2e) hipLaunchKernelGGL((k_mpnn_edge16<true, true>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
        else hipLaunchKernelGGL((k_mpnn_edge16<false, true>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
    } else {
        if (l2e) hipLaunchKernelGGL((k_mpnn_edge16<true, false>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
        else hipLaunchKernelGGL((k_mpnn_edge16<false, false>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
    }
    const int nranges = (int)g.x * 8, tpw = (tiles + nranges - 1) / nranges;      // as computed inside the kernel
    const int fb = max((n_rows + 255) / 256, (nranges + 3) / 4);
    hipLaunchKernelGGL(k_edge_range_fixup, dim3(fb), dim3(256), 0, s, tgt, rowptr, n_rows, n_edges, tpw, 4, nranges,
                       l2e ? MAGNET_LOG2E : 1.0f, partials, agg);
    return (int)hipGetLastError();
}
