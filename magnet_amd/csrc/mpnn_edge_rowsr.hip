// Fused MPNN edge kernel, generation 12: row-aligned tiles (generation 11, mpnn_edge_rows.hip) with message_net_2 in REGISTERS.
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// What generation 11's ablations said (profiles/r4/edge_rows.txt): its time is the near-additive sum of the matrix pipe (0.375 ms), the
// LDS weight-fragment reads (0.17 ms: every wave re-reads all 64 KiB of W2's planes per 32-edge tile - 64 ds_read_b128 per lane, bound
// by the return path into the VGPRs), the gathers (0.13) and the two Swish.  The second item is not work, it is where the weights live.
// Here a wave keeps BOTH fp16 planes of the whole 128 x 128 matrix in the accumulation half of its register file - a[0:255], 8 k steps x
// 4 output blocks x 2 planes x 4 registers - read ONCE from the pack, and the MFMAs take them from there as their B operand (gfx950:
// SrcA / SrcB may be AGPRs).  Price: 512 registers per lane, i.e. ONE wave per SIMD (4-wave workgroups, one per CU) instead of two;
// the 32x32x16 shape never profited from the second wave (profiles/r2/overlap_probe.txt: the partner's instructions queue, they do
// not hide).  No LDS besides the parked A row (2 KiB per workgroup).
//
// hipcc does not put a value into the accumulation file because an MFMA could read it there (it uses a[] as spill space and copies back:
// 84 v_accvgpr_read per tile in a probe); an inline-asm MFMA with an "a" operand does.  An asm MFMA is opaque to hipcc's hazard
// recognizer: the wait states between the last MFMA of a tile and the first VALU read of its accumulators are written out below
// (results_ready), and magnet_amd/hazard_check.py checks every path of the linked code object at build time.
//
// Everything else is generation 11: one tile = up to 32 edges of ONE CSR row (ops.row_tiles), transposed GEMM D = X W2^T (lane = output
// feature, the mean is a sum over 16 accumulator registers + one v_permlane32_swap), A[i] once per tile through LDS, one accumulator
// for the three products of the fp16x3 split (planes pre-scaled by 2^11: packing.pack_edge32t_fp16x2), the rows' 33rd edges as S tiles
// at the head of the wave's pipeline.
#include "common.cuh"
#include <type_traits>

typedef unsigned int u32;

namespace {

constexpr float R_SCALE = 2048.0f;         // every MFMA product carries 2^11 (host-scaled weight planes)
constexpr int S_FLAG = 1 << 30;            // descriptor word z of a straggler tile (mpnn_edge_rows.hip)

// D (+)= X W^T: A operand = activations (row = edge = lane & 31, arch VGPRs), B operand = weights (column = feature = lane & 31, AGPRs)
__device__ __forceinline__ void mfma_xw_acc(f32x16& c, const u32x4& x, const u32x4& w) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "a"(w));
}
__device__ __forceinline__ void mfma_xw_zero(f32x16& c, const u32x4& x, const u32x4& w) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(x), "a"(w));
}
// every MFMA above is ordered before, every read of the four accumulators after this statement; an 8-pass MFMA's result may be read by a
// VALU instruction passes + 2 (+ 1 on gfx950) wait states after it issued
__device__ __forceinline__ void results_ready(f32x16 (&o)[4]) {
    asm volatile("s_nop 11" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
}

// Swish of the pre-activation pair (v0, v1) -> one word of the high plane and one of the low plane (mpnn_edge_rows.hip)
template <bool L2E>
__device__ __forceinline__ void swish_pieces(float v0, float v1, u32& hw, u32& lw) {
    const float e0 = __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E), e1 = __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E);
    const float r0 = __builtin_amdgcn_rcpf(1.0f + e0), r1 = __builtin_amdgcn_rcpf(1.0f + e1);
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

// upper 32 lanes of a <-> lower 32 lanes of b (wait states on both sides: the operands come out of VALU instructions and go into them)
__device__ __forceinline__ void half_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

}  // namespace

#ifndef ROWSR_DEPTH
#define ROWSR_DEPTH 2                       // k steps of gathered Bm columns in flight (register sets)
#endif

template <bool L2E>
__global__ void __launch_bounds__(256) k_mpnn_edge_rowsr(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const int32_t* __restrict__ csr_src, const int4* __restrict__ tiles, int ntiles,
                                                         const int32_t* __restrict__ strag_ptr /* [rows + 1] or nullptr */,
                                                         const int32_t* __restrict__ strag_row, const int32_t* __restrict__ strag_src,
                                                         const u32x4* __restrict__ w2p /* [8 ks][4 nt][2 planes][64] */,
                                                         const float* __restrict__ b2, float* __restrict__ agg) {
    constexpr float l2e = L2E ? MAGNET_LOG2E : 1.0f;                       // the Swish outputs carry log2 e when the inputs do
    constexpr int D = ROWSR_DEPTH;
    __shared__ __attribute__((aligned(16))) float arow[4][128];            // the A row of the row tile whose inputs are in production
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges (k_mpnn_edge6): consecutive tiles = neighbouring targets
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    const int t_lo = (vb * 4 + wave) * tpw, t_hi = min(t_lo + tpw, ntiles);
    if (t_lo >= t_hi) return;              // (no workgroup barrier anywhere in this kernel)
    // ---- both planes of W2, resident in a[0:255]: fragment (ks, nt, plane) of this lane
    u32x4 wq[8][4][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) wq[ks][nt][p] = w2p[((ks * 4 + nt) * 2 + p) * 64 + lane];
    const int n_r = t_hi - t_lo;
    int s_lo = 0, n_s = 0;                 // this wave's stragglers: those of the rows of its tiles
    if (strag_ptr != nullptr) {
        s_lo = strag_ptr[tiles[t_lo].y];
        n_s = strag_ptr[tiles[t_hi - 1].y + 1] - s_lo;
    }
    const int ks_tiles = (n_s + 31) / 32;  // S tiles, then n_r row tiles
    const int K = ks_tiles + n_r;          // elements of this wave's sequence
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
        return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
    };
    // four values of a tile's input: columns 16 ks + 8 half + 4 part + 0..3  ->  words 2 part, 2 part + 1 of both planes
    auto pieces4 = [&](int ks, int part, const f32x4& a, const f32x4& b) {
        u32 h0, l0, h1, l1;
        swish_pieces<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
        swish_pieces<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
        xh[ks][2 * part] = h0, xh[ks][2 * part + 1] = h1;
        xl[ks][2 * part] = l0, xl[ks][2 * part + 1] = l1;
    };

    // ---- one iteration: GEMM + epilogue of the current tile (type CUR_S), input production of the next one (type NXT_S),
    // index / A-row prefetch of the one after (type known at run time)
    auto step = [&](auto cur_tag, auto nxt_tag, int k) __attribute__((always_inline)) {
        constexpr bool CUR_S = decltype(cur_tag)::value, NXT_S = decltype(nxt_tag)::value;
        f32x4 lb[D][2];                        // gathered Bm columns 16 ks + 8 half + 0..7 of one k step per set
        f32x4 lag[D][2];                       // NXT_S: the A columns likewise
        auto request = [&](int set, int ks, const float* pa, const float* pb) {
            lb[set][0] = *reinterpret_cast<const f32x4*>(pb + 16 * ks);
            lb[set][1] = *reinterpret_cast<const f32x4*>(pb + 16 * ks + 4);
            if (NXT_S) {
                lag[set][0] = *reinterpret_cast<const f32x4*>(pa + 16 * ks);
                lag[set][1] = *reinterpret_cast<const f32x4*>(pa + 16 * ks + 4);
            }
        };
        auto produce = [&](int set, int ks, int part, const f32x4& a_lds) {
            const f32x4 a = NXT_S ? lag[set][part] : a_lds;
            pieces4(ks, part, a, lb[set][part]);
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
        f32x4 la = {0.f, 0.f, 0.f, 0.f};      // A columns of the four values the next double group produces
        // the first D - 1 k steps of the next tile's Bm columns are requested up front (the registers are free: the previous iteration's
        // production is over), k step ks + D - 1 in double group 2 ks
#pragma unroll
        for (int ks = 0; ks < D - 1; ++ks) request(ks % D, ks, pa, pb);
        // ---- GEMM phase: double group dg = 2 ks + q handles k step ks for the output blocks 2 q and 2 q + 1; the six MFMAs alternate
        // between the two accumulators.  Beside them: four input values of the next tile (k step (dg - 2) / 2 - its registers were last
        // read by double group dg - 1), the gather request of a later k step.
#pragma unroll
        for (int dg = 0; dg < 16; ++dg) {
            const int ks = dg >> 1, q = dg & 1, n0 = 2 * q, n1 = 2 * q + 1;
            const f32x4 la_now = la;
            if (dg >= 1) la = lds_a(16 * ((dg - 1) >> 1) + 4 * ((dg - 1) & 1));   // for double group dg + 1
            __builtin_amdgcn_sched_barrier(0);    // LDS requests first
            if (ks == 0) {
                mfma_xw_zero(out[n0], xh[0], wq[0][n0][1]);
                mfma_xw_zero(out[n1], xh[0], wq[0][n1][1]);
            } else {
                mfma_xw_acc(out[n0], xh[ks], wq[ks][n0][1]);
                mfma_xw_acc(out[n1], xh[ks], wq[ks][n1][1]);
            }
            mfma_xw_acc(out[n0], xl[ks], wq[ks][n0][0]);
            mfma_xw_acc(out[n1], xl[ks], wq[ks][n1][0]);
            mfma_xw_acc(out[n0], xh[ks], wq[ks][n0][0]);
            mfma_xw_acc(out[n1], xh[ks], wq[ks][n1][0]);
            if (dg >= 2) produce(((dg - 2) >> 1) % D, (dg - 2) >> 1, dg & 1, la_now);
            // set (ks + D - 1) % D: k step ks - 1's, drained by double groups 2 ks, 2 ks + 1 - requested behind this one's production
            if (q == 1 && ks + D - 1 < 8) request((ks + D - 1) % D, ks + D - 1, pa, pb);
            if (dg == 6) edge_of(d2, s2, idx2, pa2);        // the tile after next (d2 has arrived by now)
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue phase: k step 7 of the next tile's inputs, then Swish of this tile's 64 outputs per lane
        produce(7 % D, 7, 0, la);
        produce(7 % D, 7, 1, lds_a(16 * 7 + 4));
        results_ready(out);
        if (CUR_S) {                           // every edge's message to its own row; register i <-> edge 8 (i / 4) + 4 half + i % 4
            const int lim = cnt - 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int erow[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) erow[c] = strag_row[d0.x + max(min(8 * g + 4 * half + c, cnt - 1), 0)];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float* dst = agg + (size_t)erow[c] * 128 + j;
                    float m[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) m[nt] = swish2(out[nt][4 * g + c], bl[nt]) * (1.0f / l2e);
                    if (8 * g + c < lim) {
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
                        acc[nt] = fmaf(y, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E)), acc[nt]);
                    }
            } else {                           // a short row: register i of lane half h is edge 8 (i / 4) + 4 h + i % 4
                const int lim = cnt - 4 * half;
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        float m = swish2(out[nt][i], bl[nt]);
                        asm("" : "+v"(m));                  // (computed for every lane, then selected: left alone hipcc branches around each
                        acc[nt] += (8 * (i >> 2) + (i & 3) < lim) ? m : 0.f;      //  of the 64 values and spills the partial sums)
                    }
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
                for (int part = 0; part < 2; ++part) pieces4(4 * hk + q, part, ga[q][part], gb[q][part]);
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

// ------------------------------------------------------------------------------------------------------------------------------------
// Generation 12b: the same kernel with the stragglers GROUPED WITH THEIR ROWS.  Generations 11 / 12 send all of a wave's 33rd edges
// through the pipeline first (8 S tiles for 256 rows): their 32 A rows and 32 Bm rows are first touches, their messages leave as 64
// scattered 128-byte stores per tile, a vmcnt(0) drain separates them from the row tiles and every row tile reads its message back
// from HBM - 3 % of the edges cost 9 % of the time (profiles/r4/edge_rows.txt).  Here a wave's sequence is
//     S_0  R_0 .. R_31   S_1  R_32 .. R_63   ...
// S_g = the 33rd edges of the NEXT 32 rows (lane e <-> row tile 32 g + e: no straggler lists, the row tiles' descriptors say it all).
// Its A rows are the rows the next 32 tiles use, its messages are parked in a wave-private LDS block (17 KiB per wave - LDS is free
// since W2 moved into the registers) and the row tile of row 32 g + e picks its own up from there: no scattered stores, no drain, no
// read-back.  Rows with MORE than one edge beyond their tile (degree > 33: no radius graph of the reference has them) keep the
// generation-11 contract: the mean over those edges is in agg[row] on entry.
constexpr int PARK_ST = 136;               // floats per parked message: 4 rows apart = 32 banks apart (the two lane halves of a store)

template <bool L2E, bool STRAG>
__global__ void __launch_bounds__(256) k_mpnn_edge_rowsg(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const int32_t* __restrict__ csr_src, const int4* __restrict__ tiles, int ntiles,
                                                         const u32x4* __restrict__ w2p /* [8 ks][4 nt][2 planes][64] */,
                                                         const float* __restrict__ b2, float* __restrict__ agg) {
    constexpr float l2e = L2E ? MAGNET_LOG2E : 1.0f;
    constexpr int D = ROWSR_DEPTH;
    __shared__ __attribute__((aligned(16))) float arow[4][128];            // the A row of the row tile whose inputs are in production
    __shared__ __attribute__((aligned(16))) float parks[STRAG ? 4 : 1][STRAG ? 32 * PARK_ST : 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb8 = gridDim.x / 8;
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    const int t_lo = (vb * 4 + wave) * tpw, t_hi = min(t_lo + tpw, ntiles);
    if (t_lo >= t_hi) return;              // (no workgroup barrier anywhere in this kernel)
    u32x4 wq[8][4][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) wq[ks][nt][p] = w2p[((ks * 4 + nt) * 2 + p) * 64 + lane];
    const int n_r = t_hi - t_lo;
    const int G = STRAG ? (n_r + 31) / 32 : 0;         // groups of 32 row tiles, each headed by its S tile
    const int K = n_r + G;                             // elements of this wave's sequence; the last one is a row tile
    float* const ar = arow[wave];
    const float* const arl = ar + 8 * half;
    float* const park = parks[STRAG ? wave : 0];
    float bl[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl[nt] = b2[32 * nt + j] * l2e;

    // element k: STRAG: group k / 33, position k % 33 (0 = the S tile, p > 0 = row tile 32 g + p - 1); else row tile k
    // (plain ternaries: std::min / max hand back REFERENCES, and a phi of two stack addresses kept the pipeline state in scratch)
    auto is_s = [=](int k) { return STRAG && (k < K - 1 ? k : K - 1) % 33 == 0; };
    auto tile_of = [=](int k) {                        // row tile index (inside the wave) of element k; for an S tile: its first row tile
        k = k < K - 1 ? k : K - 1;
        const int p = k % 33 - 1;
        return STRAG ? 32 * (k / 33) + (p > 0 ? p : 0) : k;
    };
    // per lane: the source index of the lane's edge and the pointer to its A data.  R: edge j of the tile, the tile's row (8 bytes per
    // lane, parked in LDS by the next iteration).  S: lane e <-> row tile first + e: its 33rd edge when it has exactly one edge beyond
    // the tile (any edge of the row otherwise: the lane's message is then never picked up), columns 8 half .. of ITS row
    auto edge_of = [=](int k, bool s, int& idx, const float*& pa) {     // (captures by VALUE: see is_s)
        const int t = tile_of(k);
        if (s) {
            const int tl = t + j < n_r - 1 ? t + j : n_r - 1;
            const int4 dl = tiles[t_lo + tl];
            const bool has = ((dl.z >> 8) == 1) && t + j < n_r;
            idx = csr_src[dl.x + (has ? 32 : 0)];
            pa = A + (size_t)dl.y * 128 + 8 * half;
        } else {
            const int4 d = tiles[t_lo + t];
            const int c1 = (d.z & 0xFF) - 1, jj = j < c1 ? j : c1;
            idx = csr_src[d.x + (jj > 0 ? jj : 0)];
            pa = A + (size_t)d.y * 128 + 2 * lane;
        }
    };

    u32x4 xh[8], xl[8];
    f32x16 out[4];
    int idx1;
    const float* pa1;
    int4 d0;                               // descriptor of the current element when it is a row tile

    auto swish2 = [&](float o, float bias) {
        const float y = fmaf(o, 1.0f / R_SCALE, bias);
        return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
    };
    auto pieces4 = [&](int ks, int part, const f32x4& a, const f32x4& b) {
        u32 h0, l0, h1, l1;
        swish_pieces<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
        swish_pieces<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
        xh[ks][2 * part] = h0, xh[ks][2 * part + 1] = h1;
        xl[ks][2 * part] = l0, xl[ks][2 * part + 1] = l1;
    };

    auto step = [&](auto cur_tag, auto nxt_tag, int k) __attribute__((always_inline)) {
        constexpr bool CUR_S = decltype(cur_tag)::value, NXT_S = decltype(nxt_tag)::value;
        f32x4 lb[D][2];
        f32x4 lag[D][2];
        auto request = [&](int set, int ks, const float* pa, const float* pb) {
            lb[set][0] = *reinterpret_cast<const f32x4*>(pb + 16 * ks);
            lb[set][1] = *reinterpret_cast<const f32x4*>(pb + 16 * ks + 4);
            if (NXT_S) {
                lag[set][0] = *reinterpret_cast<const f32x4*>(pa + 16 * ks);
                lag[set][1] = *reinterpret_cast<const f32x4*>(pa + 16 * ks + 4);
            }
        };
        auto produce = [&](int set, int ks, int part, const f32x4& a_lds) {
            const f32x4 a = NXT_S ? lag[set][part] : a_lds;
            pieces4(ks, part, a, lb[set][part]);
        };
        auto lds_a = [&](int off) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            return NXT_S ? z : *reinterpret_cast<const f32x4*>(arl + off);
        };
        if (!NXT_S) *reinterpret_cast<f32x2*>(ar + 2 * lane) = *reinterpret_cast<const f32x2*>(pa1);
        const float* pb = Bm + (size_t)idx1 * 128 + 8 * half;
        const float* pa = pa1;
        const int4 dn = tiles[t_lo + tile_of(k + 1)];  // the next element's descriptor, should it be a row tile (wave-uniform load)
        const int cnt = d0.z & 0xFF, rem = d0.z >> 8;
        const int pos = STRAG ? (k % 33) - 1 : 0;      // CUR row tile: its place in the group = the lane of S_g that carried its edge
        float old0 = 0.f, old1 = 0.f;
        if (!CUR_S && rem > (STRAG ? 1 : 0)) old0 = agg[(size_t)d0.y * 128 + lane], old1 = agg[(size_t)d0.y * 128 + 64 + lane];
        int idx2 = 0;
        const float* pa2 = pa1;
        f32x4 la = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < D - 1; ++ks) request(ks % D, ks, pa, pb);
#pragma unroll
        for (int dg = 0; dg < 16; ++dg) {
            const int ks = dg >> 1, q = dg & 1, n0 = 2 * q, n1 = 2 * q + 1;
            const f32x4 la_now = la;
            if (dg >= 1) la = lds_a(16 * ((dg - 1) >> 1) + 4 * ((dg - 1) & 1));
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                mfma_xw_zero(out[n0], xh[0], wq[0][n0][1]);
                mfma_xw_zero(out[n1], xh[0], wq[0][n1][1]);
            } else {
                mfma_xw_acc(out[n0], xh[ks], wq[ks][n0][1]);
                mfma_xw_acc(out[n1], xh[ks], wq[ks][n1][1]);
            }
            mfma_xw_acc(out[n0], xl[ks], wq[ks][n0][0]);
            mfma_xw_acc(out[n1], xl[ks], wq[ks][n1][0]);
            mfma_xw_acc(out[n0], xh[ks], wq[ks][n0][0]);
            mfma_xw_acc(out[n1], xh[ks], wq[ks][n1][0]);
            if (dg >= 2) produce(((dg - 2) >> 1) % D, (dg - 2) >> 1, dg & 1, la_now);
            if (q == 1 && ks + D - 1 < 8) request((ks + D - 1) % D, ks + D - 1, pa, pb);
            if (dg == 6) edge_of(k + 2, is_s(k + 2), idx2, pa2);
            __builtin_amdgcn_sched_barrier(0);
        }
        produce(7 % D, 7, 0, la);
        produce(7 % D, 7, 1, lds_a(16 * 7 + 4));
        results_ready(out);
        if (CUR_S) {                           // every edge's message -> the wave's park; register i <-> edge 8 (i / 4) + 4 half + i % 4
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float* dst = park + (8 * (i >> 2) + 4 * half + (i & 3)) * PARK_ST + j;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) dst[32 * nt] = swish2(out[nt][i], bl[nt]) * (1.0f / l2e);
            }
        } else {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (cnt == 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float y = fmaf(out[nt][i], 1.0f / R_SCALE, bl[nt]);
                        acc[nt] = fmaf(y, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E)), acc[nt]);
                    }
            } else {
                const int lim = cnt - 4 * half;
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        float m = swish2(out[nt][i], bl[nt]);
                        asm("" : "+v"(m));                  // (computed for every lane, then selected: left alone hipcc branches around each
                        acc[nt] += (8 * (i >> 2) + (i & 3) < lim) ? m : 0.f;      //  of the 64 values and spills the partial sums)
                    }
            }
            half_swap(acc[0], acc[1]);
            half_swap(acc[2], acc[3]);
            if (STRAG && rem == 1) old0 = park[pos * PARK_ST + lane], old1 = park[pos * PARK_ST + 64 + lane];
            const float invd = __builtin_amdgcn_rcpf((float)d0.w);
            const float wsum = invd * (1.0f / l2e), wold = (float)rem * invd;
            agg[(size_t)d0.y * 128 + lane] = fmaf(acc[0] + acc[1], wsum, old0 * wold);
            agg[(size_t)d0.y * 128 + 64 + lane] = fmaf(acc[2] + acc[3], wsum, old1 * wold);
        }
        d0 = dn, idx1 = idx2, pa1 = pa2;
    };

    // ---- prologue: the first element's inputs
    {
        int i0;
        const float* pa0;
        const bool s0 = is_s(0);
        edge_of(0, s0, i0, pa0);
        d0 = tiles[t_lo];
        const float* pb = Bm + (size_t)i0 * 128 + 8 * half;
        if (!s0) *reinterpret_cast<f32x2*>(ar + 2 * lane) = *reinterpret_cast<const f32x2*>(pa0);
#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {
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
                for (int part = 0; part < 2; ++part) pieces4(4 * hk + q, part, ga[q][part], gb[q][part]);
        }
    }
    edge_of(1, is_s(1), idx1, pa1);

    // one flat loop (the pipeline state stays in registers: with the three loops nested hipcc kept it in scratch); the element types are
    // wave-uniform: S_g is followed by its first row tile, a group's last row tile by S_(g + 1)
    for (int k = 0; k < K; ++k) {
        if (STRAG && k % 33 == 0) step(std::true_type{}, std::false_type{}, k);
        else if (STRAG && k + 1 < K && (k + 1) % 33 == 0) step(std::false_type{}, std::true_type{}, k);
        else step(std::false_type{}, std::false_type{}, k);
    }
}

// Same contract as magnet_mpnn_edge_rows_fp16x3_f32 (mpnn_edge_rows.hip): A, Bm [n_rows, 128] fp32 (times log2 e when
// a_bm_log2e_scaled & 1); src [n_edges] the CSR's sources; tiles / strag_* from magnet_row_tiles_fill; w2_t32 / b2 from
// packing.pack_edge32t_fp16x2; agg rows with edges outside their tile must hold the mean over those on entry.
extern "C" int magnet_mpnn_edge_rowsr_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
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
    const dim3 g(min((n_tiles + 3) / 4, ncu)), b(256);
    const int4* td = reinterpret_cast<const int4*>(tiles);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
    hipStream_t s = (hipStream_t)stream;
    if (a_bm_log2e_scaled & 1)
        hipLaunchKernelGGL((k_mpnn_edge_rowsr<true>), g, b, 0, s, A, Bm, src, td, n_tiles, strag_ptr, strag_row, strag_src, ws, b2, agg);
    else
        hipLaunchKernelGGL((k_mpnn_edge_rowsr<false>), g, b, 0, s, A, Bm, src, td, n_tiles, strag_ptr, strag_row, strag_src, ws, b2, agg);
    return (int)hipGetLastError();
}

// Generation 12b (k_mpnn_edge_rowsg): the stragglers grouped with their rows - no straggler lists.  has_stragglers = 0: every tile's
// rows are complete without one (the S tiles are left out).  Rows with two or more edges beyond their tile: the mean over those in
// agg[row] on entry, as above.
extern "C" int magnet_mpnn_edge_rowsg_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                                 int has_stragglers, const void* w2_t32, const float* b2, float* agg,
                                                 int a_bm_log2e_scaled, void* stream) {
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
    const dim3 g(min((n_tiles + 3) / 4, ncu)), b(256);
    const int4* td = reinterpret_cast<const int4*>(tiles);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
    hipStream_t s = (hipStream_t)stream;
    const bool l2e = (a_bm_log2e_scaled & 1) != 0;
    if (has_stragglers) {
        if (l2e) hipLaunchKernelGGL((k_mpnn_edge_rowsg<true, true>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
        else hipLaunchKernelGGL((k_mpnn_edge_rowsg<false, true>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
    } else {
        if (l2e) hipLaunchKernelGGL((k_mpnn_edge_rowsg<true, false>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
        else hipLaunchKernelGGL((k_mpnn_edge_rowsg<false, false>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
    }
    return (int)hipGetLastError();
}
