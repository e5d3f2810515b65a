// Fused MPNN edge kernel, generation 9: the GEMM TRANSPOSED, D = X . W2^T with the 32 edges of a tile as the ROWS of the
// 32x32x16 MFMA and 32 output features as its columns.
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// Generations 6-8 (mpnn_edge_bf16.hip) put the features in the rows: a lane then owns one EDGE, and the per-target mean - a
// reduction over edges, i.e. over lanes - needed a DPP scan (gen. 6: ~1 250 instructions per tile) or a round trip through a
// per-wave LDS transposition tile (gen. 7 / 8: 66 KiB of LDS, 16 wide LDS writes + 32 reads per 32 edges, and the reason a
// 32-edge tile could not run beside a second wave).  Here a lane owns one output FEATURE and its accumulator registers are the
// tile's edges, so the mean is a plain running sum over registers.  What this removes / changes per 32-edge tile and wave:
//
// * no LDS transposition: after `v_permlane32_swap` of the accumulators of two 32-feature blocks every register holds ONE
//   edge for 64 features (lane = feature), run ends are wave-uniform (scalar branch) and a finished row leaves with one
//   coalesced 256-byte store per 64 features; the sum is the plain edge-order sum (deterministic, same order as gen. 7 / 8);
// * ONE accumulator per output block: the fp16x3 split x.w ~ h.hw + h.lw' + l.hw is summed in a single fp32 accumulator by
//   pre-scaling the weight planes on the host (Hw = 2^11 hw, Lw = 2^11 (w - hw)) and keeping the low piece of the activation
//   UNSCALED (l = x - h, an fp16 subnormal below |x| ~ 0.12 - the MFMA keeps fp16 subnormals, scripts/dev/edge9_probe.cpp);
//   every product then carries the factor 2^11, which the epilogue's bias fma divides out.  No cross accumulator (64
//   registers), no fold (64 fma), and the accumulators start from the inline constant 0;
// * the bias is per LANE (feature), not per register: 2 registers instead of 64, added by the fma that removes the 2^11;
// * the activation pieces come straight out of the Swish: h = fp16(v * r) and l = fp16(v * r - h) are one
//   v_fma_mixlo/hi_f16 each (r = 1 / (1 + 2^-v)): 4 instructions per value pair instead of 2 multiplies + a 5-instruction split;
// * half the LDS weight traffic per edge of the 16-edge kernel (64 ds_read_b128 per 32 edges).
//
// One wave per SIMD (256-thread workgroups, one per CU, persistent over an XCD-aware contiguous tile range as in gen. 6-8);
// everything a tile needs is software-pipelined inside the wave around its 96 MFMAs (16 double groups of 6):
//   - weight fragments of the next double group                                   (4 ds_read_b128)
//   - two edges of the PREVIOUS tile's output: bias fma, Swish, running sum, run-end store     (epilogue)
//   - four values of the NEXT tile's input: A[i] + Bm[j], Swish, the two fp16 pieces            (prologue of tile t+1)
//   - the gathered A / Bm rows of one k slot, requested 6 double groups (~2 500 cycles) before they are consumed,
//     through a ring of four register sets; tile indices two tiles ahead.
#include "common.cuh"
#include "edge_tile.cuh"
#include <type_traits>

typedef unsigned int u32;

namespace {

constexpr float E9_SCALE = 2048.0f;        // every MFMA product carries 2^11 (host-scaled weight planes)
#ifndef E9_VALU_PER_MFMA
#define E9_VALU_PER_MFMA 8               // VALU instructions hipcc places behind every MFMA of a double group (sched_group_barrier)
#endif

__device__ __forceinline__ f32x16 mfma_xw(const u32x4& x, const u32x4& w, const f32x16& c) {
    // A operand = activations (row = edge = lane & 31), B operand = weights (column = feature = lane & 31)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, w), c, 0, 0, 0);
}

// Swish of the pre-activation pair (v0, v1) -> one word of the high plane and one of the low plane:
//   h = fp16(v r), l = fp16(v r - h), r = 1 / (1 + 2^-v')   (v' = v when the inputs carry log2 e, else v log2 e)
template <bool L2E>
__device__ __forceinline__ void swish_split_pair(float v0, float v1, u32& hw, u32& lw) {
    const float e0 = __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E), e1 = __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E);
    const float r0 = __builtin_amdgcn_rcpf(1.0f + e0), r1 = __builtin_amdgcn_rcpf(1.0f + e1);
    u32 h, l;                                  // mixlo writes the low half, mixhi the high half: no initialisation needed
    asm(MAGNET_TRANS_HEAD "v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(r0));
    asm(MAGNET_TRANS_HEAD "v_fma_mixhi_f16 %0, %1, %2, 0" MAGNET_MIX_TAIL : "+v"(h) : "v"(v1), "v"(r1));
    // remainder against the fp16 piece read in place: low half for value 0, high half for value 1
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(r0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" MAGNET_MIX_TAIL : "+v"(l) : "v"(v1), "v"(r1), "v"(h));
    hw = h;
    lw = l;
}

// the two fp16 pieces of v0 r0 and v1 r1 (stage 3 of the pipelined Swish: r = 1 / (1 + 2^-v) was computed a double group earlier)
__device__ __forceinline__ void split_pair_vr(float v0, float r0, float v1, float r1, u32& hw, u32& lw) {
    u32 h, l;
    asm(MAGNET_TRANS_HEAD "v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(r0));
    asm(MAGNET_TRANS_HEAD "v_fma_mixhi_f16 %0, %1, %2, 0" MAGNET_MIX_TAIL : "+v"(h) : "v"(v1), "v"(r1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(r0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" MAGNET_MIX_TAIL : "+v"(l) : "v"(v1), "v"(r1), "v"(h));
    hw = h;
    lw = l;
}

// upper 32 lanes of a <-> lower 32 lanes of b: (a, b) = one accumulator register of two 32-feature blocks -> first = the register's
// edge of lane half 0 for 64 features (lane = feature), second = the edge of lane half 1 (four edges further on).
// The accumulators live in the AGPR half of the register file (the kernel needs ~340 registers, VALU instructions address 256):
// each value is fetched with ONE v_accvgpr_read right before its swap.  Left to itself hipcc copies all 64 registers of the
// previous tile to VGPRs at the top of the loop and keeps them there (512 registers, spills).  hipcc pads no hazards for an asm
// statement: inside the loop six MFMAs of the next tile lie between the last MFMA that wrote these registers and the first
// read; the tail, which reads right behind the last MFMAs, waits explicitly (e9_settle).
__device__ __forceinline__ void lane_half_swap(float a_in, float b_in, float& first, float& second) {
    // (asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 loses the builtin's SECOND result - it reuses the first one, see
    // scripts/dev/edge9_probe.cpp / the git log of this file; `s_nop 1`: the VALU-write -> permlane wait states hipcc would add)
    float a, b;
    asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1"
                 : "=&v"(a), "=&v"(b)
                 : "a"(a_in), "a"(b_in));
    first = a;
    second = b;
}

// the four accumulators of a finished tile: ordered behind the MFMAs that wrote them (they are operands), kept in AGPRs, and
// `wait` != 0: the matrix pipe has drained before anything reads them (two MFMAs can be in flight: 2 x 32 + 18 cycles)
template <bool WAIT>
__device__ __forceinline__ void e9_settle(f32x16 (&t)[4]) {
    if (WAIT) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                           : "+a"(t[0]), "+a"(t[1]), "+a"(t[2]), "+a"(t[3]));
    else asm volatile("" : "+a"(t[0]), "+a"(t[1]), "+a"(t[2]), "+a"(t[3]));
}

template <bool WAIT>
__device__ __forceinline__ void e9_settle2(f32x16 (&t)[2]) {
    if (WAIT) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(t[0]), "+a"(t[1]));
    else asm volatile("" : "+a"(t[0]), "+a"(t[1]));
}

// one output value of the previous tile: x = acc / 2^11 + b and r = 1 / (1 + 2^-x'); Swish = x r
template <bool L2E>
__device__ __forceinline__ void swish_parts(float a, float bias, float& x, float& r) {
    x = fmaf(a, 1.0f / E9_SCALE, bias);
    r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -x : -x * MAGNET_LOG2E));
}

// ... -> running sum:  sum += x / (1 + 2^-x')
template <bool L2E>
__device__ __forceinline__ void swish_accumulate(float a, float bias, float& sum) {
    const float x = fmaf(a, 1.0f / E9_SCALE, bias);
    const float e = __builtin_amdgcn_exp2f(L2E ? -x : -x * MAGNET_LOG2E);
    sum = fmaf(x, __builtin_amdgcn_rcpf(1.0f + e), sum);
}

// The run that ends at edge e leaves: 64 features per store, lane = feature.  Same protocol as emit_run (edge_tile.cuh):
// finished rows are divided by their degree, the two runs crossing the ends of the wave's range go to `partials`.
__device__ __forceinline__ void emit_run64(float s0, float s1, const RunMeta& md, int e, int e_lo, int e_hi, int range,
                                           float deg_scale, float* __restrict__ agg, float* __restrict__ partials, int lane) {
    const int ti = __builtin_amdgcn_readlane(md.ti, e);
    const int rs = __builtin_amdgcn_readlane(md.rs, e), re = __builtin_amdgcn_readlane(md.re, e);
    float* dst;
    if (rs >= e_lo && re <= e_hi) {
        const float dn = (float)(re - rs) * deg_scale;
        s0 = s0 / dn;
        s1 = s1 / dn;
        dst = agg + (size_t)ti * 128;
    } else {
        dst = partials + ((size_t)range * 2 + (rs < e_lo ? 0 : 1)) * 128;
    }
    dst[lane] = s0;
    dst[64 + lane] = s1;
}

// emit_run64 for ONE feature half (k_mpnn_edge32p: the two halves of a row leave half a tile apart); fcol = 64 p + lane
__device__ __forceinline__ void emit_run_half(float sv, const RunMeta& md, int e, int e_lo, int e_hi, int range, float deg_scale,
                                              float* __restrict__ agg, float* __restrict__ partials, int fcol) {
    const int ti = __builtin_amdgcn_readlane(md.ti, e);
    const int rs = __builtin_amdgcn_readlane(md.rs, e), re = __builtin_amdgcn_readlane(md.re, e);
    float* dst;
    if (rs >= e_lo && re <= e_hi) {
        sv = sv / ((float)(re - rs) * deg_scale);
        dst = agg + (size_t)ti * 128;
    } else {
        dst = partials + ((size_t)range * 2 + (rs < e_lo ? 0 : 1)) * 128;
    }
    dst[fcol] = sv;
}

// register of edge e inside the swapped accumulator pair: e = 8 g + w -> index 4 g + (w & 3), first (w < 4) or second block
__device__ __forceinline__ constexpr int e9_reg(int e) { return 4 * (e >> 3) + (e & 3); }

}  // namespace

// ABL (development builds only, results INVALID): 1 no MFMAs, 2 no epilogue transcendentals, 4 no input Swish / split, 8 gathers from
// L1-resident rows, 16 no LDS weight reads, 32 no gather loads at all - to attribute the kernel's time (scripts/bench_edge9_abl.py).
//
// What the ablations showed (profiles/r3/edge9_ablations.txt): removing the 96 MFMAs of a tile changes NOTHING (1.11 ms either way) -
// the matrix pipe runs entirely in the shadow of the other ~990 instructions, and those take ~9.5 cycles each although a wave can
// issue one every ~4.5: a wave issues in order, so every instruction that needs the result of a transcendental (or of another VALU
// instruction) a few slots earlier stalls the whole stream.  Hence the Swish evaluations are software-pipelined ACROSS double
// groups: stage 1 (A + Bm / bias fma, 2^-x) in double group k, stage 2 (1 + e, reciprocal) in k + 1, stage 3 (the fp16 pieces /
// the running sum) in k + 2 - inside one double group no instruction depends on another one of the same group.
template <bool L2E, int ABL = 0>
__global__ void __launch_bounds__(256, 1) k_mpnn_edge32t(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                         const int32_t* __restrict__ rowptr, int E,
                                                         const u32x4* __restrict__ w2p /* [8 s][4 nt][2 planes][64] */,
                                                         const float* __restrict__ b2, float* __restrict__ agg,
                                                         float* __restrict__ partials) {
    constexpr float deg_scale = L2E ? MAGNET_LOG2E : 1.0f;            // the Swish outputs carry log2 e when the inputs do
    __shared__ __attribute__((aligned(16))) u32x4 wl[8 * 4 * 2 * 64];  // 64 KiB
    for (int i = threadIdx.x; i < 8 * 4 * 2 * 64; i += 256) wl[i] = w2p[i];
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
    const float bias0 = b2[lane] * deg_scale, bias1 = b2[64 + lane] * deg_scale;
    const u32x4* const wlane = wl + lane;

    u32x4 xh[8], xl[8];                       // fp16 pieces of the current tile's activations, one k slot (16 columns) each
    f32x4 la[4][2], lb[4][2];                 // gather ring: A[i] / Bm[j] columns 16 s + 8 half + 0..7 of one slot per entry
    f32x16 accA[4], accB[4];                  // GEMM outputs of tile t / tile t - 1 (roles alternate)
    float sum0 = 0.f, sum1 = 0.f;             // running sums of the current run, features lane and 64 + lane
    // Swish pipelines, four buffers of four values each (three in flight; 16 double groups per tile = a multiple of four, so the
    // buffer of double group dg is dg & 3 in every tile): inputs iv = A + Bm, iq = 2^-v then 1 / (1 + 2^-v);
    // outputs ex = acc / 2^11 + b, eq likewise
    float iv[4][4], iq[4][4], ex[4][4], eq[4][4];

    auto request = [&](int entry, int slot, const float* pa, const float* pb) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            la[entry][g] = *reinterpret_cast<const f32x4*>(pa + 16 * slot + 4 * g);
            lb[entry][g] = *reinterpret_cast<const f32x4*>(pb + 16 * slot + 4 * g);
        }
    };
    auto ex2 = [](float x) { return __builtin_amdgcn_exp2f(L2E ? -x : -x * MAGNET_LOG2E); };
    // half of a slot (four values), all three stages at once (prologue only)
    auto consume = [&](int entry, int slot, int hf) {
        u32 h0, l0, h1, l1;
        const f32x4 a = la[entry][hf], b = lb[entry][hf];
        swish_split_pair<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
        swish_split_pair<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
        xh[slot][2 * hf] = h0, xh[slot][2 * hf + 1] = h1;
        xl[slot][2 * hf] = l0, xl[slot][2 * hf + 1] = l1;
    };
    // ---- the three stages of the input pipeline (buffer b; stage 3 writes half hf of slot `slot`)
    auto in1 = [&](int b, int entry, int hf) {
        const f32x4 a = la[entry][hf], c = lb[entry][hf];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            iv[b][i] = a[i] + c[i];
            iq[b][i] = (ABL & 4) ? iv[b][i] : ex2(iv[b][i]);
        }
    };
    auto in2 = [&](int b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) iq[b][i] = (ABL & 4) ? iq[b][i] : __builtin_amdgcn_rcpf(1.0f + iq[b][i]);
    };
    auto in3 = [&](int b, int slot, int hf) {
        u32 h0, l0, h1, l1;
        if (ABL & 4) {
            h0 = __builtin_bit_cast(u32, iv[b][0]), l0 = __builtin_bit_cast(u32, iq[b][1]);
            h1 = __builtin_bit_cast(u32, iv[b][2]), l1 = __builtin_bit_cast(u32, iq[b][3]);
        } else {
            split_pair_vr(iv[b][0], iq[b][0], iv[b][1], iq[b][1], h0, l0);
            split_pair_vr(iv[b][2], iq[b][2], iv[b][3], iq[b][3], h1, l1);
        }
        xh[slot][2 * hf] = h0, xh[slot][2 * hf + 1] = h1;
        xl[slot][2 * hf] = l0, xl[slot][2 * hf + 1] = l1;
    };

    TileIdx ix = tile_idx<32>(tile, e_hi, src, tgt, j);
    TileIdx ix1 = tile_idx<32>(min(tile + 1, tend - 1), e_hi, src, tgt, j);
    RunMeta md;
    {   // prologue: the first tile's slots 0-6 are computed here, its slot 7 and the next tile's slots 0 / 1 are left in the ring
        const float* pa = A + (size_t)ix.ti * 128 + 8 * half;
        const float* pb = Bm + (size_t)ix.sj * 128 + 8 * half;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            request(s & 3, s, pa, pb);
            consume(s & 3, s, 0);
            consume(s & 3, s, 1);
        }
        request(3, 7, pa, pb);
        const float* pan = A + (size_t)ix1.ti * 128 + 8 * half;
        const float* pbn = Bm + (size_t)ix1.sj * 128 + 8 * half;
        request(0, 0, pan, pbn);
        request(1, 1, pan, pbn);
        md.ti = ix.ti;
        md.rs = rowptr[ix.ti];
        md.re = rowptr[ix.ti + 1];
        md.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix.e_ok && ix.tnext != ix.ti);   // both halves agree: low word
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) accB[nt][i] = 0.f;            // "previous tile" of the first iteration (its sums are dropped)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) iv[b][i] = iq[b][i] = ex[b][i] = eq[b][i] = 0.f;
    }
    RunMeta pmd = md, ppmd = md;               // meta of the previous tile / of the one before (its last four edges finish two
    pmd.ends = 0, ppmd.ends = 0;               // double groups into the next tile); no previous tile yet: nothing leaves
    bool fresh = true;                         // no real tile has been summed yet
    asm volatile("" ::"v"(ix1.ti), "v"(ix1.sj), "v"(ix1.tnext), "v"(md.rs), "v"(md.re));   // see k_mpnn_edge7

    // stage 3 of the output pipeline: sum += x r for the two edges e0, e0 + 1 (both feature halves), run ends leave
    auto out3 = [&](int b, int e0, const RunMeta& meta) __attribute__((always_inline)) {
        const unsigned m2 = (meta.ends >> e0) & 3u;
        if (m2 == 0) {
            sum0 = fmaf(ex[b][2], eq[b][2], fmaf(ex[b][0], eq[b][0], sum0));
            sum1 = fmaf(ex[b][3], eq[b][3], fmaf(ex[b][1], eq[b][1], sum1));
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sum0 = fmaf(ex[b][2 * c], eq[b][2 * c], sum0);
                sum1 = fmaf(ex[b][2 * c + 1], eq[b][2 * c + 1], sum1);
                if (m2 & (1u << c)) {
                    emit_run64(sum0, sum1, meta, e0 + c, e_lo, e_hi, range, deg_scale, agg, partials, lane);
                    sum0 = 0.f;
                    sum1 = 0.f;
                }
            }
        }
    };

    // one tile: GEMM of `tile` into cur, epilogue of the previous tile from prev, inputs of tile + 1 into xh / xl
    auto body = [&](f32x16 (&cur)[4], f32x16 (&prev)[4], auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        e9_settle<false>(prev);
        const TileIdx ix2 = tile_idx<32>(min(tile + 2, tend - 1), e_hi, src, tgt, j);   // used from double group 12 on
        RunMeta md1;
        md1.ti = ix1.ti;
        md1.rs = rowptr[ix1.ti];
        md1.re = rowptr[ix1.ti + 1];
        const float* pan = A + (size_t)((ABL & 8) ? j : ix1.ti) * 128 + 8 * half;       // clamped: the last iteration prefetches unused data
        const float* pbn = Bm + (size_t)((ABL & 8) ? j : ix1.sj) * 128 + 8 * half;
        u32x4 wr[2][2][2];                    // weight fragments: [buffer][block of the pair][plane]
        float late0[4], late1[4];             // second halves of the swapped registers of the current group of eight edges
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            wr[0][u][0] = wlane[(u * 2 + 0) * 64];
            wr[0][u][1] = wlane[(u * 2 + 1) * 64];
        }
        // double group dg = 2 s + q: k slot s, output blocks nt = 2 q and 2 q + 1; the six MFMAs alternate between the two
        // accumulators.  Stage 1 of: edges 2 dg, 2 dg + 1 of the previous tile and half of slot (s + 7) & 7 of the inputs;
        // stage 2 of what double group dg - 1 started, stage 3 of what dg - 2 started.
#pragma unroll
        for (int dg = 0; dg < 16; ++dg) {
            const int s = dg >> 1, q = dg & 1, n0 = 2 * q, n1 = 2 * q + 1, bsel = dg & 1;
            const int b1 = dg & 3, b2 = (dg + 3) & 3, b3 = (dg + 2) & 3, dg3 = (dg + 14) & 15;   // dg3: the double group stage 3 finishes
            if (dg + 1 < 16) {
                const int s1 = (dg + 1) >> 1, q1 = (dg + 1) & 1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (ABL & 16) continue;
                    wr[(dg + 1) & 1][u][0] = wlane[(((s1 * 4 + 2 * q1 + u) * 2) + 0) * 64];
                    wr[(dg + 1) & 1][u][1] = wlane[(((s1 * 4 + 2 * q1 + u) * 2) + 1) * 64];
                }
            }
            if (ABL & 1) {                        // no MFMAs: the operands still have to exist
                if (s == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        cur[n0][i] = __builtin_bit_cast(float, xh[s][i] ^ wr[bsel][0][0][i] ^ wr[bsel][0][1][i]);
                        cur[n1][i] = __builtin_bit_cast(float, xl[s][i] ^ wr[bsel][1][0][i] ^ wr[bsel][1][1][i]);
                    }
                } else {
                    cur[n0][s] = __builtin_bit_cast(float, xh[s][0] ^ wr[bsel][0][0][1] ^ wr[bsel][0][1][2]);
                    cur[n1][s] = __builtin_bit_cast(float, xl[s][1] ^ wr[bsel][1][0][3] ^ wr[bsel][1][1][0]);
                }
            } else if (s == 0) {                  // first k slot: the accumulators start from the inline constant 0
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                cur[n0] = mfma_xw(xh[s], wr[bsel][0][1], z);
                cur[n1] = mfma_xw(xh[s], wr[bsel][1][1], z);
            } else {
                cur[n0] = mfma_xw(xh[s], wr[bsel][0][1], cur[n0]);
                cur[n1] = mfma_xw(xh[s], wr[bsel][1][1], cur[n1]);
            }
            if (!(ABL & 1)) {
                cur[n0] = mfma_xw(xh[s], wr[bsel][0][0], cur[n0]);
                cur[n1] = mfma_xw(xh[s], wr[bsel][1][0], cur[n1]);
                cur[n0] = mfma_xw(xl[s], wr[bsel][0][0], cur[n0]);
                cur[n1] = mfma_xw(xl[s], wr[bsel][1][0], cur[n1]);
            }
            // ---- outputs, stage 2 (started by dg - 1) and stage 1 (edges 2 dg, 2 dg + 1 of the previous tile)
#pragma unroll
            for (int i = 0; i < 4; ++i) eq[b2][i] = (ABL & 2) ? eq[b2][i] : __builtin_amdgcn_rcpf(1.0f + eq[b2][i]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = 2 * dg + c, r = e9_reg(e), w = e & 7;
                float v0, v1;
                if (w < 4) {                      // register r of the four blocks: edge e now, edge e + 4 four edges later
                    lane_half_swap(prev[0][r], prev[1][r], v0, late0[w]);      // features lane
                    lane_half_swap(prev[2][r], prev[3][r], v1, late1[w]);      // features 64 + lane
                } else {
                    v0 = late0[w - 4];
                    v1 = late1[w - 4];
                }
                ex[b1][2 * c] = fmaf(v0, 1.0f / E9_SCALE, bias0);
                ex[b1][2 * c + 1] = fmaf(v1, 1.0f / E9_SCALE, bias1);
                eq[b1][2 * c] = (ABL & 2) ? bias0 : ex2(ex[b1][2 * c]);
                eq[b1][2 * c + 1] = (ABL & 2) ? bias1 : ex2(ex[b1][2 * c + 1]);
            }
            // ---- inputs: stage 3 (started by dg - 2), stage 2 (dg - 1), stage 1: slot 7 of THIS tile in double groups 0 / 1,
            // slots 0-6 of the next tile in double groups 2-15
            if (!FIRST || dg >= 2) in3(b3, ((dg3 >> 1) + 7) & 7, dg3 & 1);
            if (!FIRST || dg >= 1) in2(b2);
            {
                const int sc = (s + 7) & 7;
                in1(b1, sc & 3, dg & 1);
            }
            // ---- gather requests, 6 double groups ahead of their consumption (ring entry = slot & 3)
            if ((dg & 1) == 0 && !(ABL & 32)) {
                const int sr = (dg >> 1) + 2;     // 2 .. 9
                if (sr < 8) {
                    request(sr & 3, sr, pan, pbn);
                } else {                          // slots 0 / 1 of tile + 2
                    const float* pa2 = A + (size_t)((ABL & 8) ? j : ix2.ti) * 128 + 8 * half;
                    const float* pb2 = Bm + (size_t)((ABL & 8) ? j : ix2.sj) * 128 + 8 * half;
                    request(sr & 3, sr - 8, pa2, pb2);
                }
            }
            // ---- the order hipcc has to emit this block in: the LDS requests of the next double group first, then every MFMA
            // followed by its share of the ~50 VALU / transcendental instructions (a wave issues in order: MFMAs back to back
            // stall the issue for 32 cycles each, VALU issued right behind an MFMA runs in its shadow)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, E9_VALU_PER_MFMA, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- outputs, stage 3 (started by dg - 2): running sums and run ends, one branch per double group.  Double groups 0 / 1
            // finish edges 28-31 of the tile BEFORE the previous one
            if (dg < 2) out3(b3, 2 * dg3, ppmd);
            else out3(b3, 2 * dg3, pmd);
            if (!FIRST && dg == 1 && fresh) {      // everything summed so far belonged to the zero tile in front of the first real one
                sum0 = 0.f, sum1 = 0.f;
                fresh = false;
            }
        }
        ppmd = pmd;
        pmd = md;
        md1.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix1.e_ok && ix1.tnext != ix1.ti);
        md = md1;
        ix1 = ix2;
    };

    bool last_in_A = true;
    body(accA, accB, std::true_type{});
    ++tile;
    for (; tile < tend; ++tile) {
        body(accB, accA, std::false_type{});
        last_in_A = false;
        if (++tile >= tend) break;
        body(accA, accB, std::false_type{});
        last_in_A = true;
    }
    // the pipeline's last two double groups (edges 28-31 of the tile before the last one), then the last tile's rows
    auto tail = [&](f32x16 (&prev)[4]) __attribute__((always_inline)) {
        float late0[4], late1[4];
        if (fresh) {                           // a single tile: only the zero tile has been through the pipeline
            sum0 = 0.f, sum1 = 0.f;
        } else {
            out3(2, 28, ppmd);                 // double group 14: x and r are complete
#pragma unroll
            for (int i = 0; i < 4; ++i) eq[3][i] = (ABL & 2) ? eq[3][i] : __builtin_amdgcn_rcpf(1.0f + eq[3][i]);
            out3(3, 30, ppmd);                 // double group 15: stage 2 was still due
        }
        e9_settle<true>(prev);
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int r = e9_reg(e), w = e & 7;
            float v0, v1;
            if (w < 4) {
                lane_half_swap(prev[0][r], prev[1][r], v0, late0[w]);
                lane_half_swap(prev[2][r], prev[3][r], v1, late1[w]);
            } else {
                v0 = late0[w - 4];
                v1 = late1[w - 4];
            }
            swish_accumulate<L2E>(v0, bias0, sum0);
            swish_accumulate<L2E>(v1, bias1, sum1);
            if (pmd.ends & (1u << e)) {
                emit_run64(sum0, sum1, pmd, e, e_lo, e_hi, range, deg_scale, agg, partials, lane);
                sum0 = 0.f;
                sum1 = 0.f;
            }
        }
    };
    if (last_in_A) tail(accA);
    else tail(accB);
}


// ------------------------------------------------------------------------------------------------
// The same tile, TWO waves per SIMD (512-thread workgroups): k_mpnn_edge32p.  k_mpnn_edge32t keeps two full sets of
// accumulators (this tile's and the previous tile's, 128 registers) and ~390 registers in all; one wave per SIMD then has nobody to
// fill its s_waitcnt and dependency stalls (SQ counters: 27 % of the wave cycles).  Here the tile is computed as two PAIRS of
// 32-feature blocks, one after the other (double group dg = 8 np + s: pair np, k slot s), so that only two pairs of accumulators
// are alive at any time (64 registers): while pair 0 of tile t accumulates (dg 0-7) the epilogue of pair 1 of tile t-1 runs in the
// MFMA shadows, and while pair 1 accumulates (dg 8-15) the epilogue of pair 0 of THIS tile runs.  The activations of the
// tile serve both passes, so the inputs of tile t+1 can only replace them during the second pass (slot sc in double group
// 9 + sc, slot 7 in double group 0 of the next tile); their gathered rows are requested E9P_LEAD double groups earlier into a
// ring of E9P_LEAD register sets - a short lead (the loads take ~2 000 cycles under load), but the partner wave runs meanwhile.
#ifndef E9P_LEAD
#define E9P_LEAD 2
#endif
template <bool L2E>
__global__ void __launch_bounds__(512, 2) k_mpnn_edge32p(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                         const int32_t* __restrict__ rowptr, int E,
                                                         const u32x4* __restrict__ w2p /* [8 s][4 nt][2 planes][64] */,
                                                         const float* __restrict__ b2, float* __restrict__ agg,
                                                         float* __restrict__ partials) {
    constexpr float deg_scale = L2E ? MAGNET_LOG2E : 1.0f;
    constexpr int NR = E9P_LEAD;                                       // ring entries
    __shared__ __attribute__((aligned(16))) u32x4 wl[8 * 4 * 2 * 64];  // 64 KiB
    for (int i = threadIdx.x; i < 8 * 4 * 2 * 64; i += 512) wl[i] = w2p[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int ntiles = (E + 31) / 32;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges, see k_mpnn_edge6
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 8 - 1) / (gridDim.x * 8);
    const int range = vb * 8 + wave;
    int tile = range * tpw;
    const int tend = min(tile + tpw, ntiles);
    if (tile >= tend) return;
    const int e_lo = tile * 32, e_hi = min(tend * 32, E);           // this wave's edges
    const float bias0 = b2[lane] * deg_scale, bias1 = b2[64 + lane] * deg_scale;
    const u32x4* const wlane = wl + lane;

    u32x4 xh[8], xl[8];                       // fp16 pieces of the current tile's activations, one k slot each
    f32x4 la[NR][2], lb[NR][2];               // gather ring
    f32x16 P0[2], P1[2];                      // accumulators of pair 0 (features 0-63) and pair 1 (features 64-127)
    float sum0 = 0.f, sum1 = 0.f;             // running sums of the two feature halves (they run half a tile apart)

    auto request = [&](int entry, int slot, const float* pa, const float* pb) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            la[entry][g] = *reinterpret_cast<const f32x4*>(pa + 16 * slot + 4 * g);
            lb[entry][g] = *reinterpret_cast<const f32x4*>(pb + 16 * slot + 4 * g);
        }
    };
    auto consume = [&](int entry, int slot) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            u32 h0, l0, h1, l1;
            const f32x4 a = la[entry][hf], b = lb[entry][hf];
            swish_split_pair<L2E>(a[0] + b[0], a[1] + b[1], h0, l0);
            swish_split_pair<L2E>(a[2] + b[2], a[3] + b[3], h1, l1);
            xh[slot][2 * hf] = h0, xh[slot][2 * hf + 1] = h1;
            xl[slot][2 * hf] = l0, xl[slot][2 * hf + 1] = l1;
        }
    };

    TileIdx ix = tile_idx<32>(tile, e_hi, src, tgt, j);
    RunMeta md;
    {   // prologue: the first tile's eight slots
        const float* pa = A + (size_t)ix.ti * 128 + 8 * half;
        const float* pb = Bm + (size_t)ix.sj * 128 + 8 * half;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            request(s % NR, s, pa, pb);
            consume(s % NR, s);
        }
        md.ti = ix.ti;
        md.rs = rowptr[ix.ti];
        md.re = rowptr[ix.ti + 1];
        md.ends = (unsigned)__builtin_amdgcn_ballot_w64(ix.e_ok && ix.tnext != ix.ti);   // both halves agree: low word
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) P1[u][i] = 0.f;               // "previous tile" of the first iteration (its sum is dropped)
    }
    RunMeta pmd = md;
    pmd.ends = 0;                              // no previous tile yet: nothing leaves in the first half of the first iteration

    // four edges of one accumulator pair -> x and r of their Swish (lane = feature of the pair)
    auto pair_edges = [&](f32x16 (&P)[2], int g4 /* 0..7: edges 4 g4 .. 4 g4 + 3 */, float bias, float (&late)[4], float (&xv)[4],
                          float (&rv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 4 * g4 + c, r = e9_reg(e), w = e & 7;
            float v;
            if (w < 4) lane_half_swap(P[0][r], P[1][r], v, late[w]);
            else v = late[w - 4];
            swish_parts<L2E>(v, bias, xv[c], rv[c]);
        }
    };
    auto accumulate = [&](const float (&xv)[4], const float (&rv)[4], unsigned m4, int e0, float& sum, const RunMeta& meta,
                          int foff) __attribute__((always_inline)) {
        if (m4 == 0) {
            sum = fmaf(xv[3], rv[3], fmaf(xv[2], rv[2], fmaf(xv[1], rv[1], fmaf(xv[0], rv[0], sum))));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sum = fmaf(xv[c], rv[c], sum);
                if (m4 & (1u << c)) {
                    emit_run_half(sum, meta, e0 + c, e_lo, e_hi, range, deg_scale, agg, partials, foff + lane);
                    sum = 0.f;
                }
            }
        }
    };

    bool first = true;
    for (; tile < tend; ++tile) {
        const TileIdx ixn = tile_idx<32>(min(tile + 1, tend - 1), e_hi, src, tgt, j);   // used from double group 9 - LEAD on
        RunMeta md1;
        u32x4 wr[2][2][2];                    // weight fragments: [buffer][block of the pair][plane]
        float late[4];
        e9_settle2<false>(P1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            wr[0][u][0] = wlane[(u * 2 + 0) * 64];
            wr[0][u][1] = wlane[(u * 2 + 1) * 64];
        }
        const float* pan = nullptr;
        const float* pbn = nullptr;
#pragma unroll
        for (int dg = 0; dg < 16; ++dg) {
            const int np = dg >> 3, s = dg & 7, bsel = dg & 1;
            f32x16(&cur)[2] = np ? P1 : P0;
            if (dg + 1 < 16) {
                const int np1 = (dg + 1) >> 3, s1 = (dg + 1) & 7;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    wr[(dg + 1) & 1][u][0] = wlane[(((s1 * 4 + 2 * np1 + u) * 2) + 0) * 64];
                    wr[(dg + 1) & 1][u][1] = wlane[(((s1 * 4 + 2 * np1 + u) * 2) + 1) * 64];
                }
            }
            // ---- epilogue: pair 1 of the previous tile while pair 0 accumulates, pair 0 of this tile while pair 1 accumulates.
            // The accumulator reads come BEFORE this double group's MFMAs in program order when they read the pair that is about to
            // be restarted (dg 8 reads P1 of the previous tile?  no: dg 0-7 read P1, which restarts at dg 8; dg 8-15 read P0, which
            // restarts at dg 0 of the next tile) - every register is read in an earlier double group than the MFMA that overwrites it.
            float xv[4], rv[4];
            if (np == 0) pair_edges(P1, s, bias1, late, xv, rv);
            else pair_edges(P0, s, bias0, late, xv, rv);
            if (s == 0) {                         // first k slot: the accumulators start from the inline constant 0
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                cur[0] = mfma_xw(xh[s], wr[bsel][0][1], z);
                cur[1] = mfma_xw(xh[s], wr[bsel][1][1], z);
            } else {
                cur[0] = mfma_xw(xh[s], wr[bsel][0][1], cur[0]);
                cur[1] = mfma_xw(xh[s], wr[bsel][1][1], cur[1]);
            }
            cur[0] = mfma_xw(xh[s], wr[bsel][0][0], cur[0]);
            cur[1] = mfma_xw(xh[s], wr[bsel][1][0], cur[1]);
            cur[0] = mfma_xw(xl[s], wr[bsel][0][0], cur[0]);
            cur[1] = mfma_xw(xl[s], wr[bsel][1][0], cur[1]);
            // ---- inputs of the next tile: slot sc replaces this tile's slot sc once both passes have used it
            if (dg >= 9) consume((dg - 9) % NR, dg - 9);
            if (dg == 0 && !first) consume(7 % NR, 7);                 // (slot 7 of THIS tile, requested by the previous iteration)
            // ---- gather requests E9P_LEAD double groups ahead (slot sc is consumed in double group 9 + sc; 7 -> dg 0 of the next tile)
            if (dg == 9 - NR) {
                pan = A + (size_t)ixn.ti * 128 + 8 * half;             // clamped: the last iteration prefetches unused data
                pbn = Bm + (size_t)ixn.sj * 128 + 8 * half;
                md1.ti = ixn.ti;
                md1.rs = rowptr[ixn.ti];
                md1.re = rowptr[ixn.ti + 1];
            }
            if (dg >= 9 - NR && dg - (9 - NR) < 8) request((dg - (9 - NR)) % NR, dg - (9 - NR), pan, pbn);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, E9_VALU_PER_MFMA, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- running sums and run ends (wave-uniform): one branch per double group
            if (np == 0) accumulate(xv, rv, (pmd.ends >> (4 * s)) & 15u, 4 * s, sum1, pmd, 64);
            else accumulate(xv, rv, (md.ends >> (4 * s)) & 15u, 4 * s, sum0, md, 0);
            if (dg == 7 && first) sum1 = 0.f;    // the first iteration summed the zero "previous tile": drop it
        }
        first = false;
        pmd = md;
        md1.ends = (unsigned)__builtin_amdgcn_ballot_w64(ixn.e_ok && ixn.tnext != ixn.ti);
        md = md1;
    }
    // pair 1 of the last tile
    {
        float late[4];
        e9_settle2<true>(P1);
#pragma unroll
        for (int g4 = 0; g4 < 8; ++g4) {
            float xv[4], rv[4];
            pair_edges(P1, g4, bias1, late, xv, rv);
            accumulate(xv, rv, (pmd.ends >> (4 * g4)) & 15u, 4 * g4, sum1, pmd, 64);
        }
    }
}

#ifdef MAGNET_DEV_VARIANTS
int g_edge9_ablation = 0;          // development builds only: magnet_debug_set_variant(2, flags), see k_mpnn_edge32t
#endif

// message_net_2 as fp16 planes [8 k slots][4 output blocks][2 planes][64 lanes][8 fp16], plane 0 = 2^11 fp16(w), plane 1 =
// 2^11 (w - fp16(w)) (magnet_amd/packing.py:pack_edge32t_fp16x2); b2 the plain fp32 bias.
extern "C" int magnet_mpnn_edge_fp16x3_t32_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                               const int32_t* rowptr, int n_edges, int n_rows, const void* w2_t32,
                                               const float* b2, float* agg, float* partials, int a_bm_log2e_scaled,
                                               void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (n_edges + 31) / 32;
    const bool l2e = (a_bm_log2e_scaled & 1) != 0, paired = (a_bm_log2e_scaled & 2) != 0;     // bit 1: k_mpnn_edge32p (two waves per SIMD)
    const int wpb = paired ? 8 : 4;
    const int nblocks = tiles > 0 ? min((tiles + wpb - 1) / wpb, ncu) : 0;
    if (nblocks > 0) {
        const dim3 g(nblocks), b(64 * wpb);
        const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
        if (paired) {
            if (l2e) hipLaunchKernelGGL(k_mpnn_edge32p<true>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
            else hipLaunchKernelGGL(k_mpnn_edge32p<false>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
        } else {
#ifdef MAGNET_DEV_VARIANTS
#define E9_ABL_CASE(v) case v: hipLaunchKernelGGL((k_mpnn_edge32t<true, v>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
            if (g_edge9_ablation && l2e) {
                switch (g_edge9_ablation) {
                    E9_ABL_CASE(1) E9_ABL_CASE(2) E9_ABL_CASE(4) E9_ABL_CASE(6) E9_ABL_CASE(7) E9_ABL_CASE(8) E9_ABL_CASE(16) E9_ABL_CASE(17)
                    E9_ABL_CASE(31) E9_ABL_CASE(30) E9_ABL_CASE(24) E9_ABL_CASE(32) E9_ABL_CASE(63) E9_ABL_CASE(33) E9_ABL_CASE(48)
                    default: return -2;
                }
            } else
#endif
            if (l2e) hipLaunchKernelGGL(k_mpnn_edge32t<true>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
            else hipLaunchKernelGGL(k_mpnn_edge32t<false>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
        }
    }
    const int nranges = nblocks * wpb, tpw = nranges ? (tiles + nranges - 1) / nranges : 1;      // as computed inside the kernel
    return magnet_launch_edge_range_fixup(tgt, rowptr, n_rows, n_edges, tpw, 5, nranges, l2e ? MAGNET_LOG2E : 1.0f,
                                          partials, agg, stream);
}
