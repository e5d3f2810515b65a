// Shared device primitives for the gfx950 (MI355X) MAgNet kernels.
//
// Core idea ("register-chained transposed MLP"): a wave owns 32 rows (edges, nodes or INR
// queries).  Every Linear is computed as  Y^T = W * X^T  with v_mfma_f32_32x32x2_f32:
//   A operand = weight fragment  W[n0 + (lane&31)][k]      (pre-packed, 1 KiB coalesced loads)
//   B operand = activation       X[row = lane&31][k]
//   C/D       = Y^T tile: lane (row = lane&31, half = lane>>5) register r holds
//               Y[row][n0 + (r&3) + 8*(r>>2) + 4*half]
// The k index an MFMA step consumes is free to choose as long as A and B agree; step s of a
// 32-wide k tile uses k = (s&3) + 8*(s>>2) + 4*half -- exactly the C/D register map.  So the
// output registers of one layer ARE the B operands of the next: a whole MLP runs in VGPRs with
// no LDS round trip, no transposes and no barriers.  fp32 MFMA is exact fp32 (fmaf chain), so
// results match the reference's fp32 Linear up to summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MAGNET_MAX_KT 36      // first-layer K up to 1152 (a 3x3 convolution over 128 channels as 9 gathered taps)
#define MAGNET_MAX_LAYERS 6
#ifndef MAGNET_PF
#define MAGNET_PF 0
#endif

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2 };
enum { EPI_STORE = 0, EPI_STORE_RES = 1, EPI_SEGMEAN = 2, EPI_STORE_MASK = 3,   // MASK: y = res > 0 ? mlp(x) : 0 (ReLU backward)
       EPI_QUADBLEND = 4 };   // y[r / 4] = sum over the 4 consecutive rows 4 (r / 4) .. + 3 of res[row] * mlp(x)[row]  (res: [M] weights)

// one source of a first-layer k tile: value = scale * p[(idx ? idx[row] : row / rdiv) * ld + col + k]
struct MagnetSrc {
    const float* p;
    const int32_t* idx;
    int32_t ld;
    int32_t col;
    float scale;
    int32_t rdiv;      // >=1: dense sources may be shared by rdiv consecutive rows
};
struct MagnetKTile {
    MagnetSrc s[2];
    int32_t n;         // 1 or 2 sources (summed)
    int32_t pad_[3];
};
struct MagnetLayer {
    const f32x4* w;    // packed [KT][NT][4][64] float4 (see magnet_hip.h)
    const float* b;    // [NT*32] zero padded
};
struct MagnetMlpArgs {
    int32_t M;                     // rows
    int32_t nkt;                   // first layer k tiles
    MagnetKTile kt[MAGNET_MAX_KT];
    MagnetLayer layer[MAGNET_MAX_LAYERS];
    const float* ln_g;             // LayerNorm gain/bias (ncols), used when LN
    const float* ln_b;
    float ln_eps;
    // store epilogue
    float* y;
    int32_t ldy;
    int32_t ncols;                 // valid output columns (<= OT*32)
    const float* res;              // residual (same layout as y) for EPI_STORE_RES
    int32_t ldres;
    // segment-mean epilogue (rows are CSR-ordered edges)
    const int32_t* tgt;            // [M] target of each edge
    const int32_t* rowptr;         // [N+1]
    float* partials;               // [ntiles][2][ncols]
    int32_t ncolgroups;            // NL == 1 only: > 1 splits the output columns over gridDim.y, group g uses layer[g] (own packed slice)
    int32_t groups_in_block;       // > 1 (= ncolgroups): the groups run one after the other inside each workgroup instead of over gridDim.y
    int64_t y_group_stride;        // floats between the outputs of consecutive groups; 0 = OT*32 columns of the same rows
    MagnetSrc pre[2];              // gathered rows added to the first Linear's pre-activation (split-GEMM kernels), p == nullptr: off
    float* x0_out;                 // tape outputs (training backward), nullptr = off: assembled first-layer input rows [M][nkt*32]
    float* hid[MAGNET_MAX_LAYERS]; // hid[l]: layer l's activated output [M][HT*32]; hid[NL-1]: raw output of the last Linear [M][OT*32]
                                   // (NL == 2: hid[2] = the hidden layer's pre-activation)
    uint64_t x0_skip;              // bit kt set: k tile kt of x0_out is not written
    int32_t* status;               // fp16 split modes, NL > 1, nullptr = off: `status_bit` is OR-ed in when a HIDDEN activation reaches 65504
    int32_t status_bit;
    int32_t pad_;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float act_swish(float x) {
    // x * sigmoid(x); exp overflow -> rcp(inf) = 0 -> -0, the correct limit
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// Swish of x given xs = log2(e) * x, returned with the same factor: xs * sigmoid(x) = xs / (1 + 2^-xs).  One multiply fewer
// than act_swish (v_exp_f32 is a base-2 exponential and takes the negation as an operand modifier); the factor is linear,
// so a GEMM or a mean behind it carries it along until somebody divides it out.
#define MAGNET_LOG2E 1.44269504088896340736f
__device__ __forceinline__ float act_swish_l2e(float xs) {
    return xs * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-xs));
}

template <int ACT>
__device__ __forceinline__ void apply_act(f32x16& v) {
    if (ACT == ACT_SWISH) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = act_swish(v[i]);
    } else if (ACT == ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.0f);
    }
}

// running maximum of |v| over a register tile (v_max3_f32 with |.| modifiers: one instruction per two values)
__device__ __forceinline__ void track_abs_max(const f32x16& v, float& m) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(m, fabsf(v[i])), fabsf(v[i + 1]));
    asm volatile("" : "+v"(m));             // formed HERE: a result that is only read at the end of the kernel is sunk there, with the tile alive
}

// accumulators <- bias (lane (row, half) register 4g+c holds column nt*32 + 8g + 4half + c)
template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&out)[NT], const float* __restrict__ b, int half) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(b + nt * 32 + 8 * g + 4 * half);
            out[nt][4 * g + 0] = t[0];
            out[nt][4 * g + 1] = t[1];
            out[nt][4 * g + 2] = t[2];
            out[nt][4 * g + 3] = t[3];
        }
    }
}

// out[nt] += src.p[(idx ? idx[row] : row) * ld + the lane's columns of tile nt]: a gathered row in the accumulator layout
template <int NT>
__device__ __forceinline__ void add_rows(f32x16 (&out)[NT], const MagnetSrc& src, int row, int half) {
    const int r = src.idx ? src.idx[row] : row;
    const float* p = src.p + (size_t)r * src.ld + 4 * half;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + nt * 32 + 8 * g);
            out[nt][4 * g + 0] += t[0];
            out[nt][4 * g + 1] += t[1];
            out[nt][4 * g + 2] += t[2];
            out[nt][4 * g + 3] += t[3];
        }
    }
}

// out[nt] += W(kt, nt) * in   for one 32-wide k tile
template <int NT>
__device__ __forceinline__ void mma_ktile(const f32x16& in, f32x16 (&out)[NT], const f32x4* __restrict__ w,
                                           int kt, int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4* wp = w + (size_t)((kt * NT + nt) * 4) * 64 + lane;
        const f32x4 w0 = wp[0], w1 = wp[64], w2 = wp[128], w3 = wp[192];
#pragma unroll
        for (int c = 0; c < 4; ++c) out[nt] = mfma32(w0[c], in[c], out[nt]);
#pragma unroll
        for (int c = 0; c < 4; ++c) out[nt] = mfma32(w1[c], in[4 + c], out[nt]);
#pragma unroll
        for (int c = 0; c < 4; ++c) out[nt] = mfma32(w2[c], in[8 + c], out[nt]);
#pragma unroll
        for (int c = 0; c < 4; ++c) out[nt] = mfma32(w3[c], in[12 + c], out[nt]);
    }
}

// load the 4 float4 weight fragments of block (kt, nt)
template <int NT>
__device__ __forceinline__ void load_wblock(f32x4 (&w)[4], const f32x4* __restrict__ wp, int kt, int nt, int lane) {
    const f32x4* p = wp + (size_t)((kt * NT + nt) * 4) * 64 + lane;
    w[0] = p[0];
    w[1] = p[64];
    w[2] = p[128];
    w[3] = p[192];
}
__device__ __forceinline__ void mma_block(const f32x4 (&w)[4], const f32x16& in, f32x16& out) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) out = mfma32(w[g][c], in[4 * g + c], out);
}

// register-resident layer: out = W * in + b.  PF selects how weight fragments reach the MFMAs:
//   0  plain loop, hipcc schedules the loads
//   1  3-deep register ring, sched_barriers pin each prefetch two blocks (32 MFMAs) ahead
//   2  3-deep register ring without the barriers
template <int KT, int NT, int PF = MAGNET_PF>
__device__ __forceinline__ void dense_layer(const f32x16 (&in)[KT], f32x16 (&out)[NT], const MagnetLayer& L,
                                             int lane) {
    init_bias<NT>(out, L.b, lane >> 5);
    if (PF == 0) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) mma_ktile<NT>(in[kt], out, L.w, kt, lane);
        return;
    }
    if (PF == 3) {      // all NT accumulators interleaved per k step (no back-to-back dependent MFMAs)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 w[NT][4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) load_wblock<NT>(w[nt], L.w, kt, nt, lane);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) out[nt] = mfma32(w[nt][g][c], in[kt][4 * g + c], out[nt]);
        }
        return;
    }
    constexpr int NB = KT * NT;
    f32x4 w[3][4];
    load_wblock<NT>(w[0], L.w, 0, 0, lane);
    if (NB > 1) load_wblock<NT>(w[1], L.w, 1 / NT, 1 % NT, lane);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b + 2 < NB) load_wblock<NT>(w[(b + 2) % 3], L.w, (b + 2) / NT, (b + 2) % NT, lane);
        if (PF == 1) __builtin_amdgcn_sched_barrier(0);
        mma_block(w[b % 3], in[b / NT], out[b % NT]);
        if (PF == 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores ("bf16x6 split").  Every fp32 value is written EXACTLY as
// hi + mid + lo with three bf16 pieces (8 significant bits each, truncation), and a product x*w is
// evaluated as the six partial products whose weight is >= 2^-16 of the leading one
// (hh, hm, mh, hl, lh, mm); each partial product is exact in fp32 and the dropped terms are <= 3*2^-24
// of the product, i.e. fp32-rounding class (measured 1.7e-7 relative L2 vs fp64 at K = 128, the plain
// fp32 fmaf chain gives 2.1e-7).  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so six
// of them are 2.7x faster, and - unlike v_mfma_f32_32x32x2_f32, which was measured not to hide any VALU
// work - VALU instructions issued between bf16 MFMAs of the same wave are hidden.
// Fragment order: one instruction contracts 16 k's; lane (row, half) supplies slots c' = 0..7 <->
// k = 8*(2m + (c'>>2)) + 4*half + (c'&3) of the 32-wide k tile, m = 0,1 - i.e. registers 8m .. 8m+7
// of the C/D map, so a layer's output registers are again the next layer's B operands.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_top16(float v1, float v0) {   // (top16(v1) << 16) | top16(v0)
    return __builtin_amdgcn_perm(__float_as_uint(v1), __float_as_uint(v0), 0x07060302u);
}
// split 8 fp32 values into the three bf16 planes (exact)
__device__ __forceinline__ void split_bf16x3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    u32x4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a0 = v[2 * p], a1 = v[2 * p + 1];
        const float r0 = a0 - __uint_as_float(__float_as_uint(a0) & 0xFFFF0000u);
        const float r1 = a1 - __uint_as_float(__float_as_uint(a1) & 0xFFFF0000u);
        const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u);
        const float q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
        h[p] = pack_top16(a1, a0);
        m[p] = pack_top16(r1, r0);
        l[p] = pack_top16(q1, q0);
    }
    hi = __builtin_bit_cast(bf16x8, h);
    mid = __builtin_bit_cast(bf16x8, m);
    lo = __builtin_bit_cast(bf16x8, l);
}
// the same split, one register pair (= one 32-bit word of each plane) at a time so that it can be
// spread over MFMA shadows: word p of planes[0..2] <- values 8m+2p, 8m+2p+1 of the k tile
__device__ __forceinline__ void split_quarter(const f32x16& tile, int m, int p, u32x4 (&planes)[3]) {
    const float a0 = tile[8 * m + 2 * p], a1 = tile[8 * m + 2 * p + 1];
    const float r0 = a0 - __uint_as_float(__float_as_uint(a0) & 0xFFFF0000u);
    const float r1 = a1 - __uint_as_float(__float_as_uint(a1) & 0xFFFF0000u);
    const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u);
    const float q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
    planes[0][p] = pack_top16(a1, a0);
    planes[1][p] = pack_top16(r1, r0);
    planes[2][p] = pack_top16(q1, q0);
}
__device__ __forceinline__ void split_pair4(const f32x16& tile, int m, u32x4 (&planes)[3]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) split_quarter(tile, m, p, planes);
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// acc += W(slot) * X for one 16-k slot: the six retained partial products, small ones first
__device__ __forceinline__ f32x16 mma_split6(bf16x8 wh, bf16x8 wm, bf16x8 wl, bf16x8 xh, bf16x8 xm, bf16x8 xl,
                                             f32x16 acc) {
    acc = mfma_bf16(wm, xm, acc);
    acc = mfma_bf16(wl, xh, acc);
    acc = mfma_bf16(wh, xl, acc);
    acc = mfma_bf16(wm, xh, acc);
    acc = mfma_bf16(wh, xm, acc);
    acc = mfma_bf16(wh, xh, acc);
    return acc;
}

// ------------------------------------------------------------------------------------------------
// "fp16x3": the same idea with two fp16 pieces.  x = h + l/2048 with h = RTZ_fp16(x) and
// l = RNE_fp16((x - h) * 2048) (the remainder is exact and the 2^11 scaling keeps it out of the fp16 subnormals);
// x*w = h*hw + 2^-11 (h*lw + l*hw) + O(2^-22): three v_mfma_f32_32x32x16_f16 per 16-k slot - half the matrix-pipe
// cycles and about half the split instructions of bf16x6 - with the cross terms kept in their own accumulator
// and folded in once per layer.  Measured 2.2-3.1e-7 relative L2 vs fp64 at K = 128 (fp32 fmaf chain 2.0e-7).
// Range: |x| must stay below the fp16 maximum 65504 (true for normalised PDE fields, coordinates and every
// post-norm hidden state on this path); bf16x6 has the full fp32 range and remains selectable.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
enum { GEMM_BF16X6 = 0, GEMM_FP16X3 = 1, GEMM_BF16 = 2, GEMM_FP16X3S = 3 };

template <int MODE>
struct Split;

// hipcc's hazard recognizer does not look inside inline asm: between an asm VALU instruction that writes a VGPR and a v_mfma that reads it
// as SrcA / SrcB / SrcC it inserts nothing, where it keeps two wait states for every VALU instruction it emitted itself.
// MAGNET_MIX_TAIL closes the asm block of the LAST piece of a plane word with the missing wait states.
#ifndef MAGNET_MIX_TAIL
#define MAGNET_MIX_TAIL "\n\ts_nop 1"
#endif
// ... and one wait state between a transcendental (v_exp / v_rcp ...) and a VALU instruction that reads its result (gfx940+ "trans
// forwarding hazard"): asm blocks whose inputs come straight out of a Swish start with MAGNET_TRANS_HEAD
#define MAGNET_TRANS_HEAD "s_nop 0\n\t"
// the same guarantee for a whole operand at once: every write to the two planes happens before, every MFMA that reads them after
__device__ __forceinline__ void mfma_operand_fence(u32x4& hi, u32x4& lo) { asm volatile("s_nop 1" : "+v"(hi), "+v"(lo)); }

template <>
struct Split<GEMM_BF16X6> {
    static constexpr int NP = 3;                       // planes per slot
    static __device__ __forceinline__ void quarter(const f32x16& tile, int m, int p, u32x4 (&planes)[3]) {
        split_quarter(tile, m, p, planes);
    }
    // wp: plane 0 of the slot for this lane, planes 64 u32x4 apart (LDS or registers copied from it)
    static __device__ __forceinline__ void mma(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x16& accM, f32x16&) {
        accM = mma_split6(__builtin_bit_cast(bf16x8, w[0]), __builtin_bit_cast(bf16x8, w[1]), __builtin_bit_cast(bf16x8, w[2]),
                          __builtin_bit_cast(bf16x8, x[0]), __builtin_bit_cast(bf16x8, x[1]), __builtin_bit_cast(bf16x8, x[2]), accM);
    }
    static __device__ __forceinline__ void fold(f32x16&, const f32x16&) {}
};

template <>
struct Split<GEMM_FP16X3> {
    static constexpr int NP = 2;
    static __device__ __forceinline__ void quarter(const f32x16& tile, int m, int p, u32x4 (&planes)[2]) {
        const u32x2 w = pair(tile[8 * m + 2 * p], tile[8 * m + 2 * p + 1]);
        planes[0][p] = w[0];
        planes[1][p] = w[1];
    }
    // one word of each plane from two values: {(h1 << 16 | h0), (l1 << 16 | l0)}.  TAIL = false: without the closing wait states - the
    // caller fences the finished planes itself (mfma_operand_fence) before an MFMA reads them.
    template <bool TAIL = true>
    static __device__ __forceinline__ u32x2 pair(float a0, float a1) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
        // five instructions per value pair (hipcc's own lowering of the C expression takes nine): the remainder a - h is a
        // mixed-precision fma that reads the fp16 piece in place (exact, no conversion back to fp32), and the two scaled
        // remainders are rounded (RNE) straight into the low / high half of the plane word
        float d0, d1;
        unsigned l;
        const float k2048 = 2048.f;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(h), "v"(a0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(h), "v"(a1));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(d0), "s"(k2048));
        if (TAIL) asm("v_fma_mixhi_f16 %0, %1, %2, 0" MAGNET_MIX_TAIL : "+v"(l) : "v"(d1), "s"(k2048));
        else asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(d1), "s"(k2048));
        const u32x2 w = {h, l};
        return w;
    }
    static __device__ __forceinline__ void mma(const u32x4 (&w)[2], const u32x4 (&x)[2], f32x16& accM, f32x16& accC) {
        const f16x8 wh = __builtin_bit_cast(f16x8, w[0]), wl = __builtin_bit_cast(f16x8, w[1]);
        const f16x8 xh = __builtin_bit_cast(f16x8, x[0]), xl = __builtin_bit_cast(f16x8, x[1]);
        // the main product sits between the two cross products: no two consecutive MFMAs write the same accumulator
        // (the matrix pipe stalls on a back-to-back dependent accumulate), and each accumulator's own order is unchanged
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accC, 0, 0, 0);
        accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accM, 0, 0, 0);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accC, 0, 0, 0);
    }
    static __device__ __forceinline__ void fold(f32x16& accM, const f32x16& accC) {
#pragma unroll
        for (int i = 0; i < 16; ++i) accM[i] = fmaf(accC[i], 1.0f / 2048.0f, accM[i]);
    }
};

// "fp16x3s": the same three products in ONE accumulator (the arithmetic of split_single.cuh / k_mpnn_edge32t).  The
// weight planes are (2^11 hw, 2^11 lw) - packing mode "fp16x3s" scales the high plane and the bias on the way in, |w| < 31.98 -
// and the activation's low piece is the UNSCALED fp16 remainder (an fp16 subnormal below |x| ~ 0.12, which the MFMA keeps: absolute
// error 2^-25 per value), so all three products carry 2^11 and a layer ends with one multiply by 2^-11 instead of
// "zero the cross accumulator, fold it in": per value and layer 1 (descale) + 1.5 (split: cvt_pkrtz per pair, one mixed-precision
// fma per value) VALU instructions against 1 + 1 + 2.5, and NT x 16 fewer registers.
template <>
struct Split<GEMM_FP16X3S> {
    static constexpr int NP = 2;
    static __device__ __forceinline__ void quarter(const f32x16& tile, int m, int p, u32x4 (&planes)[2]) {
        const float a0 = tile[8 * m + 2 * p], a1 = tile[8 * m + 2 * p + 1];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
        unsigned l;
        // l = fp16_rne(a - h): the difference is exact inside the fma (h is a truncation of a), one rounding into the plane word
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" MAGNET_MIX_TAIL : "+v"(l) : "v"(h), "v"(a1));
        planes[0][p] = h;
        planes[1][p] = l;
    }
    static __device__ __forceinline__ void mma(const u32x4 (&w)[2], const u32x4 (&x)[2], f32x16& accM, f32x16&) {
        const f16x8 wh = __builtin_bit_cast(f16x8, w[0]), wl = __builtin_bit_cast(f16x8, w[1]);
        const f16x8 xh = __builtin_bit_cast(f16x8, x[0]), xl = __builtin_bit_cast(f16x8, x[1]);
        accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accM, 0, 0, 0);
        accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accM, 0, 0, 0);
        accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accM, 0, 0, 0);
    }
    // two output blocks at a time: consecutive MFMAs never write the same accumulator
    static __device__ __forceinline__ void mma2(const u32x4 (&w0)[2], const u32x4 (&w1)[2], const u32x4 (&x)[2], f32x16& a0, f32x16& a1) {
        const f16x8 w0h = __builtin_bit_cast(f16x8, w0[0]), w0l = __builtin_bit_cast(f16x8, w0[1]);
        const f16x8 w1h = __builtin_bit_cast(f16x8, w1[0]), w1l = __builtin_bit_cast(f16x8, w1[1]);
        const f16x8 xh = __builtin_bit_cast(f16x8, x[0]), xl = __builtin_bit_cast(f16x8, x[1]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, xh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, xh, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, xl, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, xl, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, xh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, xh, a1, 0, 0, 0);
    }
    static __device__ __forceinline__ void fold(f32x16& accM, const f32x16&) {
#pragma unroll
        for (int i = 0; i < 16; ++i) accM[i] *= 1.0f / 2048.0f;
    }
};

// Reduced precision (opt-in, BASELINE.json config 2): ONE bf16 piece per value, round-to-nearest-even, one MFMA product.
__device__ __forceinline__ unsigned bf16_rne_pair(float v1, float v0) {          // (bf16(v1) << 16) | bf16(v0)
    const unsigned a = __float_as_uint(v0), b = __float_as_uint(v1);
    const unsigned ra = a + 0x7FFFu + ((a >> 16) & 1u), rb = b + 0x7FFFu + ((b >> 16) & 1u);
    return __builtin_amdgcn_perm(rb, ra, 0x07060302u);
}
template <>
struct Split<GEMM_BF16> {
    static constexpr int NP = 1;
    static __device__ __forceinline__ void quarter(const f32x16& tile, int m, int p, u32x4 (&planes)[1]) {
        planes[0][p] = bf16_rne_pair(tile[8 * m + 2 * p + 1], tile[8 * m + 2 * p]);
    }
    static __device__ __forceinline__ void mma(const u32x4 (&w)[1], const u32x4 (&x)[1], f32x16& accM, f32x16&) {
        accM = mfma_bf16(__builtin_bit_cast(bf16x8, w[0]), __builtin_bit_cast(bf16x8, x[0]), accM);
    }
    static __device__ __forceinline__ void fold(f32x16&, const f32x16&) {}
};

template <int MODE>
__device__ __forceinline__ void split_slot(const f32x16& tile, int m, u32x4 (&planes)[Split<MODE>::NP]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) Split<MODE>::quarter(tile, m, p, planes);
}
template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) t[n][i] = 0.f;
}

// first-layer k tile from the (gathered / concatenated / differenced) sources
// Row selection and data fetch are separate steps so that a kernel can request the indices of k tile kt+2 while the
// data of k tile kt+1 is in flight (a gathered load is two dependent memory round trips; done back to back inside the k
// loop they stall the wave twice per tile).
struct KRows {
    long r0, r1;
};
__device__ __forceinline__ KRows ktile_rows(const MagnetKTile& d, int row) {
    KRows r;
    r.r0 = d.s[0].idx ? (long)d.s[0].idx[row] : (long)(row / d.s[0].rdiv);
    r.r1 = 0;
    if (d.n > 1) r.r1 = d.s[1].idx ? (long)d.s[1].idx[row] : (long)(row / d.s[1].rdiv);
    return r;
}
__device__ __forceinline__ f32x16 ktile_data(const MagnetKTile& d, const KRows& r, int half) {
    f32x16 v;
    {
        const MagnetSrc& a = d.s[0];
        const float* p = a.p + r.r0 * a.ld + a.col + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 8 * g);
            v[4 * g + 0] = t[0] * a.scale;
            v[4 * g + 1] = t[1] * a.scale;
            v[4 * g + 2] = t[2] * a.scale;
            v[4 * g + 3] = t[3] * a.scale;
        }
    }
    if (d.n > 1) {
        const MagnetSrc& a = d.s[1];
        const float* p = a.p + r.r1 * a.ld + a.col + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 8 * g);
            v[4 * g + 0] += t[0] * a.scale;
            v[4 * g + 1] += t[1] * a.scale;
            v[4 * g + 2] += t[2] * a.scale;
            v[4 * g + 3] += t[3] * a.scale;
        }
    }
    return v;
}
__device__ __forceinline__ f32x16 load_ktile(const MagnetKTile& d, int row, int half) {
    return ktile_data(d, ktile_rows(d, row), half);
}
// The same fetch in two steps: ktile_raw only ISSUES the loads (the loaded registers are not touched), ktile_finish applies the
// scales / the sum of the two sources.  A kernel that requests k tile kt+1 while it works on k tile kt must not do arithmetic on
// the loaded values in the issuing phase - the multiply by `scale` in ktile_data makes the wave wait for the data right there (cycle
// trace of the C5 edge chain: 3 300 cycles per first-layer iteration between the barrier and the MFMAs, a full memory round trip).
struct KRaw {
    f32x4 a[4];
};
__device__ __forceinline__ KRaw ktile_raw(const MagnetKTile& d, const KRows& r, int half) {
    KRaw v;
    const float* p = d.s[0].p + r.r0 * d.s[0].ld + d.s[0].col + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) v.a[g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
    if (d.n > 1) {             // two summed sources (the edge encoders' u_s - u_r): combined here as before - staging both raw
                               // costs 16 more registers and the 64-wide chains a wave per SIMD (138 -> 170 registers, measured)
        const float* q = d.s[1].p + r.r1 * d.s[1].ld + d.s[1].col + 4 * half;
        const float s0 = d.s[0].scale, s1 = d.s[1].scale;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(q + 8 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) v.a[g][c] = v.a[g][c] * s0 + t[c] * s1;
        }
    }
    return v;
}
__device__ __forceinline__ f32x16 ktile_finish(const MagnetKTile& d, const KRaw& v) {
    f32x16 x;
    const float s0 = d.n > 1 ? 1.0f : d.s[0].scale;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) x[4 * g + c] = v.a[g][c] * s0;
    return x;
}
// accumulator tiles <- gathered rows (pure loads: nothing touches the registers until their first use)
template <int NT>
__device__ __forceinline__ void load_rows(f32x16 (&out)[NT], const MagnetSrc& src, int row, int half) {
    const int r = src.idx ? src.idx[row] : row;
    const float* p = src.p + (size_t)r * src.ld + 4 * half;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + nt * 32 + 8 * g);
            out[nt][4 * g + 0] = t[0];
            out[nt][4 * g + 1] = t[1];
            out[nt][4 * g + 2] = t[2];
            out[nt][4 * g + 3] = t[3];
        }
}

// LayerNorm over the NT*32 columns of each row (row split across lanes l and l^32)
template <int NT>
__device__ __forceinline__ void layer_norm_rows(f32x16 (&v)[NT], const float* __restrict__ g,
                                                 const float* __restrict__ b, float eps, int half) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[nt][i];
    s += __shfl_xor(s, 32);
    const float mean = s / (float)(NT * 32);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d = v[nt][i] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q / (float)(NT * 32) + eps);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n0 = nt * 32 + 8 * gq + 4 * half;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + n0);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b + n0);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[nt][4 * gq + c] = (v[nt][4 * gq + c] - mean) * rstd * gg[c] + bb[c];
        }
}

// one 32-column tile of a row block -> global memory (tape of the first layer's assembled input)
__device__ __forceinline__ void store_ktile(const f32x16& x, float* __restrict__ y, int ldy, int col0, int row, bool row_ok, int half) {
    if (!row_ok) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
        *reinterpret_cast<f32x4*>(y + (size_t)row * ldy + col0 + 8 * g + 4 * half) = t;
    }
}

// rows -> global memory; RES = 1: + residual; RES = 2: kept where res > 0, else 0 (the ReLU mask of a backward-data GEMM: res is the
// layer's post-ReLU activation); ragged widths handled per element
template <int NT, int RES>
__device__ __forceinline__ void store_rows(const f32x16 (&v)[NT], float* __restrict__ y, int ldy, int ncols,
                                            const float* __restrict__ res, int ldres, int row, bool row_ok,
                                            int half, float* vmax = nullptr /* running max of |what was stored| (vector path) */) {
    if (!row_ok) return;
    const bool vec = ((ldy & 3) == 0) && (!RES || (ldres & 3) == 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = nt * 32 + 8 * g + 4 * half;
            if (n0 >= ncols) continue;
            f32x4 t = {v[nt][4 * g], v[nt][4 * g + 1], v[nt][4 * g + 2], v[nt][4 * g + 3]};
            if (vec && n0 + 3 < ncols) {
                if (RES == 1) t += *reinterpret_cast<const f32x4*>(res + (size_t)row * ldres + n0);
                if (RES == 2) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(res + (size_t)row * ldres + n0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = r[c] > 0.f ? t[c] : 0.f;
                }
                *reinterpret_cast<f32x4*>(y + (size_t)row * ldy + n0) = t;
                if (vmax) *vmax = fmaxf(fmaxf(*vmax, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (n0 + c < ncols) {
                        float x = t[c];
                        if (RES == 1) x += res[(size_t)row * ldres + n0 + c];
                        if (RES == 2) x = res[(size_t)row * ldres + n0 + c] > 0.f ? x : 0.f;
                        y[(size_t)row * ldy + n0 + c] = x;
                        if (vmax) *vmax = fmaxf(*vmax, fabsf(x));
                    }
            }
        }
}

#define NH_TS 36                                                    // row stride (floats) of a wave's 32 x 32 transposition tile
// (used by the node kernels of mlp_bf16_kernels.hip and the fused edge backward)
// One 32-row x 32-column tile from the accumulator layout (lane = row, 16 B pieces at a 512 B stride: 32 quarter lines per store
// instruction) through a wave-private LDS tile to whole 128-byte lines (lane l -> row 8 it + l / 8, 16 B piece l % 8)
template <int LDY = 128>
__device__ __forceinline__ void store_tile_lines(const f32x16& v, float* __restrict__ tl, float* __restrict__ y, int col0, long row0,
                                                 long n, int lane) {
    const int j = lane & 31, half = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        *reinterpret_cast<f32x4*>(tl + j * NH_TS + 8 * g + 4 * half) = t;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3);
        const f32x4 t = *reinterpret_cast<const f32x4*>(tl + r * NH_TS + 4 * (lane & 7));
        if (row0 + r < n) *reinterpret_cast<f32x4*>(y + (row0 + r) * LDY + col0 + 4 * (lane & 7)) = t;
    }
}

// ---- DPP lane moves (all lanes must be enabled when these execute: never put them behind a
// short-circuit condition, a DPP read from a disabled lane returns 0)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, true));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, true);
}
#define DPP_SHR1 0x111
#define DPP_SHR2 0x112
#define DPP_SHR4 0x114
#define DPP_SHR8 0x118
#define DPP_BCAST15 0x142

// The LDS slab of segmean_epilogue is private to one wave and a wave's DS operations execute in
// program order, so ordering the slab's writes against its reads only needs the compiler not to
// reorder them: a wavefront-scope fence pair around a wave barrier (no s_barrier, no cross-wave stall).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


