// fp16 split GEMM pieces with ONE accumulator, and the LDS hand-over helpers of the kernels that form a weight gradient next to
// their data path (mpnn_edge_bwd.hip: k_mpnn_edge_bwd_w; linear_bwd.hip: k_linear_bwd128).  See the header comment of k_mpnn_edge_bwd_w.
#pragma once
#include "common.cuh"

// one word of each fp16 plane from two values: h = RTZ fp16, l = RNE fp16 of the exact remainder, NOT scaled
__device__ __forceinline__ void split_f16_plain(float a0, float a1, unsigned& h, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
    float d0, d1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(h), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(h), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, 0" : "=v"(l) : "v"(d0));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, 0" MAGNET_MIX_TAIL : "+v"(l) : "v"(d1));    // (wait states: common.cuh, MAGNET_MIX_TAIL)
}
// the same planes of the PRODUCTS a0 m0, a1 m1 in four instructions instead of two multiplies + five: v_fma_mix multiplies, subtracts the
// high piece and converts in one step (h = RNE fp16(a m); l = RNE fp16(a m - h) with the product exact inside the fma) - the
// forward edge kernels' construction (mpnn_edge_rows.hip: swish_pieces)
__device__ __forceinline__ void split_f16_prod(float a0, float m0, float a1, float m1, unsigned& h, unsigned& l) {
    unsigned hh, ll;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %5, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" MAGNET_MIX_TAIL
        : "=&v"(hh), "=&v"(ll)
        : "v"(a0), "v"(a1), "v"(m0), "v"(m1));
    h = hh;
    l = ll;
}
// 8 consecutive edges of one staged feature row -> the two planes of an MFMA operand
__device__ __forceinline__ float stage_operand(const float* __restrict__ row, f16x8& hi, f16x8& lo) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row), b = *reinterpret_cast<const f32x4*>(row + 4);
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_f16_plain(a[0], a[1], h0, l0);
    split_f16_plain(a[2], a[3], h1, l1);
    split_f16_plain(b[0], b[1], h2, l2);
    split_f16_plain(b[2], b[3], h3, l3);
    const u32x4 h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
}

// acc[nt] += 2^11 W x for the wave's [128 x 32] register tile with ONE accumulator per output block: the planes in LDS are
// (2^11 hw, 2^11 lw) - plane 0 of the standard fp16x3 pack scaled while it was copied in - and the activation's low piece is the
// unscaled fp16 remainder, so the three products of the split share one scale (no cross accumulator, no fold: 64 registers that the
// dW2 slab needs).  Two output blocks per group: consecutive MFMAs never write the same accumulator.
template <int NT>
__device__ __forceinline__ void gemm_single(const u32x4* __restrict__ wl, const f32x16 (&x)[NT], f32x16 (&acc)[NT], int lane) {
    static_assert(NT % 2 == 0, "output blocks are taken in pairs");
    constexpr int NPR = NT / 2, NG = 2 * NT * NPR;  // pairs per slot, groups: gq = slot * NPR + pair, slot = kt * 2 + m
    u32x4 xs[2][2], wr[2][2][2];                   // xs[slot parity][plane]; wr[group parity][block of the pair][plane]
    auto quarter = [&](int slot, int q) {
        unsigned h, l;
        const f32x16& t = x[slot >> 1];
        const int m = slot & 1;
        split_f16_plain(t[8 * m + 2 * q], t[8 * m + 2 * q + 1], h, l);
        xs[slot & 1][0][q] = h;
        xs[slot & 1][1][q] = l;
    };
    auto wload = [&](int gq) {
        const int slot = gq / NPR, pr = gq % NPR;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const u32x4* wp = wl + ((((slot >> 1) * NT + 2 * pr + b) * 2 + (slot & 1)) * 2) * 64 + lane;
            wr[gq & 1][b][0] = wp[0];
            wr[gq & 1][b][1] = wp[64];
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) quarter(0, q);
    wload(0);
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        const int slot = gq / NPR, pr = gq % NPR;
        if (gq + 1 < NG) wload(gq + 1);
        if (slot + 1 < 2 * NT) {                   // the next slot's split, spread over this slot's groups
#pragma unroll
            for (int q = pr * 4 / NPR; q < (pr + 1) * 4 / NPR; ++q) quarter(slot + 1, q);
        }
        const f16x8 xh = __builtin_bit_cast(f16x8, xs[slot & 1][0]), xl = __builtin_bit_cast(f16x8, xs[slot & 1][1]);
        const f16x8 w0h = __builtin_bit_cast(f16x8, wr[gq & 1][0][0]), w0l = __builtin_bit_cast(f16x8, wr[gq & 1][0][1]);
        const f16x8 w1h = __builtin_bit_cast(f16x8, wr[gq & 1][1][0]), w1l = __builtin_bit_cast(f16x8, wr[gq & 1][1][1]);
        f32x16& a0 = acc[2 * pr];
        f32x16& a1 = acc[2 * pr + 1];
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0l, xh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, xh, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, xl, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, xl, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0h, xh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, xh, a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void gemm128_single(const u32x4* __restrict__ wl, const f32x16 (&x)[4], f32x16 (&acc)[4], int lane) {
    gemm_single<4>(wl, x, acc, lane);
}
// plane 0 (the fp16 high pieces) of the standard pack times 2^11
__device__ __forceinline__ u32x4 scale_plane_2048(u32x4 v) {
    // (one vector multiply: written as a loop over the four words that assigns o[c], hipcc 7.2 scales word 0 and stores it in all four)
    return __builtin_bit_cast(u32x4, (f16x8)(__builtin_bit_cast(f16x8, v) * (_Float16)2048.0f));
}

// workgroup barrier for LDS hand-overs only: __syncthreads() also waits for vmcnt(0), i.e. for the dp stores and the prefetched rows
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// per-block maxima of |x| over n4 float4s (the launch-wide power-of-two scale G of a staged gradient); defined in mpnn_edge_bwd.hip
int magnet_launch_absmax_partial(const float* x, long n4, float* part, int nblocks, void* stream);
