// Helpers of the kernels that keep a whole [128 x 128] fp32 weight-gradient accumulator PER WAVE in the accumulator half of the
// register file (mpnn_edge_bwd_wp.hip: k_mpnn_edge_bwd_wp; the same construction for magnet_linear_bwd128_f32 was built and measured
// slower: profiles/r4/linear_bwd_gen2.txt): asm MFMAs with an explicit register file,
// the pair-of-output-blocks GEMM over operands that already are fp16 planes, and the per-wave LDS block that turns "lane = row,
// registers = features" plane words into "lane = feature, registers = rows" MFMA operands.
#pragma once
#include "common.cuh"
#include "split_single.cuh"

#define WP_RS 36                          // words per row of the transposition block (32 edges + 4: conflict-free 16-byte reads)
#define WP_PLANE (16 * WP_RS)             // one fp16 plane of a 32-feature block
#define WP_SCR (2 * WP_PLANE)

// The register FILE of each accumulator is the source's choice: left alone, hipcc puts EVERY MFMA result in the accumulator half
// (a[0:255]) once a kernel uses it - the 96 accumulator registers of the data path on top of the weight gradient's 256: 329 spilled
// registers and 936 v_accvgpr copies per tile.  The data path's MFMAs are builtins compiled with -mllvm -amdgpu-mfma-vgpr-form (arch
// VGPRs; hipcc knows their latency and hazards), the weight gradient's are asm with "+a" (hipcc does not see an MFMA in an asm
// statement: the wait states before the first read of such a result are written out, and magnet_amd/hazard_check.py checks the
// linked code object).  Sources that include this header are compiled with that flag and -fno-slp-vectorize (magnet_amd/build.py).
__device__ __forceinline__ void mfma_v(f32x16& acc, const f16x8& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_a(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_results_ready(f32x16& a0, f32x16& a1) {
    asm volatile("s_nop 10" : "+v"(a0), "+v"(a1));                      // 8 passes + 2 wait states, + 1
}

// a0 / a1 (+)= 2^11 W x for output blocks 2 PR, 2 PR + 1; x given as its two fp16 planes, one u32x4 per k slot (slot = kt * 2 + m)
// fill(i), i = 0 .. 47, is placed behind the i-th MFMA: independent VALU work of another phase that runs while the matrix pipe is busy
// (one wave per SIMD: nothing else overlaps the two pipes; an MFMA occupies the pipe for 32 cycles = eight fp32 VALU instructions)
struct NoFill {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int PR, class Fill = NoFill>
__device__ __forceinline__ void gemm_pair_planes(const u32x4* __restrict__ wl, const u32x4 (&xh)[8], const u32x4 (&xl)[8], f32x16& a0,
                                                 f32x16& a1, int lane, Fill fill = Fill()) {
    u32x4 wr[2][2][2];
    auto wload = [&](int slot) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const u32x4* wp = wl + ((((slot >> 1) * 4 + 2 * PR + b) * 2 + (slot & 1)) * 2) * 64 + lane;
            wr[slot & 1][b][0] = wp[0];
            wr[slot & 1][b][1] = wp[64];
        }
    };
    wload(0);
#pragma unroll
    for (int slot = 0; slot < 8; ++slot) {
        if (slot + 1 < 8) wload(slot + 1);
        const f16x8 h = __builtin_bit_cast(f16x8, xh[slot]), l = __builtin_bit_cast(f16x8, xl[slot]);
        const f16x8 w0h = __builtin_bit_cast(f16x8, wr[slot & 1][0][0]), w0l = __builtin_bit_cast(f16x8, wr[slot & 1][0][1]);
        const f16x8 w1h = __builtin_bit_cast(f16x8, wr[slot & 1][1][0]), w1l = __builtin_bit_cast(f16x8, wr[slot & 1][1][1]);
        mfma_v(a0, w0l, h);
        fill(6 * slot);
        __builtin_amdgcn_sched_barrier(0);
        mfma_v(a1, w1l, h);
        fill(6 * slot + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_v(a0, w0h, l);
        fill(6 * slot + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_v(a1, w1h, l);
        fill(6 * slot + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_v(a0, w0h, h);
        fill(6 * slot + 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_v(a1, w1h, h);
        fill(6 * slot + 5);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one 32-feature block, lane = edge: the 8 + 8 plane words of the block (word w = features 8 (w >> 1) + 4 half + 2 (w & 1), + 1)
__device__ __forceinline__ void wp_write_block(unsigned* __restrict__ sw, const u32x4& h0, const u32x4& h1, const u32x4& l0,
                                               const u32x4& l1) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int row = 4 * (w >> 1) + (w & 1);                   // feature pair = row + 2 half (in sw)
        sw[row * WP_RS] = w < 4 ? h0[w] : h1[w - 4];
        sw[row * WP_RS + WP_PLANE] = w < 4 ? l0[w] : l1[w - 4];
    }
}
// ... read back with lane = feature: edges 16 ks + 8 (lane >> 5) + 0..7 of feature (lane & 31) as one MFMA operand per plane
__device__ __forceinline__ void wp_read_block(const unsigned* __restrict__ sr, unsigned sel, u32x4 (&hi)[2], u32x4 (&lo)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(sr + 16 * ks), b = *reinterpret_cast<const u32x4*>(sr + 16 * ks + 4);
        const u32x4 c = *reinterpret_cast<const u32x4*>(sr + WP_PLANE + 16 * ks);
        const u32x4 d = *reinterpret_cast<const u32x4*>(sr + WP_PLANE + 16 * ks + 4);
        hi[ks][0] = __builtin_amdgcn_perm(a[1], a[0], sel);
        hi[ks][1] = __builtin_amdgcn_perm(a[3], a[2], sel);
        hi[ks][2] = __builtin_amdgcn_perm(b[1], b[0], sel);
        hi[ks][3] = __builtin_amdgcn_perm(b[3], b[2], sel);
        lo[ks][0] = __builtin_amdgcn_perm(c[1], c[0], sel);
        lo[ks][1] = __builtin_amdgcn_perm(c[3], c[2], sel);
        lo[ks][2] = __builtin_amdgcn_perm(d[1], d[0], sel);
        lo[ks][3] = __builtin_amdgcn_perm(d[3], d[2], sel);
    }
}

// the same in two halves, for callers that put other work between the LDS reads and the moment their data are needed
__device__ __forceinline__ void wp_read_raw(const unsigned* __restrict__ sr, u32x4 (&raw)[8]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        raw[4 * ks + 0] = *reinterpret_cast<const u32x4*>(sr + 16 * ks);
        raw[4 * ks + 1] = *reinterpret_cast<const u32x4*>(sr + 16 * ks + 4);
        raw[4 * ks + 2] = *reinterpret_cast<const u32x4*>(sr + WP_PLANE + 16 * ks);
        raw[4 * ks + 3] = *reinterpret_cast<const u32x4*>(sr + WP_PLANE + 16 * ks + 4);
    }
}
__device__ __forceinline__ void wp_perm(const u32x4 (&raw)[8], unsigned sel, u32x4 (&hi)[2], u32x4 (&lo)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const u32x4 &a = raw[4 * ks], &b = raw[4 * ks + 1], &c = raw[4 * ks + 2], &d = raw[4 * ks + 3];
        hi[ks][0] = __builtin_amdgcn_perm(a[1], a[0], sel);
        hi[ks][1] = __builtin_amdgcn_perm(a[3], a[2], sel);
        hi[ks][2] = __builtin_amdgcn_perm(b[1], b[0], sel);
        hi[ks][3] = __builtin_amdgcn_perm(b[3], b[2], sel);
        lo[ks][0] = __builtin_amdgcn_perm(c[1], c[0], sel);
        lo[ks][1] = __builtin_amdgcn_perm(c[3], c[2], sel);
        lo[ks][2] = __builtin_amdgcn_perm(d[1], d[0], sel);
        lo[ks][3] = __builtin_amdgcn_perm(d[3], d[2], sel);
    }
}

// 16 bytes at a wave-uniform base + a 32-bit per-lane byte offset (global_load / store with an SGPR base)
__device__ __forceinline__ f32x4 ldg4(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg4(float* __restrict__ base, unsigned byte_off, const f32x4& v) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot_ones(unsigned w, float acc) {
    const f16x2_t one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w), one, acc, false);
}

