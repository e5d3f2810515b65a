// Fused backward of the MPNN edge block (message_net_1 pre-activation -> Swish -> message_net_2 -> Swish -> mean),
// the adjoint of mpnn_edge_bf16.hip.  Reference: torch autograd through GNN_Layer.message / aggr='mean',
// models/mpnn.py:73-79,46.
//
// Per 32-edge tile (one wave, edges sorted by target i, source j), everything in registers in the transposed
// "lane = edge, registers = features" layout of the forward kernel:
//     p   = A[i] + Bm[j]                      s1 = Swish(p)                 (recomputed, gathers)
//     q   = W2 s1 + b2                        GEMM 1  (fp16x3 split, as the forward)
//     dq  = dagg[i] / deg(i) * Swish'(q)
//     ds1 = W2^T dq                           GEMM 2  (fp16x3 split of dq scaled PER EDGE by a power of two:
//                                             gradients of a mean loss are ~1e-7, below fp16's normal range;
//                                             a row scale is exact for a linear map and is undone afterwards)
//     dp  = ds1 * Swish'(p)
//     dA[i] = sum_e dp                        segmented DPP scan over the target-sorted lanes (+ tile-boundary partials)
// Written to HBM for the follow-up kernels: s1 and dq (weight gradient dW2 = dq^T s1, db2), dp (scatter to the
// *source* nodes for dBm through the by-source CSR permutation).
// Both weight packings (W2 and W2^T, 64 KiB of fp16 planes each) stay resident in LDS; one wave per SIMD.
#include "common.cuh"
#include "edge_tile.cuh"
#include "split_single.cuh"

// EB_INTERLEAVE = 1 moves the s1 / dq stores between the MFMA groups of the two GEMMs (to drain the HBM write queue behind
// the matrix pipe).  Measured slower: hipcc then spills 147 registers (8.1 ms vs 5.3 ms); kept as a switch.
#ifndef EB_INTERLEAVE
#define EB_INTERLEAVE 0
#endif
#ifndef EB_LINE_STORES
#define EB_LINE_STORES 0        // A/B: 1 = s1 / dq / dp through the LDS tile as whole lines (4.23 vs 4.51 ms alone, but 12 spilled VGPRs: the training step is 1 % slower)
#endif

__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float dswish(float x) {
    const float s = sigmoid_fast(x);
    return s * (1.0f + x * (1.0f - s));
}

#ifdef MAGNET_DEV_VARIANTS
// development ablations (results invalid): 1 = no s1 / dq stores, 2 = no dp store, 4 = every store instruction writes 1 KiB of
// consecutive addresses (what an LDS-transposed, row-contiguous store would cost), instead of 64 pieces of 16 B in 32 rows
__device__ int g_ebwd_abl = 0;
#define EBWD_ABL_INIT const int abl = __builtin_amdgcn_readfirstlane(g_ebwd_abl);
extern "C" int magnet_debug_set_edge_bwd_ablation(int value) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ebwd_abl), &value, sizeof(int)); }
#else
#define EBWD_ABL_INIT constexpr int abl = 0;
#endif

__device__ __forceinline__ void store_tile(float* __restrict__ dst_row /* row e, + 4*half */, const f32x16 (&v)[4], int so = 1) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 o = {v[nt][4 * g], v[nt][4 * g + 1], v[nt][4 * g + 2], v[nt][4 * g + 3]};
            *reinterpret_cast<f32x4*>(dst_row + (nt * 32 + 8 * g) * so) = o;
        }
}

// store_tile_lines with the tile's base pointer wave-uniform (SGPR pair) and 32-bit lane offsets: no per-lane 64-bit pointers to keep
__device__ __forceinline__ void store_tile_lines_u(const f32x16& v, float* __restrict__ tl, float* __restrict__ ytile /* uniform */,
                                                   int col0, int rows_valid /* uniform */, int lane) {
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
        if (r < rows_valid) *reinterpret_cast<f32x4*>(ytile + (r * 128 + col0 + 4 * (lane & 7))) = t;
    }
}
// one [32 edges][128] block of a per-edge array: through the LDS tile as whole lines (so == 1), or the ablation's fake layout
__device__ __forceinline__ void store_lines(float* __restrict__ out, const f32x16 (&v)[4], float* __restrict__ tl, int tile, int E,
                                            int lane, bool e_ok, size_t row, int so) {
    if (so != 1 || !EB_LINE_STORES) {
        if (e_ok) store_tile(out + row, v, so);
        return;
    }
    float* yt = out + (size_t)__builtin_amdgcn_readfirstlane(tile) * 4096;
    const int rv = min(32, E - __builtin_amdgcn_readfirstlane(tile) * 32);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) store_tile_lines_u(v[nt], tl, yt, nt * 32, rv, lane);
}

__global__ void __launch_bounds__(256, 1) k_mpnn_edge_bwd(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          const float* __restrict__ dagg, const int32_t* __restrict__ src,
                                                          const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr,
                                                          int E, const u32x4* __restrict__ w2s, const u32x4* __restrict__ w2ts,
                                                          const float* __restrict__ b2, float* __restrict__ s1_out,
                                                          float* __restrict__ dq_out, float* __restrict__ dp_out,
                                                          float* __restrict__ dA, float* __restrict__ partials) {
    constexpr int MODE = GEMM_FP16X3;
    constexpr int NP = Split<MODE>::NP;
    __shared__ __attribute__((aligned(16))) u32x4 wl[32 * NP * 64];          // W2   planes, 64 KiB
    __shared__ __attribute__((aligned(16))) u32x4 wt[32 * NP * 64];          // W2^T planes, 64 KiB
    // b2 in LDS, re-read per tile at an offset the compiler cannot see through: read from global memory the 64 bias registers are loop
    // invariant, hipcc hoists them out of the tile loop and then SPILLS them (they were 56 of the kernel's 70 spilled VGPRs) - and every
    // scratch reload is a VMEM load consumed at once, which (in-order vmcnt) waits for all the stores and prefetches issued before it
    __shared__ __attribute__((aligned(16))) float bl[128];
    // s1 / dq / dp leave through a wave-private 32 x 36-float LDS tile as whole 128-byte lines (store_tile_lines): the accumulator
    // layout (lane = edge) stores 16 B pieces at a 512 B stride - 32 quarter lines per instruction, measured 0.85 ms of the kernel
#if EB_LINE_STORES
    __shared__ __attribute__((aligned(16))) float tls[4][32 * NH_TS];
#endif
    for (int i = threadIdx.x; i < 32 * NP * 64; i += 256) {
        wl[i] = w2s[i];
        wt[i] = w2ts[i];
    }
    if (threadIdx.x < 128) bl[threadIdx.x] = b2[threadIdx.x];
    __syncthreads();
    EBWD_ABL_INIT
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
#if EB_LINE_STORES
    float* tl = tls[wave];
#else
    float* tl = nullptr;
#endif
    const int ntiles = (E + 31) / 32;
    // XCD-aware contiguous tile ranges, as in k_mpnn_edge6 (workgroups go round-robin over the 8 XCDs and their L2s)
    const int nb8 = gridDim.x / 8;
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    int tile = (vb * 4 + wave) * tpw;
    const int tend = min(tile + tpw, ntiles);
    constexpr int stride = 1;
    if (tile >= tend) return;
    // software pipeline over the tiles of this wave: the Bm rows of tile t+1 are requested before GEMM 2 of tile t, its A
    // rows after it, the dagg rows of tile t before its GEMM 1 - with one wave per SIMD nothing else hides the latency.
    f32x16 ga[4], gb[4];
    {
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
                for (int c = 0; c < 4; ++c) {
                    ga[kt][4 * g + c] = a[c];
                    gb[kt][4 * g + c] = b[c];
                }
            }
    }
    // the tile's CSR bookkeeping (target, next target, row bounds) is loaded one tile ahead as well: read at the top of its own tile it
    // would be the wave's youngest load, and waiting for it (in-order vmcnt) waits for the previous tile's dp stores
    int m_ti, m_tnext, m_rs, m_re;
    {
        const int e_raw = tile * 32 + j;
        m_ti = tgt[e_raw < E ? e_raw : E - 1];
        m_tnext = (e_raw + 1 < E && j < 31) ? tgt[e_raw + 1] : -2;
        m_rs = rowptr[m_ti];
        m_re = rowptr[m_ti + 1];
    }
    for (; tile < tend; tile += stride) {
        const TileMeta md = tile_meta_from(tile, E, j, m_ti, m_tnext, m_rs, m_re);
        const int e = min(tile * 32 + j, E - 1);
        const int ti = md.ti;
        const float inv_deg = 1.0f / (float)(md.re - md.rs);
        const int so = (abl & 4) ? 32 : 1;
        const size_t row = (abl & 4) ? (size_t)tile * 4096 + lane * 4 : (size_t)e * 128 + 4 * half;
        const int tn = tile + stride < tend ? tile + stride : tile;       // clamped: the last iteration prefetches unused rows
        const int en = min(tn * 32 + j, E - 1);
        const int tin = tgt[en];                                           // next tile's target / next target: consumed mid-tile
        const int tnn = (tn * 32 + j + 1 < E && j < 31) ? tgt[tn * 32 + j + 1] : -2;
        const float* pan = A + (size_t)tin * 128 + 4 * half;
        const float* pbn = Bm + (size_t)src[en] * 128 + 4 * half;
        f32x16 p[4], x[4], acc[4], accc[4], gd[4];
        {   // dagg rows (consumed after GEMM 1)
            const float* pd = dagg + (size_t)ti * 128 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 d = *reinterpret_cast<const f32x4*>(pd + nt * 32 + 8 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) gd[nt][4 * g + c] = d[c];
                }
        }
        // p = A[i] + Bm[j], x = s1 = Swish(p)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = ga[kt][i] + gb[kt][i];
                p[kt][i] = v;
                x[kt][i] = act_swish(v);
            }
        // ---- GEMM 1: q = W2 s1 + b2
        {
            int boff = 0;
            asm volatile("" : "+v"(boff));                        // loop-variant as far as the compiler knows: no hoisting
            init_bias<4>(acc, bl + boff, half);
        }
        zero_tiles<4>(accc);
#if EB_INTERLEAVE
        gemm128<MODE>(wl, x, acc, accc, lane, md.e_ok ? s1_out + row : nullptr, 1.0f);
#else
        if (!(abl & 1)) store_lines(s1_out, x, tl, tile, E, lane, md.e_ok, row, so);
        gemm128<MODE>(wl, x, acc, accc, lane, nullptr, 1.0f);
#endif
        // ---- dq = dagg[i] / deg * Swish'(q)   (into x), per-edge power-of-two scale for the fp16 split
        float mx = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = gd[nt][i] * inv_deg * dswish(acc[nt][i]);
                x[nt][i] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const unsigned eb = (__float_as_uint(mx) >> 23) & 0xFFu;              // max in [2^(eb-127), 2^(eb-126))
        const float sc = __uint_as_float((254u - eb) << 23);                   // 2^(127-eb): scaled max in [1, 2)
        const float usc = __uint_as_float(eb << 23);                           // 2^(eb-127)  (0 for an all-zero row)
#if !EB_INTERLEAVE
        if (!(abl & 1)) store_lines(dq_out, x, tl, tile, E, lane, md.e_ok, row, so);
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[nt][i] *= sc;
        // next tile's Bm rows (scattered sources, long latency): in flight during GEMM 2 and the dp phase
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(pbn + kt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) gb[kt][4 * g + c] = b[c];
            }
        const int rsn = rowptr[tin], ren = rowptr[tin + 1];                 // next tile's row bounds, consumed at the next tile's top
        // ---- GEMM 2: ds1 = W2^T dq
        zero_tiles<4>(acc);
        zero_tiles<4>(accc);
#if EB_INTERLEAVE
        gemm128<MODE>(wt, x, acc, accc, lane, md.e_ok ? dq_out + row : nullptr, usc);   // stores the unscaled dq
#else
        gemm128<MODE>(wt, x, acc, accc, lane, nullptr, 1.0f);
#endif
        // next tile's A rows (a handful of distinct targets, cache hits): requested once GEMM 2 has freed its cross
        // accumulators, in flight during the dp phase
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pan + kt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) ga[kt][4 * g + c] = a[c];
            }
        // ---- dp = ds1 * Swish'(p)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kt][i] = acc[kt][i] * usc * dswish(p[kt][i]);
        // ---- store dp and form dA[i] = sum over the edges of target i, k tile by k tile (stores spread between the scans)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#if EB_LINE_STORES
            if (!(abl & 2) && so == 1)
                store_tile_lines_u(acc[kt], tl, dp_out + (size_t)__builtin_amdgcn_readfirstlane(tile) * 4096, kt * 32,
                                   min(32, E - __builtin_amdgcn_readfirstlane(tile) * 32), lane);
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (md.e_ok && !(abl & 2) && !(EB_LINE_STORES && so == 1)) {
                    const f32x4 o = {acc[kt][4 * g], acc[kt][4 * g + 1], acc[kt][4 * g + 2], acc[kt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dp_out + row + (kt * 32 + 8 * g) * so) = o;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[kt][4 * g + c] = seg_scan_sum(acc[kt][4 * g + c], md);
            }
        }
        store_run_results<4, false>(acc, md, dA, partials, half);
        m_ti = tin, m_tnext = tnn, m_rs = rsn, m_re = ren;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same backward with the weight gradient of message_net_2 formed IN the kernel: dW2 = dq^T s1 and db2 = colsum(dq) never see
// s1 / dq in HBM (8.8 GB written and 8.8 GB read back per hop at 8.6 M edges; k_mpnn_edge_bwd above + k_wgrad_split).
//
// The contraction of dW2 runs over EDGES, the lane dimension of every register tile this kernel holds, and a 128 x 128 fp32
// accumulator is 256 registers per wave - so the four waves of the workgroup (one per SIMD, in lockstep on the t-th tile of their
// ranges) each own a 32-row slab of dW2 (rows = features of dq; 64 accumulator registers) and see every wave's tile through LDS:
//   * what is left of the 160 KiB beside the two weight orientations (128 KiB) holds ONE tile's dq [128 features][32 edges] and HALF
//     of its s1 [64 features][32 edges], fp32, feature-major with a 36-float row stride: the publishing wave writes dwords (lane =
//     edge: conflict-free), every wave reads 16-byte runs of 8 consecutive edges of one feature (conflict-free for the b128 lane
//     groups) - i.e. its MFMA operands "lane = feature, registers = edges" come straight out of the transposition;
//   * the waves publish in turn (4 rounds per step, 4 barriers per round); a round's second s1 half is written while the others run
//     the MFMAs of the first half.  Only the publisher's STORES are conditional: values defined under `if (wave == s)` (a
//     parallel-staging variant with resident operands was tried first) made hipcc sink computations into the branches and spill
//     300-500 registers;
//   * arithmetic: the fp16 split with ONE accumulator (x y ~ hx hy + lx hy + hx ly, the low pieces UNSCALED fp16 - subnormal at
//     worst, the matrix cores keep subnormals), 3 v_mfma_f32_32x32x16_f16 per 16 edges and 32 x 32 block; the two GEMMs of the data
//     path use the same scheme (gemm128_single) because the cross accumulators of the two-accumulator form are the 64 registers the
//     dW2 slab needs.  dq ~ 1e-7 is outside fp16's range, so it is staged times G = 2^k, ONE power of two per launch chosen from
//     max|dagg| (k_absmax_partial; |dq| <= 1.1 max|dagg|, G max|dq| < 2^15), exact and divided out of the per-workgroup partial sums;
//   * register roles are rotated so that nothing is recomputed: s1 stays in x until it has been staged (GEMM 1 only reads it), dq is
//     formed in place in GEMM 1's accumulator, GEMM 2 accumulates into x, and Swish'(p) comes from the same sigmoid as s1 - one
//     sigmoid per element less than k_mpnn_edge_bwd.
// Per-workgroup partial sums [grid][128][128] / [grid][128] leave once at the end; k_reduce_partials adds them in workgroup order.
#define EBW_ES 36

__global__ void __launch_bounds__(256) k_absmax_partial(const float* __restrict__ x, long n4, float* __restrict__ part) {
    __shared__ float wm[4];
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}

__global__ void __launch_bounds__(256, 1) k_mpnn_edge_bwd_w(const float* __restrict__ A, const float* __restrict__ Bm,
                                                            const float* __restrict__ dagg, const int32_t* __restrict__ src,
                                                            const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr,
                                                            int E, const u32x4* __restrict__ w2s, const u32x4* __restrict__ w2ts,
                                                            const float* __restrict__ b2, float* __restrict__ dp_out,
                                                            float* __restrict__ dA, float* __restrict__ partials,
                                                            const float* __restrict__ amax_part, int n_amax,
                                                            float* __restrict__ wpart, float* __restrict__ bpart) {
    constexpr int NP = 2;
    constexpr int ES = EBW_ES;
    __shared__ __attribute__((aligned(16))) u32x4 wl[32 * NP * 64];          // W2   planes (2^11 hw, 2^11 lw), 64 KiB
    __shared__ __attribute__((aligned(16))) u32x4 wt[32 * NP * 64];          // W2^T planes, 64 KiB
    __shared__ __attribute__((aligned(16))) float bl[128];                   // 2^11 b2
    __shared__ __attribute__((aligned(16))) float sdq[128 * ES];             // one tile's G dq, [feature][edge]
    __shared__ __attribute__((aligned(16))) float ss1[64 * ES];              // half of its s1
    __shared__ float gsc[4];
    for (int i = threadIdx.x; i < 32 * NP * 64; i += 256) {                  // [kt][nt][m][plane][64 lanes]: plane = bit 6 of i
        const bool hi_plane = ((i >> 6) & 1) == 0;
        wl[i] = hi_plane ? scale_plane_2048(w2s[i]) : w2s[i];
        wt[i] = hi_plane ? scale_plane_2048(w2ts[i]) : w2ts[i];
    }
    if (threadIdx.x < 128) bl[threadIdx.x] = b2[threadIdx.x] * 2048.0f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the per-wave branches are s_cbranch)
    const int half = lane >> 5, j = lane & 31;
    float G, invG;
    {
        float m = 0.f;
        for (int i = threadIdx.x; i < n_amax; i += 256) m = fmaxf(m, amax_part[i]);
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) gsc[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(gsc[0], gsc[1]), fmaxf(gsc[2], gsc[3]));
        const unsigned eb = (__float_as_uint(m) >> 23) & 0xFFu;               // max|dagg| in [2^(eb-127), 2^(eb-126))
        const bool ok = eb >= 14u && eb < 255u;
        G = ok ? __uint_as_float((267u - eb) << 23) : 1.0f;                   // 2^(140-eb): G * 1.1 max|dagg| < 2^15
        invG = ok ? __uint_as_float((eb - 13u) << 23) : 1.0f;
    }
    const int ntiles = (E + 31) / 32;
    const int nb8 = gridDim.x / 8;
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    int tile = (vb * 4 + wave) * tpw;
    // every wave runs tpw steps (the barriers below are workgroup-wide); tiles past the end are all-invalid lanes: clamped loads,
    // zero contributions, no stores
    f32x16 ga[4], gb[4];
    {
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
                for (int c = 0; c < 4; ++c) {
                    ga[kt][4 * g + c] = a[c];
                    gb[kt][4 * g + c] = b[c];
                }
            }
    }
    int m_ti, m_tnext, m_rs, m_re;
    {
        const int e_raw = tile * 32 + j;
        m_ti = tgt[e_raw < E ? e_raw : E - 1];
        m_tnext = (e_raw + 1 < E && j < 31) ? tgt[e_raw + 1] : -2;
        m_rs = rowptr[m_ti];
        m_re = rowptr[m_ti + 1];
    }
    f32x16 accw[4];                                                           // G dW2[32 wave + row][32 b + column]
    zero_tiles<4>(accw);
    float bsum = 0.f;
    for (int step = 0; step < tpw; ++step, ++tile) {
        const TileMeta md = tile_meta_from(tile, E, j, m_ti, m_tnext, m_rs, m_re);
        const int e = min(tile * 32 + j, E - 1);
        const int ti = md.ti;
        const float inv_deg = md.e_ok ? 1.0f / (float)(md.re - md.rs) : 0.0f;   // (invalid lanes: dq = 0)
        const size_t row = (size_t)e * 128 + 4 * half;
        const int tn = tile + 1;
        const int en = min(tn * 32 + j, E - 1);
        const int tin = tgt[en];
        const int tnn = (tn * 32 + j + 1 < E && j < 31) ? tgt[min(tn * 32 + j + 1, E - 1)] : -2;
        const int sjn = src[en];                                           // (the row pointers are formed where the rows are requested)
        // register roles: p = A[i] + Bm[j], then Swish'(p) (both from ONE sigmoid, for the dp phase); x = s1, intact until it has
        // been staged for the weight gradient, then GEMM 2's accumulator; acc = GEMM 1's accumulator, then dq in place
        f32x16 p[4], x[4], acc[4], gd[4];
        {   // dagg rows (consumed after GEMM 1)
            const float* pd = dagg + (size_t)ti * 128 + 4 * half;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 d = *reinterpret_cast<const f32x4*>(pd + nt * 32 + 8 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) gd[nt][4 * g + c] = d[c];
                }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = ga[kt][i] + gb[kt][i], sg = sigmoid_fast(v);
                x[kt][i] = v * sg;
                p[kt][i] = sg * (1.0f + v * (1.0f - sg));
            }
            asm volatile("" : "+v"(p[kt]));                                   // formed here: sunk to the dp phase, v AND sg would stay alive
        }
        // ---- GEMM 1: 2^11 q = 2^11 (W2 s1 + b2)
        {
            int boff = 0;
            asm volatile("" : "+v"(boff));
            init_bias<4>(acc, bl + boff, half);
        }
        gemm128_single(wl, x, acc, lane);
        // ---- dq = dagg[i] / deg * Swish'(q), in place; then times the per-edge power of two of GEMM 2's fp16 split
        float mx = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = gd[nt][i] * inv_deg * dswish(acc[nt][i] * (1.0f / 2048.0f));
                acc[nt][i] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const unsigned eb = (__float_as_uint(mx) >> 23) & 0xFFu;
        const float sc = __uint_as_float((254u - eb) << 23);
        const float usc = __uint_as_float(eb << 23);
        // ---- dW2 += dq^T s1, db2 += colsum(dq): the four tiles of this step, one publisher at a time (its stores are the only
        // conditional code: no value is defined under a branch)
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            wg_barrier();                                                     // staging free: everybody has read the previous round
            // (the dq of publisher s > 0 is already there: staged during the second half of round s - 1, see below)
            if (wave == 0 && s == 0) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sdq[(nt * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = acc[nt][i] * G;
            }
            if (wave == s) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ss1[(nt * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = x[nt][i];
            }
            wg_barrier();
            f16x8 ah[2], al[2], xh[2][2], xl[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bsum += stage_operand(sdq + (wave * 32 + j) * ES + 16 * ks + 8 * half, ah[ks], al[ks]);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) stage_operand(ss1 + (b * 32 + j) * ES + 16 * ks + 8 * half, xh[b][ks], xl[b][ks]);
            wg_barrier();                                                     // first half of s1 consumed (it sits in registers)
            if (wave == s) {
#pragma unroll
                for (int nt = 2; nt < 4; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ss1[((nt - 2) * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = x[nt][i];
            }
            // everybody holds its dq operand in registers since the barrier above: the NEXT publisher stages its dq now, behind the
            // MFMAs of the others, instead of at the top of its own round where the whole workgroup waits for it
            if (wave == s + 1) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) sdq[(nt * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = acc[nt][i] * G;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                accw[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], xh[0][ks], accw[0], 0, 0, 0);
                accw[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], xh[1][ks], accw[1], 0, 0, 0);
                accw[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xl[0][ks], accw[0], 0, 0, 0);
                accw[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xl[1][ks], accw[1], 0, 0, 0);
                accw[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xh[0][ks], accw[0], 0, 0, 0);
                accw[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xh[1][ks], accw[1], 0, 0, 0);
            }
            wg_barrier();                                                     // second half staged
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) stage_operand(ss1 + (b * 32 + j) * ES + 16 * ks + 8 * half, xh[b][ks], xl[b][ks]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                accw[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], xh[0][ks], accw[2], 0, 0, 0);
                accw[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], xh[1][ks], accw[3], 0, 0, 0);
                accw[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xl[0][ks], accw[2], 0, 0, 0);
                accw[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xl[1][ks], accw[3], 0, 0, 0);
                accw[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xh[0][ks], accw[2], 0, 0, 0);
                accw[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], xh[1][ks], accw[3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] *= sc;
        // next tile's Bm rows (scattered sources, long latency): in flight during GEMM 2 and the dp phase
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(Bm + (size_t)sjn * 128 + 4 * half + kt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) gb[kt][4 * g + c] = b[c];
            }
        const int rsn = rowptr[tin], ren = rowptr[tin + 1];
        // ---- GEMM 2: 2^11 ds1 = 2^11 W2^T dq, accumulated in x
        zero_tiles<4>(x);
        gemm128_single(wt, acc, x, lane);
        // next tile's A rows (a handful of distinct targets, cache hits): in flight during the dp phase
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)tin * 128 + 4 * half + kt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) ga[kt][4 * g + c] = a[c];
            }
        // ---- dp = ds1 * Swish'(p)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[kt][i] = x[kt][i] * (1.0f / 2048.0f) * usc * p[kt][i];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (md.e_ok) {
                    const f32x4 o = {x[kt][4 * g], x[kt][4 * g + 1], x[kt][4 * g + 2], x[kt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dp_out + row + kt * 32 + 8 * g) = o;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) x[kt][4 * g + c] = seg_scan_sum(x[kt][4 * g + c], md);
            }
        }
        store_run_results<4, false>(x, md, dA, partials, half);
        m_ti = tin, m_tnext = tnn, m_rs = rsn, m_re = ren;
    }
    // ---- this workgroup's partial sums (C/D row = dq feature, column = s1 feature)
    float* wout = wpart + (size_t)blockIdx.x * 128 * 128;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            wout[(size_t)n * 128 + b * 32 + j] = accw[b][r] * invG;
        }
    bsum += __shfl_xor(bsum, 32);
    if (half == 0) bpart[(size_t)blockIdx.x * 128 + wave * 32 + j] = bsum * invG;
}

// combine the tile-boundary partial sums and zero the rows without edges
__global__ void __launch_bounds__(256) k_segsum_fixup(const int32_t* __restrict__ rowptr, int N, const float* __restrict__ partials,
                                                      float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= N) return;
    const int s = rowptr[r], e = rowptr[r + 1];
    if (e == s) {
        for (int f = lane; f < 128; f += 64) out[(size_t)r * 128 + f] = 0.f;
        return;
    }
    const int ta = s >> 5, tb = (e - 1) >> 5;
    if (ta == tb) return;
    for (int f = lane; f < 128; f += 64) {
        float acc = partials[((size_t)ta * 2 + 1) * 128 + f];
        for (int t = ta + 1; t <= tb; ++t) acc += partials[((size_t)t * 2 + 0) * 128 + f];
        out[(size_t)r * 128 + f] = acc;
    }
}

// generation 3 (k_mpnn_edge_bwd_co carries the open row from tile to tile): only the rows cut by a WORKGROUP's range of tpw tiles have
// pieces - partials[range][0] = the range's part of the row that entered it, [range][1] = of the row that leaves it (and did not enter
// it).  The wave of the first boundary a row crosses adds them up in range order; every thread also zeroes one row without edges.
__global__ void __launch_bounds__(256) k_segsum_fixup_wg(const int32_t* __restrict__ rowptr, int N, const int32_t* __restrict__ tgt, int E,
                                                         int tpw, int nwg, const float* __restrict__ partials, float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < N && rowptr[r] == rowptr[r + 1]) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < 128; f += 4) *reinterpret_cast<f32x4*>(out + (size_t)r * 128 + f) = z;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave + 1;                                 // the boundary in front of range b
    if (b >= nwg) return;
    const long eb = (long)b * tpw * 32;
    if (eb >= E) return;
    const int t = tgt[eb];
    const int s = rowptr[t], e = rowptr[t + 1];
    if (s >= eb) return;                                                     // the row starts with the range
    const int ws = (s >> 5) / tpw, we = ((e - 1) >> 5) / tpw;
    if (ws != b - 1) return;                                                 // (a row longer than a range: its first boundary does the work)
    for (int f = lane; f < 128; f += 64) {
        float acc = partials[((size_t)ws * 2 + 1) * 128 + f];
        for (int w = ws + 1; w <= we; ++w) acc += partials[((size_t)w * 2 + 0) * 128 + f];
        out[(size_t)t * 128 + f] = acc;
    }
}

// w2_f16 / w2t_f16: fp16 planes of W2 and of W2^T (magnet_amd/packing.py:pack_linear_fp16x2); partials: ceil(E/32)*2*128
// floats.  s1, dq, dp: [E][128]; dA: [n_nodes][128] (every row is written).
extern "C" int magnet_mpnn_edge_bwd_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src,
                                        const int32_t* tgt, const int32_t* rowptr, int n_nodes, int n_edges,
                                        const void* w2_f16, const void* w2t_f16, const float* b2, float* s1, float* dq,
                                        float* dp, float* dA, float* partials, void* stream) {
    (void)hipGetLastError();
    if (n_nodes <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_edges > 0) {
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
        const u32x4* w = reinterpret_cast<const u32x4*>(w2_f16);
        const u32x4* wT = reinterpret_cast<const u32x4*>(w2t_f16);
hipLaunchKernelGGL(k_mpnn_edge_bwd, g, b, 0, s, A, Bm, dagg, src, tgt, rowptr, n_edges, w, wT, b2, s1, dq, dp, dA, partials);
    }
    hipLaunchKernelGGL(k_segsum_fixup, dim3((n_nodes + 3) / 4), dim3(256), 0, s, rowptr, n_nodes, partials, dA);
    return (int)hipGetLastError();
}

int magnet_launch_absmax_partial(const float* x, long n4, float* part, int nblocks, void* stream) {
    hipLaunchKernelGGL(k_absmax_partial, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, x, n4, part);
    return (int)hipGetLastError();
}

// The same, with dW2 [128][128] (+)= dq^T s1 and db2 [128] (+)= colsum(dq) formed in the kernel (k_mpnn_edge_bwd_w): s1 and dq are
// never written.  workspace: magnet_mpnn_edge_bwd_w_workspace() floats (per-workgroup partial sums + the |dagg| maxima).
#define EBW_AMAX_BLOCKS 256
int magnet_launch_reduce_partials2(const float* pa, long na, float* oa, const float* pb, long nb_, float* ob, int nblk, float scale,
                                   int accumulate, void* stream);      // train_kernels.hip: both reductions in one launch
// mpnn_edge_bwd_wp.hip: generation 2 of the kernel (every wave forms the weight gradient of its own tiles; no barrier in the loop)
int magnet_launch_edge_bwd_wp(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                              const int32_t* rowptr, int n_edges, const void* w2_f16, const void* w2t_f16, const float* b2, float* dp,
                              float* dA, float* partials, const float* amax, int n_amax, float* wpart, float* bpart, float* pscr,
                              int grid, void* stream);
// mpnn_edge_bwd_co.hip: generation 3 (eight co-operating waves per tile, a wave owns 16 features; two waves per SIMD)
int magnet_launch_edge_bwd_co(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                              const int32_t* rowptr, int n_edges, const void* w2_f16, const void* w2t_f16, const float* b2, float* dp,
                              float* dA, float* partials, const float* amax, int n_amax, float* wpart, float* bpart, int ab_l2e, int grid,
                              void* stream);
// 2 (default): k_mpnn_edge_bwd_co; 1: k_mpnn_edge_bwd_wp (round 4); 0: k_mpnn_edge_bwd_w (the shared-slab kernel of round 3) - kept for
// A/B runs; same results up to the order of the fp32 additions
static int g_ebw_generation = 2;
extern "C" int magnet_mpnn_edge_bwd_w_generation(int generation) {
    const int was = g_ebw_generation;
    if (generation >= 0 && generation <= 2) g_ebw_generation = generation;
    return was;
}
extern "C" int magnet_mpnn_edge_bwd_w_workspace(void) {
    int dev, ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    // per-workgroup partial sums of dW2 and db2, the |dagg| maxima, and generation 2's parking block (16 KiB per wave)
    return ncu * (128 * 128 + 128) + EBW_AMAX_BLOCKS + ncu * 4 * 4096;
}
extern "C" int magnet_mpnn_edge_bwd_w_amax_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src,
                                               const int32_t* tgt, const int32_t* rowptr, int n_nodes, int n_edges,
                                               const void* w2_f16, const void* w2t_f16, const float* b2, float* dp, float* dA,
                                               float* partials, float* dW2, float* db2, int accumulate, float* workspace,
                                               const float* amax_in, int n_amax_in, void* stream);
extern "C" int magnet_mpnn_edge_bwd_w_flags_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src,
                                                const int32_t* tgt, const int32_t* rowptr, int n_nodes, int n_edges,
                                                const void* w2_f16, const void* w2t_f16, const float* b2, float* dp, float* dA,
                                                float* partials, float* dW2, float* db2, int accumulate, float* workspace,
                                                const float* amax_in, int n_amax_in, int flags, void* stream);
extern "C" int magnet_mpnn_edge_bwd_w_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src,
                                          const int32_t* tgt, const int32_t* rowptr, int n_nodes, int n_edges,
                                          const void* w2_f16, const void* w2t_f16, const float* b2, float* dp, float* dA,
                                          float* partials, float* dW2, float* db2, int accumulate, float* workspace, void* stream) {
    return magnet_mpnn_edge_bwd_w_amax_f32(A, Bm, dagg, src, tgt, rowptr, n_nodes, n_edges, w2_f16, w2t_f16, b2, dp, dA, partials, dW2, db2,
                                           accumulate, workspace, nullptr, 0, stream);
}
// amax_in != NULL: n_amax_in block maxima of |dagg| left by the kernel that produced it (magnet_linear_bwd128_f32's amax_out) - no
// maxima pass over dagg here
extern "C" int magnet_mpnn_edge_bwd_w_amax_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src,
                                               const int32_t* tgt, const int32_t* rowptr, int n_nodes, int n_edges,
                                               const void* w2_f16, const void* w2t_f16, const float* b2, float* dp, float* dA,
                                               float* partials, float* dW2, float* db2, int accumulate, float* workspace,
                                               const float* amax_in, int n_amax_in, void* stream) {
    return magnet_mpnn_edge_bwd_w_flags_f32(A, Bm, dagg, src, tgt, rowptr, n_nodes, n_edges, w2_f16, w2t_f16, b2, dp, dA, partials, dW2, db2,
                                            accumulate, workspace, amax_in, n_amax_in, 0, stream);
}
// flags bit 0 (MAGNET_EDGE_BWD_AB_LOG2E): A / Bm are the forward's log2(e)-scaled halves (magnet_mpnn_edge_stream_fp16x3_f32 with
// MAGNET_STREAM_L2E); dp / dA / dW2 / db2 are the gradients of the unscaled block all the same.  Generation 3 only (-2 otherwise:
// magnet_mpnn_edge_bwd_w_takes_flags says beforehand whether a graph of n_nodes would take it).
extern "C" int magnet_mpnn_edge_bwd_w_takes_flags(int n_nodes) {
    return g_ebw_generation == 2 && (size_t)n_nodes * 512 < ((size_t)1 << 32);
}
extern "C" int magnet_mpnn_edge_bwd_w_flags_f32(const float* A, const float* Bm, const float* dagg, const int32_t* src,
                                                const int32_t* tgt, const int32_t* rowptr, int n_nodes, int n_edges,
                                                const void* w2_f16, const void* w2t_f16, const float* b2, float* dp, float* dA,
                                                float* partials, float* dW2, float* db2, int accumulate, float* workspace,
                                                const float* amax_in, int n_amax_in, int flags, void* stream) {
    (void)hipGetLastError();
    if (flags & ~1) return -2;
    if ((flags & 1) && !magnet_mpnn_edge_bwd_w_takes_flags(n_nodes)) return -2;
    if (n_nodes <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_edges > 0) {
        static int ncu = 0;
        if (!ncu) {
            hipDeviceProp_t prop;
            int dev;
            hipGetDevice(&dev);
            hipGetDeviceProperties(&prop, dev);
            ncu = prop.multiProcessorCount;
        }
        const int tiles = (n_edges + 31) / 32;
        const int grid = (g_ebw_generation == 2 && (size_t)n_nodes * 512 < ((size_t)1 << 32)) ? min(tiles, ncu) : min((tiles + 3) / 4, ncu);
        float* wpart = workspace;
        float* bpart = wpart + (size_t)ncu * 128 * 128;
        float* amax = bpart + (size_t)ncu * 128;
        const long n4 = (long)n_nodes * 32;
        int nab = (int)min((long)EBW_AMAX_BLOCKS, (n4 + 255) / 256);
        if (amax_in) {
            amax = const_cast<float*>(amax_in);
            nab = n_amax_in;
        } else {
            hipLaunchKernelGGL(k_absmax_partial, dim3(nab), dim3(256), 0, s, dagg, n4, amax);
        }
        const u32x4* w = reinterpret_cast<const u32x4*>(w2_f16);
        const u32x4* wT = reinterpret_cast<const u32x4*>(w2t_f16);
        float* pscr = bpart + (size_t)ncu * 128 + EBW_AMAX_BLOCKS;
        // (generation 3 addresses the node rows with 32-bit byte offsets: graphs of 8 M nodes and more take generation 2)
        const bool co_ok = (size_t)n_nodes * 512 < ((size_t)1 << 32);
        if (g_ebw_generation == 2 && co_ok)
            magnet_launch_edge_bwd_co(A, Bm, dagg, src, tgt, rowptr, n_edges, w2_f16, w2t_f16, b2, dp, dA, partials, amax, nab, wpart, bpart,
                                      flags & 1, grid, stream);
        else if (g_ebw_generation >= 1)
            magnet_launch_edge_bwd_wp(A, Bm, dagg, src, tgt, rowptr, n_edges, w2_f16, w2t_f16, b2, dp, dA, partials, amax, nab, wpart, bpart,
                                      pscr, grid, stream);
        else
            hipLaunchKernelGGL(k_mpnn_edge_bwd_w, dim3(grid), dim3(256), 0, s, A, Bm, dagg, src, tgt, rowptr, n_edges, w, wT, b2, dp, dA,
                               partials, amax, nab, wpart, bpart);
        magnet_launch_reduce_partials2(wpart, 128L * 128, dW2, bpart, 128L, db2, grid, 1.0f, accumulate, stream);
        if (g_ebw_generation == 2 && co_ok) {
            const int tpw = (tiles + grid - 1) / grid;
            hipLaunchKernelGGL(k_segsum_fixup_wg, dim3(max((n_nodes + 255) / 256, (grid + 2) / 4)), dim3(256), 0, s, rowptr, n_nodes, tgt,
                               n_edges, tpw, grid, partials, dA);
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(k_segsum_fixup, dim3((n_nodes + 3) / 4), dim3(256), 0, s, rowptr, n_nodes, partials, dA);
    return (int)hipGetLastError();
}
