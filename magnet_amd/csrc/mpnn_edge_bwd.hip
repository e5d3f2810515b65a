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
