// Fused backward of the MPNN edge block with the weight gradient of message_net_2 formed by every wave FOR ITS OWN TILES
// (k_mpnn_edge_bwd_wp; generation 2 of k_mpnn_edge_bwd_w in mpnn_edge_bwd.hip, same arguments, same outputs).
// Reference: torch autograd through GNN_Layer.message / aggr='mean', models/mpnn.py:73-79,46.
//
// k_mpnn_edge_bwd_w shares dW2 between the four waves of a workgroup (a 32-row slab each) and hands every tile's dq and s1 round
// through LDS: 16 workgroup barriers per step and one publisher at a time - the LDS stores of the publisher alone are ~15 k of the
// ~40 k cycles a step takes, against 9.2 k cycles of matrix work.  Here a wave keeps the WHOLE [128 x 128] fp32 accumulator of dW2
// (16 MFMA tiles = 256 registers: the accumulator half of the unified 512-register file at one wave per SIMD) and nothing is handed
// between waves inside the loop: no barrier, no publisher, the waves drift freely.
//   * the data path runs in <= 256 registers because s1 never exists as fp32 beyond the moment it is formed (its two fp16 planes
//     ARE the operand of GEMM 1 and of the weight gradient), GEMM 1 and GEMM 2 run two output blocks at a time (the dagg rows of
//     the next pair are requested while the first pair multiplies; ds1 is turned into dp, stored and summed pair by pair);
//   * the contraction of dW2 runs over edges - the lane dimension of the data path - so each 32-feature block of s1 and of G dq goes
//     through a 4.5 KiB per-wave LDS block once: written as the plane words the data path already holds (lane = edge, one word =
//     two consecutive features: [16 feature pairs][32 edges], rows padded to 36 words), read back as 16-byte runs of 8 consecutive
//     edges of one feature pair, and a v_perm_b32 per word picks the lane's feature out of the pair.  A wave's DS operations
//     execute in order, so the block is reused without any synchronisation;
//   * db2 = colsum(dq) falls out of the transposed operand (lane = feature): v_dot2_f32_f16 with (1, 1) per plane word;
//   * the four waves add their accumulators through the LDS that held the weights, once, after the loop (fixed order).
// Arithmetic is k_mpnn_edge_bwd_w's: fp16 split with one accumulator, W planes (2^11 hw, 2^11 lw), G = one power of two per launch
// for the staged dq, a per-edge power of two for GEMM 2.
#include "common.cuh"
#include "edge_tile.cuh"
#include "split_single.cuh"
#include "wgrad_private.cuh"

#ifndef EB_SPLIT_TWICE
#define EB_SPLIT_TWICE 0           // 1: round 4's per-edge power of two for GEMM 2's operand (a second split of dq)
#endif

namespace {

__device__ __forceinline__ float sigmoid_wp(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float dswish_wp(float x) {
    const float s = sigmoid_wp(x);
    return s * (1.0f + x * (1.0f - s));
}

}  // namespace

__global__ void __launch_bounds__(256, 1) k_mpnn_edge_bwd_wp(const float* __restrict__ A, const float* __restrict__ Bm,
                                                             const float* __restrict__ dagg, const int32_t* __restrict__ src,
                                                             const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr,
                                                             int E, const u32x4* __restrict__ w2s, const u32x4* __restrict__ w2ts,
                                                             const float* __restrict__ b2, float* __restrict__ dp_out,
                                                             float* __restrict__ dA, float* __restrict__ partials,
                                                             const float* __restrict__ amax_part, int n_amax,
                                                             float* __restrict__ wpart, float* __restrict__ bpart,
                                                             float* __restrict__ pscr) {
    __shared__ __attribute__((aligned(16))) u32x4 wl[32 * 2 * 64];           // W2   planes (2^11 hw, 2^11 lw), 64 KiB
    __shared__ __attribute__((aligned(16))) u32x4 wt[32 * 2 * 64];           // W2^T planes, 64 KiB
    __shared__ __attribute__((aligned(16))) float bl[128];                   // 2^11 b2
    __shared__ __attribute__((aligned(16))) unsigned scr[4][WP_SCR];         // per-wave transposition block
    __shared__ float gsc[4];
    for (int i = threadIdx.x; i < 32 * 2 * 64; i += 256) {                   // [kt][nt][m][plane][64 lanes]: plane = bit 6 of i
        const bool hi_plane = ((i >> 6) & 1) == 0;
        wl[i] = hi_plane ? scale_plane_2048(w2s[i]) : w2s[i];
        wt[i] = hi_plane ? scale_plane_2048(w2ts[i]) : w2ts[i];
    }
    if (threadIdx.x < 128) bl[threadIdx.x] = b2[threadIdx.x] * 2048.0f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const int nsteps = __builtin_amdgcn_readfirstlane(max(0, min(tpw, ntiles - tile)));
    unsigned* const sw = scr[wave] + (2 * half) * WP_RS + j;
    const unsigned* const sr = scr[wave] + (j >> 1) * WP_RS + 8 * half;
    const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
    // Swish'(p) of the current tile waits in a 16 KiB per-wave block of `pscr` between its sigmoid and the dp phase ([16][64 lanes]
    // 16-byte pieces: every store / load instruction moves 1 KiB of consecutive addresses; 16 MiB for the whole grid, rewritten every
    // tile: L2-resident) - 64 registers that the phases in between need
    float* const park0 = pscr + ((size_t)blockIdx.x * 4 + wave) * 4096;          // (wave-uniform)
    const unsigned lane16 = 16u * (unsigned)lane;
    // every per-lane address below is a wave-uniform base (SGPR pair) + a 32-bit byte offset: 64-bit per-lane pointers cost two
    // registers each and hipcc keeps a dozen of them alive across the whole loop body
    const unsigned lrow = (unsigned)j * 512u + 16u * (unsigned)half;          // this lane's place in a [32 rows][128] fp32 block
    f32x16 ga[4], gb[4];
    auto load_rows = [&](f32x16 (&dst)[4], const float* __restrict__ base, int r) {
        const unsigned off = (unsigned)r * 512u + 16u * (unsigned)half;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = ldg4(base + kt * 32 + 8 * g, off);
#pragma unroll
                for (int c = 0; c < 4; ++c) dst[kt][4 * g + c] = v[c];
            }
    };
    int m_ti, m_sj;
    {
        const int e = min(tile * 32 + j, E - 1);
        m_ti = tgt[e];
        m_sj = src[e];
        load_rows(ga, A, m_ti);
        load_rows(gb, Bm, m_sj);
    }
    f32x16 accw[4][4];                                                        // G dW2[32 a + row][32 b + column]
#pragma unroll
    for (int a = 0; a < 4; ++a) zero_tiles<4>(accw[a]);
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};                                     // G db2[32 a + j], this lane's edges
    for (int step = 0; step < nsteps; ++step, ++tile) {
        const int e_raw = tile * 32 + j;
        const bool e_ok = e_raw < E;
        const int ti = m_ti;
        float* park = park0;
        asm volatile("" : "+s"(park));                                        // (park + k KiB formed by scalar adds here, not 16 hoisted 64-bit lane addresses)
        const int rs = rowptr[ti], re = rowptr[ti + 1];
        const int en = min(e_raw + 32, E - 1);
        const int tin = tgt[en], sjn = src[en];                               // next tile (rows requested further down)
        const unsigned doff = (unsigned)ti * 512u + 16u * (unsigned)half;
        f32x16 gd[2];
        auto load_gd = [&](int pr) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 d = ldg4(dagg + (2 * pr + b) * 32 + 8 * g, doff);
#pragma unroll
                    for (int c = 0; c < 4; ++c) gd[b][4 * g + c] = d[c];
                }
        };
        load_gd(0);
        // ---- s1 = Swish(p) as its two fp16 planes (slot = kt * 2 + m); Swish'(p) from the same sigmoid, parked for the dp phase
        u32x4 xh[8], xl[8];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x16 pp;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float v0 = ga[kt][2 * w] + gb[kt][2 * w], v1 = ga[kt][2 * w + 1] + gb[kt][2 * w + 1];
                const float s0 = sigmoid_wp(v0), s1 = sigmoid_wp(v1);
                pp[2 * w] = s0 * (1.0f + v0 * (1.0f - s0));
                pp[2 * w + 1] = s1 * (1.0f + v1 * (1.0f - s1));
                unsigned h, l;
                split_f16_prod(v0, s0, v1, s1, h, l);
                xh[2 * kt + (w >> 2)][w & 3] = h;
                xl[2 * kt + (w >> 2)][w & 3] = l;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {pp[4 * g], pp[4 * g + 1], pp[4 * g + 2], pp[4 * g + 3]};
                stg4(park + (kt * 4 + g) * 256, lane16, o);
            }
        }
        const float inv_deg = e_ok ? 1.0f / (float)(re - rs) : 0.0f;          // (invalid lanes: dq = 0)
        // one split: 1 / deg rides in the split's own multiplier (Gd = G / deg; acc holds dagg Swish'(q) only) - 64 multiplies less
        const float dq_inv = EB_SPLIT_TWICE ? inv_deg : 1.0f;
        const float Gd = EB_SPLIT_TWICE ? G : G * inv_deg;
        // ---- GEMM 1 (2^11 q = 2^11 (W2 s1 + b2)) and dq = dagg[i] / deg * Swish'(q), two output blocks at a time
        f32x16 acc[4];
        float mx = 0.f;
        {
            int boff = 0;
            asm volatile("" : "+v"(boff));
            init_bias<2>(reinterpret_cast<f32x16(&)[2]>(acc[0]), bl + boff, half);
        }
        gemm_pair_planes<0>(wl, xh, xl, acc[0], acc[1], lane);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = gd[b][i] * dq_inv * dswish_wp(acc[b][i] * (1.0f / 2048.0f));
                acc[b][i] = v;
                if (EB_SPLIT_TWICE) mx = fmaxf(mx, fabsf(v));
            }
        load_gd(1);
        {
            int boff = 64;
            asm volatile("" : "+v"(boff));
            init_bias<2>(reinterpret_cast<f32x16(&)[2]>(acc[2]), bl + boff, half);
        }
        gemm_pair_planes<1>(wl, xh, xl, acc[2], acc[3], lane);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = gd[b][i] * dq_inv * dswish_wp(acc[2 + b][i] * (1.0f / 2048.0f));
                acc[2 + b][i] = v;
                if (EB_SPLIT_TWICE) mx = fmaxf(mx, fabsf(v));
            }
        // GEMM 2's operand scale.  Round 5: the launch-wide power of two G (the scale of the weight gradient's operand) serves GEMM 2 as
        // well - ONE split of dq instead of two (-294 of the 2 734 plain VALU instructions of a tile: the second split, its 64 multiplies
        // and the per-edge maximum), as k_linear_bwd_r does since round 4.  An edge 1e5 below the launch maximum keeps ~7e-7 instead of
        // 2.6e-7 relative in its ds1; its share of dA / dBm is 1e-5 of the largest.  EB_SPLIT_TWICE=1: the per-edge scale.
        float sc, usc;
        if (EB_SPLIT_TWICE) {
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const unsigned eb = (__float_as_uint(mx) >> 23) & 0xFFu;
            sc = __uint_as_float((254u - eb) << 23);
            usc = __uint_as_float(eb << 23) * (1.0f / 2048.0f);
        } else {
            sc = G;
            usc = invG * (1.0f / 2048.0f);
        }
        // ---- dW2 += (G dq)^T s1, db2 += colsum(G dq): s1 through the block first (its planes die here), then dq block by block
        u32x4 Bh[4][2], Bl[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            wp_write_block(sw, xh[2 * b], xh[2 * b + 1], xl[2 * b], xl[2 * b + 1]);
            wp_read_block(sr, sel, Bh[b], Bl[b]);
        }
        u32x4 dh[8], dl[8];                                                   // the planes of sc dq: GEMM 2's operand
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u32x4 th[2], tl[2];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                unsigned h, l;
                split_f16_prod(acc[a][2 * w], Gd, acc[a][2 * w + 1], Gd, h, l);
                th[w >> 2][w & 3] = h;
                tl[w >> 2][w & 3] = l;
                if (EB_SPLIT_TWICE) split_f16_plain(acc[a][2 * w] * sc, acc[a][2 * w + 1] * sc, h, l);
                dh[2 * a + (w >> 2)][w & 3] = h;
                dl[2 * a + (w >> 2)][w & 3] = l;
            }
            wp_write_block(sw, th[0], th[1], tl[0], tl[1]);
            u32x4 Ah[2], Al[2];
            wp_read_block(sr, sel, Ah, Al);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum[a] = dot_ones(Al[ks][q], dot_ones(Ah[ks][q], bsum[a]));
            asm volatile("" : "+v"(bsum[a]));                                 // (formed HERE: sunk to the end of the body, the operands of all four blocks stay alive)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 ah = __builtin_bit_cast(f16x8, Ah[ks]), al = __builtin_bit_cast(f16x8, Al[ks]);
#pragma unroll
                for (int b = 0; b < 4; ++b) mfma_a(accw[a][b], al, __builtin_bit_cast(f16x8, Bh[b][ks]));
#pragma unroll
                for (int b = 0; b < 4; ++b) mfma_a(accw[a][b], ah, __builtin_bit_cast(f16x8, Bl[b][ks]));
#pragma unroll
                for (int b = 0; b < 4; ++b) mfma_a(accw[a][b], ah, __builtin_bit_cast(f16x8, Bh[b][ks]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- GEMM 2 (2^11 sc ds1 = 2^11 W2^T sc dq), dp = ds1 * Swish'(p), dA[i] = sum over the row: two blocks at a time
        const int tnext = (e_ok && e_raw + 1 < E && j < 31) ? tgt[e_raw + 1] : -2;
        const TileMeta md = tile_meta_from(tile, E, j, ti, tnext, rs, re);
        float* const dp_tile = dp_out + (size_t)__builtin_amdgcn_readfirstlane(tile) * 4096;
        unsigned lane16u = lane16;
        asm volatile("" : "+v"(lane16u));                                     // (not the address hipcc saw stored to: a real load)
        f32x16 p[2];
        auto load_p = [&](int pk) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 o = ldg4(park + ((2 * pk + b) * 4 + g) * 256, lane16u);
#pragma unroll
                    for (int c = 0; c < 4; ++c) p[b][4 * g + c] = o[c];
                }
        };
        auto dp_store = [&](int pk, f32x16 (&x2)[2]) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int i = 0; i < 16; ++i) x2[b][i] = x2[b][i] * usc * p[b][i];
                if (e_ok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {x2[b][4 * g], x2[b][4 * g + 1], x2[b][4 * g + 2], x2[b][4 * g + 3]};
                        stg4(dp_tile + (2 * pk + b) * 32 + 8 * g, lrow, o);
                    }
                }
            }
        };
        auto dp_runs = [&](int pk, f32x16 (&x2)[2]) {
            if (e_ok && tnext != ti) {                                        // the last edge of a run holds the run's sum
                const int e0 = tile * 32;
                const bool complete = (rs >= e0) && (re <= e0 + 32);
                float* const dst = (complete ? dA + (size_t)ti * 128 : partials + ((size_t)tile * 2 + (rs < e0 ? 0 : 1)) * 128) + 4 * half;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {x2[b][4 * g], x2[b][4 * g + 1], x2[b][4 * g + 2], x2[b][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(dst + (2 * pk + b) * 32 + 8 * g) = o;
                    }
            }
        };
        load_p(0);
        {
            f32x16 x0[2], x2[2];
            zero_tiles<2>(x0);
            gemm_pair_planes<0>(wt, dh, dl, x0[0], x0[1], lane);
            dp_store(0, x0);
            load_p(1);
            zero_tiles<2>(x2);
            // the row sums of the first pair (320 DPP / fmac instructions) behind the MFMAs of the second
            gemm_pair_planes<1>(wt, dh, dl, x2[0], x2[1], lane, [&](int i) {
#pragma unroll
                for (int v = 0; v < 32; ++v)
                    if (v * 48 / 32 == i) x0[v >> 4][v & 15] = seg_scan_sum(x0[v >> 4][v & 15], md);
            });
            dp_runs(0, x0);
            dp_store(1, x2);
            __builtin_amdgcn_sched_barrier(0);
            // next tile's rows: Bm by source (scattered, L2), A by target (a handful of distinct rows) - requested once Swish'(p) and
            // the operand planes of GEMM 2 are dead, in flight during the row sums
            load_rows(gb, Bm, sjn);
            load_rows(ga, A, tin);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) x2[b][i] = seg_scan_sum(x2[b][i], md);
            dp_runs(1, x2);
        }
        m_ti = tin;
    }
    // ---- the workgroup's partial sums: the four waves add up through the LDS that held the weights (fixed order), then whole lines
    asm volatile("s_nop 10" ::: "memory");                        // (the last MFMA's result, see mfma_results_ready)
    __syncthreads();
    float* const red = reinterpret_cast<float*>(wl);                          // [128][128]
    float* const redb = reinterpret_cast<float*>(wt);                         // [128]
#pragma unroll
    for (int a = 0; a < 4; ++a) bsum[a] += __shfl_xor(bsum[a], 32);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        if (wave == s) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int idx = (a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 128 + b * 32 + j;
                        red[idx] = (s == 0 ? 0.f : red[idx]) + accw[a][b][r];
                    }
                if (half == 0) redb[a * 32 + j] = (s == 0 ? 0.f : redb[a * 32 + j]) + bsum[a];
            }
        }
        __syncthreads();
    }
    float* wout = wpart + (size_t)blockIdx.x * 128 * 128;
    for (int i = threadIdx.x; i < 128 * 128 / 4; i += 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(red)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] *= invG;
        reinterpret_cast<f32x4*>(wout)[i] = v;
    }
    if (threadIdx.x < 128) bpart[(size_t)blockIdx.x * 128 + threadIdx.x] = redb[threadIdx.x] * invG;
}

int magnet_launch_edge_bwd_wp(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                              const int32_t* rowptr, int n_edges, const void* w2_f16, const void* w2t_f16, const float* b2, float* dp,
                              float* dA, float* partials, const float* amax, int n_amax, float* wpart, float* bpart, float* pscr,
                              int grid, void* stream) {
    hipLaunchKernelGGL(k_mpnn_edge_bwd_wp, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, Bm, dagg, src, tgt, rowptr, n_edges,
                       reinterpret_cast<const u32x4*>(w2_f16), reinterpret_cast<const u32x4*>(w2t_f16), b2, dp, dA, partials, amax, n_amax,
                       wpart, bpart, pscr);
    return (int)hipGetLastError();
}
