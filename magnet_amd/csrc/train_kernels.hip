// Backward-pass building blocks for the MPNN training step (reference autograd through
// models/mpnn.py:175-203 and the rollout of :254-297).  First generation: correct and deterministic, tensors at
// edge granularity are materialised (the fused forward kernels stay in use for the forward half of the step).
#include <cstdlib>
#include "common.cuh"

__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Elementwise kernels: thread i owns elements 4i..4i+3 (one float4 when whole, scalars at the tail).
#define EW_BODY(EXPR_VEC_DECL, EXPR)                                                                  \
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;                                \
    if (i4 >= n) return;                                                                              \
    const int cnt = (n - i4) < 4 ? (int)(n - i4) : 4;

__device__ __forceinline__ f32x4 ld4(const float* p, long i4, int cnt) {
    if (cnt == 4) return *reinterpret_cast<const f32x4*>(p + i4);
    f32x4 v = {0, 0, 0, 0};
    for (int c = 0; c < cnt; ++c) v[c] = p[i4 + c];
    return v;
}
__device__ __forceinline__ void st4(float* p, long i4, int cnt, f32x4 v) {
    if (cnt == 4) { *reinterpret_cast<f32x4*>(p + i4) = v; return; }
    for (int c = 0; c < cnt; ++c) p[i4 + c] = v[c];
}
// r = z * sigmoid(z)
__global__ void k_swish_fwd(const float* __restrict__ z, float* __restrict__ r, long n) {
    EW_BODY(, )
    const f32x4 v = ld4(z, i4, cnt);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = act_swish(v[c]);
    st4(r, i4, cnt, o);
}
// dz = dr * d/dz (z sigmoid(z)) = dr * s * (1 + z * (1 - s))
__global__ void k_swish_bwd(const float* __restrict__ dr, const float* __restrict__ z, float* __restrict__ dz, long n) {
    EW_BODY(, )
    const f32x4 g = ld4(dr, i4, cnt), v = ld4(z, i4, cnt);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float s = sigmoidf_fast(v[c]);
        o[c] = g[c] * s * (1.0f + v[c] * (1.0f - s));
    }
    st4(dz, i4, cnt, o);
}
// the same, plus amax[blockIdx.x & 255] = max(..., |dz| of this workgroup): the block maxima magnet_linear_bwd128_f32 wants as amax_in
// (256 floats zeroed by the caller; a maximum does not depend on the order of the atomics: deterministic)
__global__ void __launch_bounds__(256) k_swish_bwd_amax(const float* __restrict__ dr, const float* __restrict__ z, float* __restrict__ dz,
                                                        long n, float* __restrict__ amax) {
    __shared__ float wm[4];
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int cnt = i4 >= n ? 0 : ((n - i4) < 4 ? (int)(n - i4) : 4);
    float m = 0.f;
    if (cnt) {
        const f32x4 g = ld4(dr, i4, cnt), v = ld4(z, i4, cnt);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s = sigmoidf_fast(v[c]);
            o[c] = g[c] * s * (1.0f + v[c] * (1.0f - s));
            if (c < cnt) m = fmaxf(m, fabsf(o[c]));
        }
        st4(dz, i4, cnt, o);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(reinterpret_cast<unsigned*>(amax) + (blockIdx.x & 255), __float_as_uint(m));      // (non-negative floats order as their bits)
    }
}
// out = a + alpha * b   (out may alias a)
__global__ void k_axpy(const float* a, const float* b, float alpha, float* out, long n) {
    EW_BODY(, )
    const f32x4 x = ld4(a, i4, cnt), y = ld4(b, i4, cnt);
    st4(out, i4, cnt, x + alpha * y);
}
__global__ void k_mul(const float* a, const float* b, float* out, long n) {
    EW_BODY(, )
    st4(out, i4, cnt, ld4(a, i4, cnt) * ld4(b, i4, cnt));
}
// dst[r, dcol : dcol+w] (+)= src[r, scol : scol+w]
__global__ void k_copy_cols(const float* __restrict__ src, int lds_, int scol, float* __restrict__ dst, int ldd, int dcol,
                            int w, long rows, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const long r = i / w;
    const int c = (int)(i % w);
    const float v = src[r * lds_ + scol + c];
    float* d = dst + r * ldd + dcol + c;
    *d = accumulate ? *d + v : v;
}

// dst[r, dcol : dcol+w] (+)= scale * src[r, scol : scol+w] with 16-byte accesses (w, both column offsets and both row strides multiples
// of 4): the gradient of a dense, scaled k-tile source (e.g. MAgNet's 2^k e: models/magnet_gnn.py:76-77 under Q2)
__global__ void k_copy_cols4(const float* __restrict__ src, int lds_, int scol, float* __restrict__ dst, int ldd, int dcol, int w4,
                             long rows, float scale, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w4) return;
    const long r = i / w4;
    const int c = (int)(i % w4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds_ + scol + c) * scale;
    float* d = dst + r * ldd + dcol + c;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(d);
    *reinterpret_cast<f32x4*>(d) = v;
}

// p[e] = A[tgt[e]] + Bm[src[e]]                                  (recomputed pre-activation of message_net_1)
__global__ void k_edge_pre(const float* __restrict__ A, const float* __restrict__ Bm, const int32_t* __restrict__ src,
                           const int32_t* __restrict__ tgt, long E, float* __restrict__ p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of one edge
    if (i >= E * 32) return;
    const long e = i >> 5;
    const int c = (int)(i & 31) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)tgt[e] * 128 + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(Bm + (size_t)src[e] * 128 + c);
    *reinterpret_cast<f32x4*>(p + e * 128 + c) = a + b;
}
// dm[e] = dagg[tgt[e]] / deg(tgt[e])                             (backward of the CSR mean)
__global__ void k_edge_dm(const float* __restrict__ dagg, const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr,
                          long E, float* __restrict__ dm) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * 32) return;
    const long e = i >> 5;
    const int c = (int)(i & 31) * 4;
    const int t = tgt[e];
    const float inv = 1.0f / (float)(rowptr[t + 1] - rowptr[t]);
    *reinterpret_cast<f32x4*>(dm + e * 128 + c) = *reinterpret_cast<const f32x4*>(dagg + (size_t)t * 128 + c) * inv;
}

// out[r] = sum_{k in [rowptr[r], rowptr[r+1])} x[perm ? perm[k] : k]      (H = 128)
__global__ void __launch_bounds__(256) k_segment_sum128(const float* __restrict__ x, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ perm, int N, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int s = rowptr[r], e = rowptr[r + 1];
    const int sub = lane >> 5, col = (lane & 31) * 4;
    f32x4 acc = {0, 0, 0, 0};
    // The permuted positions of up to 64 edges of the row are fetched by ONE load (lane i: edge base + i) and handed out with
    // ds_bpermute, so that the row loads of a half-wave do not wait for an index load each (round 5: a dependent pair of loads per
    // 512 bytes kept this kernel at 4.8 TB/s).  Same edges per half-wave in the same order as before: same bits.
    for (int base = s; base < e; base += 64) {
        const int cnt = min(64, e - base);
        const int mine = lane < cnt ? (perm ? perm[base + lane] : base + lane) : 0;
        for (int k0 = 0; k0 < cnt; k0 += 8) {                    // four edges per half-wave in flight (uniform trip count: all lanes shuffle)
            f32x4 v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k0 + sub + 2 * u;
                ok[u] = kk < cnt;
                const long row = __shfl(mine, ok[u] ? kk : 0);
                v[u] = *reinterpret_cast<const f32x4*>(x + row * 128 + col);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += ok[u] ? v[u][c] : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], 32);
    if (sub == 0) *reinterpret_cast<f32x4*>(out + (size_t)r * 128 + col) = acc;
}

// ---- weight gradient: partial[b][n][k] = sum over the row slabs of workgroup b of dY[m][n] * X[m][k]
// (+ column sums of dY for the bias).  v_mfma_f32_32x32x2_f32 with BOTH operands read as rows from LDS:
// A[i = n][kk = m] = dY[m][n0 + i], B[kk = m][j] = X[m][k0 + j] - consecutive lanes read consecutive floats, no
// transposes anywhere.  N = 128 (4 waves <-> 4 n tiles), KC = KT*32 columns of X.
// Slabs of 32 rows are DMA'd straight into a two-stage LDS ring (global_load_lds, 1 KiB per wave instruction) while
// the previous slab is consumed; two workgroups per CU.  HBM-bound at the edge level: 2 x 4 B per (row, column).
#define WG_SLAB 32
template <int KT>
__device__ __forceinline__ void wgrad_dma(const float* __restrict__ dY, int ldy, int N, const float* __restrict__ X, int ldx, int xcol,
                                          long m0, long M, float* sy, float* sx, int wave, int lane) {
    constexpr int KC = KT * 32;
    const int npy = N / 8;                                              // dY slab: 32 * N / 256 pieces of 1 KiB
    for (int pi = wave; pi < npy; pi += 4) {
        const int f = pi * 256 + lane * 4;
        long r = m0 + f / N;
        r = r < M ? r : M - 1;                                          // tail rows: valid address, masked at use
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(dY + r * ldy + f % N),
                                         reinterpret_cast<u32x4*>(sy + pi * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < KT; ++i) {                                     // X slab: 32*KC/256 = 4 KT pieces
        const int pi = wave + 4 * i;
        const int f = pi * 256 + lane * 4;
        long r = m0 + f / KC;
        r = r < M ? r : M - 1;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32x4*>(X + r * ldx + xcol + f % KC),
                                         reinterpret_cast<u32x4*>(sx + pi * 256), 16, 0, 0);
    }
}

template <int KT>
__global__ void __launch_bounds__(256, 2) k_wgrad_partial(const float* __restrict__ dY, int ldy, int N, const float* __restrict__ X,
                                                          int ldx, int xcol, long M, float* __restrict__ partial /* [grid][N][KT*32] */,
                                                          float* __restrict__ bpartial /* [grid][N] or null */) {
    constexpr int KC = KT * 32;
    __shared__ __attribute__((aligned(16))) float sy[2][WG_SLAB * 128];
    __shared__ __attribute__((aligned(16))) float sx[2][WG_SLAB * KC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    const bool active = wave * 32 < N;                                  // N = 32, 64 or 128 columns of dY: one n tile per wave
    f32x16 acc[KT];
    zero_tiles<KT>(acc);
    float bsum = 0.f;
    const long nslabs = (M + WG_SLAB - 1) / WG_SLAB;
    long sl = blockIdx.x;
    if (sl < nslabs) wgrad_dma<KT>(dY, ldy, N, X, ldx, xcol, sl * WG_SLAB, M, sy[0], sx[0], wave, lane);
    for (int it = 0; sl < nslabs; sl += gridDim.x, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // my pieces of slab `it` have landed
        __syncthreads();                                                // everybody's have; stage (it+1)&1 is free again
        const long nxt = sl + gridDim.x;
        if (nxt < nslabs) wgrad_dma<KT>(dY, ldy, N, X, ldx, xcol, nxt * WG_SLAB, M, sy[(it + 1) & 1], sx[(it + 1) & 1], wave, lane);
        const float* cy = sy[it & 1];
        const float* cx = sx[it & 1];
        const long m0 = sl * WG_SLAB;
        if (active) {
#pragma unroll 4
            for (int s = 0; s < WG_SLAB / 2; ++s) {
                const int m = 2 * s + half;
                float a = cy[m * N + wave * 32 + i];
                a = (m0 + m < M) ? a : 0.f;
                bsum += a;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) acc[kt] = mfma32(a, cx[m * KC + kt * 32 + i], acc[kt]);
            }
        }
    }
    if (!active) return;
    float* out = partial + (size_t)blockIdx.x * N * KC;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // C/D row = A's i = n
            out[(size_t)n * KC + kt * 32 + i] = acc[kt][r];                    // C/D col = B's j = k
        }
    if (bpartial) {
        bsum += __shfl_xor(bsum, 32);
        if (half == 0) bpartial[(size_t)blockIdx.x * N + wave * 32 + i] = bsum;
    }
}
// ---- the same weight gradient on the bf16 matrix cores through the exact three-piece split (common.cuh "bf16x6": six partial
// products, fp32-class accuracy, full fp32 exponent range - gradients of a mean loss are ~1e-7).  The fp32-MFMA kernel above needs
// 256 x 64 = 16 384 matrix-pipe cycles per 32-row slab and wave (the E-level launches were matrix-pipe bound: 2.06 of their 2.8 ms);
// here a slab costs KT x 2 x 6 = 48 MFMA 32x32x16 = 1 536 cycles.  The contraction runs over ROWS m, so both operands must hold 8
// consecutive rows of one column per lane (A[i = n][k = m], B[k = m][j = column of X]):
//   * A (this wave's 32 columns of dY) never leaves the wave: lane (n, h) loads dY[m0 + 16 ks + 8 h + c][n0 + n] (c = 0..7, ks = 0, 1:
//     16 dword loads, 128-byte coalesced per row), splits them into the three pieces in registers;
//   * B (the X slab) is shared by the four waves: wave w loads and splits the 32 columns of k tile w the same way and writes the
//     pieces to LDS in fragment order [stage][ks][piece][h][column][8 rows] (16-byte writes and reads, conflict-free); one barrier
//     per slab, two stages;
//   * the loads of slab s + 1 are issued before the MFMAs of slab s (32 staging registers).
// KT2 = 1: one more 32-column k tile from a SECOND array X2 (the MPNN node-level Linears read [h | node_in]: the pair of launches that
// took dY^T h and dY^T node_in read - and split - dY twice); the tile is converted by wave 0 beside its own, partial is [grid][N][(KT+1)*32].
template <int KT, int KT2 = 0>
__global__ void __launch_bounds__(256, KT2 ? 2 : 3) k_wgrad_split(const float* __restrict__ dY, int ldy, int N, const float* __restrict__ X,
                                                        int ldx, int xcol, long M, float* __restrict__ partial /* [grid][N][KT*32] */,
                                                        float* __restrict__ bpartial /* [grid][N] or null */,
                                                        const float* __restrict__ X2 = nullptr, int ldx2 = 0, int x2col = 0) {
    constexpr int KTT = KT + KT2;
    constexpr int KC = KTT * 32;
    __shared__ __attribute__((aligned(16))) u32x4 pb[2][2 * 3 * 2 * KC];        // [stage][ks][piece][h][column]: 12 KiB per k tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    // gridDim.y > 1 (magnet_wgrad_tiles_f32): workgroup row y takes the column tile X[:, xcol + y KC ..) and its own partials
    xcol += (int)blockIdx.y * KC;
    partial += (size_t)blockIdx.y * gridDim.x * N * KC;
    if (blockIdx.y) bpartial = nullptr;
    const bool active = wave * 32 < N;                                  // one 32-column n tile of dY per wave
    const bool conv = wave < KT;                                        // ... and one k tile of X to convert
    const bool conv2 = KT2 && wave == 0;                                // (... and the tile of X2)
    f32x16 acc[KTT];
    zero_tiles<KTT>(acc);
    float bsum = 0.f;
    const long nslabs = (M + WG_SLAB - 1) / WG_SLAB;
    float ra[16], rb[16];                                               // raw values of the slab about to be consumed
    float rc[KT2 ? 16 : 1];
    auto load = [&](long sl) {
        const long m0 = sl * WG_SLAB;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long m = m0 + 16 * (q >> 3) + 8 * half + (q & 7);
            const long mc = m < M ? m : M - 1;
            const float va = active ? dY[mc * ldy + wave * 32 + i] : 0.f;
            const float vb = conv ? X[mc * ldx + xcol + wave * 32 + i] : 0.f;
            ra[q] = m < M ? va : 0.f;
            rb[q] = m < M ? vb : 0.f;
            if constexpr (KT2 != 0) {
                const float vc = conv2 ? X2[mc * ldx2 + x2col + i] : 0.f;
                rc[q] = m < M ? vc : 0.f;
            }
        }
    };
    long sl = blockIdx.x;
    if (sl < nslabs) load(sl);
    for (int it = 0; sl < nslabs; sl += gridDim.x, ++it) {
        // ---- split this slab's values; B pieces -> LDS stage it & 1, A pieces stay in registers
        bf16x8 ah[2], am[2], al[2];
        u32x4* st = pb[it & 1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[c] = ra[8 * ks + c];
                bsum += v[c];
            }
            split_bf16x3(v, ah[ks], am[ks], al[ks]);
            if (conv) {
                bf16x8 h_, m_, l_;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = rb[8 * ks + c];
                split_bf16x3(v, h_, m_, l_);
                u32x4* d = st + ((ks * 3) * 2 + half) * KC + wave * 32 + i;
                d[0] = __builtin_bit_cast(u32x4, h_);
                d[2 * KC] = __builtin_bit_cast(u32x4, m_);
                d[4 * KC] = __builtin_bit_cast(u32x4, l_);
            }
            if constexpr (KT2 != 0) {
                if (conv2) {
                    bf16x8 h_, m_, l_;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = rc[8 * ks + c];
                    split_bf16x3(v, h_, m_, l_);
                    u32x4* d = st + ((ks * 3) * 2 + half) * KC + KT * 32 + i;
                    d[0] = __builtin_bit_cast(u32x4, h_);
                    d[2 * KC] = __builtin_bit_cast(u32x4, m_);
                    d[4 * KC] = __builtin_bit_cast(u32x4, l_);
                }
            }
        }
        __syncthreads();                                                // every k tile's pieces of this slab are in LDS; the other
                                                                        // stage is free (all waves finished the previous slab)
        const long nxt = sl + gridDim.x;
        if (nxt < nslabs) load(nxt);                                    // in flight during the MFMAs below
        if (active) {
            // the six products of a k step alternate between TWO accumulators: consecutive MFMAs never write the same one (a dependent
            // accumulate right behind its producer stalls the matrix pipe); all KT at once would need 36 more registers and spill at
            // three workgroups per CU
            constexpr int G = KTT >= 2 ? 2 : 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int k0 = 0; k0 < KTT; k0 += G) {
                    bf16x8 xh[G], xm[G], xl[G];
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        if (k0 + q >= KTT) continue;                       // (odd KT: the last group has one member)
                        const u32x4* f = st + ((ks * 3) * 2 + half) * KC + (k0 + q) * 32 + i;
                        xh[q] = __builtin_bit_cast(bf16x8, f[0]);
                        xm[q] = __builtin_bit_cast(bf16x8, f[2 * KC]);
                        xl[q] = __builtin_bit_cast(bf16x8, f[4 * KC]);
                    }
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (k0 + q < KTT) acc[k0 + q] = mfma_bf16(am[ks], xm[q], acc[k0 + q]);
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (k0 + q < KTT) acc[k0 + q] = mfma_bf16(al[ks], xh[q], acc[k0 + q]);
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (k0 + q < KTT) acc[k0 + q] = mfma_bf16(ah[ks], xl[q], acc[k0 + q]);
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (k0 + q < KTT) acc[k0 + q] = mfma_bf16(am[ks], xh[q], acc[k0 + q]);
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (k0 + q < KTT) acc[k0 + q] = mfma_bf16(ah[ks], xm[q], acc[k0 + q]);
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (k0 + q < KTT) acc[k0 + q] = mfma_bf16(ah[ks], xh[q], acc[k0 + q]);
                }
        }
    }
    if (!active) return;
    float* out = partial + (size_t)blockIdx.x * N * KC;
#pragma unroll
    for (int kt = 0; kt < KTT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // C/D row = A's i = n
            out[(size_t)n * KC + kt * 32 + i] = acc[kt][r];                    // C/D col = B's j = k
        }
    if (bpartial) {
        bsum += __shfl_xor(bsum, 32);
        if (half == 0) bpartial[(size_t)blockIdx.x * N + wave * 32 + i] = bsum;
    }
}

// out[g][j] (+)= scale * sum_b partial[g][b][j]   (fixed summation order: deterministic).  32 consecutive j per
// workgroup (coalesced), 8 interleaved sub-sums per j combined through LDS.
__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ partial, int nblk, long n, float scale,
                                                         float* __restrict__ out, int accumulate) {
    __shared__ float sm[8][32];
    const int jj = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const long j = (long)blockIdx.x * 32 + jj;
    const float* p = partial + (size_t)blockIdx.y * nblk * n;
    // eight loads in flight per thread (fixed interleaving: still deterministic): one dependent load-add per iteration made the
    // 768-partial reductions of a bias (a handful of workgroups) 25 us of pure L2 latency
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < n) {
        int b = sub;
        for (; b + 56 < nblk; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += p[(size_t)(b + 8 * u) * n + j];
        }
        for (; b < nblk; b += 8) a[0] += p[(size_t)b * n + j];
    }
    const float acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    sm[sub][jj] = acc;
    __syncthreads();
    if (sub == 0 && j < n) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += sm[q][jj];
        float* o = out + (size_t)blockIdx.y * n + j;
        *o = accumulate ? *o + scale * tot : scale * tot;
    }
}

// Two reductions of the same partial count in ONE launch (a Linear's dW [n_a] and db [n_b]: the second was four workgroups of 5 us of
// launch latency - 1 150 of the 2 050 launches of a magnet_gnn training step at the reference's batch size were reductions).  Workgroups
// [0, ceil(n_a / 32)) serve a, the rest b; same summation order as k_reduce_partials.
// (gridDim.y = 2: a second pair of arrays of the same sizes - the two Linears of magnet_linear_bwd128_pair_f32 in one launch)
__global__ void __launch_bounds__(256) k_reduce_partials2(const float* __restrict__ pa, long na, float* __restrict__ oa,
                                                          const float* __restrict__ pb, long nb_, float* __restrict__ ob, int nblk,
                                                          float scale, int accumulate, const float* __restrict__ pa2 = nullptr,
                                                          float* __restrict__ oa2 = nullptr, const float* __restrict__ pb2 = nullptr,
                                                          float* __restrict__ ob2 = nullptr) {
    __shared__ float sm[8][32];
    if (blockIdx.y) pa = pa2, oa = oa2, pb = pb2, ob = ob2;
    const int blocks_a = (int)((na + 31) / 32);
    const bool second = (int)blockIdx.x >= blocks_a;
    const float* p = second ? pb : pa;
    const long n = second ? nb_ : na;
    float* out = second ? ob : oa;
    const int jj = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const long j = (long)((int)blockIdx.x - (second ? blocks_a : 0)) * 32 + jj;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < n) {
        int b = sub;
        for (; b + 56 < nblk; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += p[(size_t)(b + 8 * u) * n + j];
        }
        for (; b < nblk; b += 8) a[0] += p[(size_t)b * n + j];
    }
    const float acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    sm[sub][jj] = acc;
    __syncthreads();
    if (sub == 0 && j < n) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += sm[q][jj];
        out[j] = accumulate ? out[j] + scale * tot : scale * tot;
    }
}

// per-graph column sums of dy and of dy * yhat over chunks of STAT_CHUNK rows (H <= 1024)
#define STAT_CHUNK 128
__global__ void k_colstat2_partial(const float* __restrict__ dy, const float* __restrict__ yhat, int rows_per_graph, int H,
                                   float* __restrict__ part1, float* __restrict__ part2) {
    const int g = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int r0 = chunk * STAT_CHUNK, r1 = min(rows_per_graph, r0 + STAT_CHUNK);
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float a1 = 0.f, a2 = 0.f;
        for (int r = r0; r < r1; ++r) {
            const size_t o = ((size_t)g * rows_per_graph + r) * H + c;
            const float d = dy[o];
            a1 += d;
            a2 += d * yhat[o];
        }
        part1[((size_t)g * nchunks + chunk) * H + c] = a1;
        part2[((size_t)g * nchunks + chunk) * H + c] = a2;
    }
}

// ---- InstanceNorm backward (two statistics per graph and channel are computed by the caller with the column-sum
// kernels): dx = (dy - m1 - yhat * m2) / sqrt(var + eps)
__global__ void k_instnorm_bwd_apply(const float* __restrict__ dy, const float* __restrict__ yhat, const float* __restrict__ m1,
                                     const float* __restrict__ m2, const float* __restrict__ var, int rows_per_graph, int H,
                                     float eps, long total, float* __restrict__ dx) {
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= total) return;
    const int c = (int)(i4 % H);
    const int g = (int)((i4 / H) / rows_per_graph);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dy + i4), y = *reinterpret_cast<const f32x4*>(yhat + i4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(m1 + (size_t)g * H + c), b = *reinterpret_cast<const f32x4*>(m2 + (size_t)g * H + c);
    const f32x4 v = *reinterpret_cast<const f32x4*>(var + (size_t)g * H + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (d[k] - a[k] - y[k] * b[k]) / sqrtf(v[k] + eps);
    *reinterpret_cast<f32x4*>(dx + i4) = o;
}

// ... and the Swish backward of the layer below it in the same pass (MPNN: x = h + update_net(..) is normalised, so the gradient of the
// norm's input IS the gradient at update_net_2's Swish output): dz = dx * Swish'(z), block maxima of |dz| as k_swish_bwd_amax leaves them.
// Same arithmetic as k_instnorm_bwd_apply followed by k_swish_bwd_amax (dx is stored and used as the fp32 value it is): dx has the same
// bits, dz differs by the compiler's choice of fused multiply-adds (a few ulp).
__global__ void __launch_bounds__(256) k_instnorm_bwd_apply_swish(const float* __restrict__ dy, const float* __restrict__ yhat,
                                                                  const float* __restrict__ m1, const float* __restrict__ m2,
                                                                  const float* __restrict__ var, int rows_per_graph, int H, float eps,
                                                                  long total, float* __restrict__ dx, const float* __restrict__ z,
                                                                  float* __restrict__ dz, float* __restrict__ amax) {
    __shared__ float wm[4];
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    float m = 0.f;
    if (i4 < total) {
        const int c = (int)(i4 % H);
        const int g = (int)((i4 / H) / rows_per_graph);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + i4), y = *reinterpret_cast<const f32x4*>(yhat + i4);
        const f32x4 a = *reinterpret_cast<const f32x4*>(m1 + (size_t)g * H + c), b = *reinterpret_cast<const f32x4*>(m2 + (size_t)g * H + c);
        const f32x4 v = *reinterpret_cast<const f32x4*>(var + (size_t)g * H + c);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(z + i4);
        f32x4 o, q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k] = (d[k] - a[k] - y[k] * b[k]) / sqrtf(v[k] + eps);
            const float s = sigmoidf_fast(zz[k]);
            q[k] = o[k] * s * (1.0f + zz[k] * (1.0f - s));
            m = fmaxf(m, fabsf(q[k]));
        }
        *reinterpret_cast<f32x4*>(dx + i4) = o;
        *reinterpret_cast<f32x4*>(dz + i4) = q;
    }
    if (!amax) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(reinterpret_cast<unsigned*>(amax) + (blockIdx.x & 255), __float_as_uint(m));
    }
}

// ---- read-out backward: out[n,k] = u[n,tw-1] + dt (k+1) conv2(act(conv1(h[n])))[k]
// one wave per node, workgroups stride over the nodes; weight / bias gradients are accumulated per wave in LDS and
// written once per workgroup as partials [grid][GP], GP = [dw1 8*k1 | db1 8 | dw2 8*k2 | db2 1].
__global__ void __launch_bounds__(256) k_mpnn_readout_bwd(const float* __restrict__ h, const float* __restrict__ dout, int N, int tw,
                                                          const float* __restrict__ w1, const float* __restrict__ b1, int k1, int s1,
                                                          const float* __restrict__ w2, int k2, int use_swish,
                                                          const float* __restrict__ dt, float* __restrict__ dh,
                                                          float* __restrict__ du_last, float* __restrict__ gpart) {
    __shared__ float sh[4][128], sz[4][8 * 64], sdz[4][8 * 64], sd[4][64], sg[4][8 * 16 + 8 + 8 * 16 + 1];
    __shared__ float sw1[8 * 16], sw2[8 * 16], sb1[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L1 = (128 - k1) / s1 + 1;
    const int GP = 8 * k1 + 8 + 8 * k2 + 1;
    for (int q = threadIdx.x; q < 8 * k1; q += 256) sw1[q] = w1[q];
    for (int q = threadIdx.x; q < 8 * k2; q += 256) sw2[q] = w2[q];
    if (threadIdx.x < 8) sb1[threadIdx.x] = b1[threadIdx.x];
    for (int q = lane; q < GP; q += 64) sg[wave][q] = 0.f;
    __syncthreads();
    // index arithmetic that does not depend on the node, hoisted (integer division is slow)
    int plo[2], phi[2];                                                 // conv1 positions p touching h[x], x = lane + 64 i
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int x = lane + 64 * i;
        phi[i] = min(x / s1, L1 - 1);
        plo[i] = x - k1 + 1 <= 0 ? 0 : (x - k1 + s1) / s1;
    }
    int c1_[2], k1_[2];                                                 // dw1 element q = lane + 64 i -> (c, k)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = lane + 64 * i;
        c1_[i] = q / k1;
        k1_[i] = q % k1;
    }
    int c2_[2], k2_[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = lane + 64 * i;
        c2_[i] = q / k2;
        k2_[i] = q % k2;
    }
    const float dt0 = dt[0];
    // every array below is private to its wave: wave-level ordering of LDS traffic is enough
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        sh[wave][lane] = h[(size_t)n * 128 + lane];
        sh[wave][lane + 64] = h[(size_t)n * 128 + lane + 64];
        float dsum = 0.f;
        if (lane < tw) {
            const float d = dout[(size_t)n * tw + lane];
            dsum = d;
            sd[wave][lane] = d * dt0 * (float)(lane + 1);               // d diff
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        if (lane == 0) du_last[n] = dsum;                               // d u[n, tw-1]
        wave_lds_sync();
        if (lane < L1) {                                                // lane = conv1 position p
            for (int c = 0; c < 8; ++c) {
                float z = sb1[c];
                for (int k = 0; k < k1; ++k) z += sw1[c * k1 + k] * sh[wave][lane * s1 + k];
                float g = 0.f;                                          // d act(z)[c][p] = sum_k w2[c][k] ddiff[p - k]
                for (int k = 0; k < k2; ++k) {
                    const int jj = lane - k;
                    if (jj >= 0 && jj < tw) g += sw2[c * k2 + k] * sd[wave][jj];
                }
                sz[wave][c * L1 + lane] = use_swish ? act_swish(z) : z; // activation output (for dw2)
                if (use_swish) {
                    const float sg_ = sigmoidf_fast(z);
                    g *= sg_ * (1.0f + z * (1.0f - sg_));
                }
                sdz[wave][c * L1 + lane] = g;
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                   // dh[x] = sum_{c,p} w1[c][x - p s1] dz[c][p]
            const int x = lane + 64 * i;
            float g = 0.f;
            for (int c = 0; c < 8; ++c)
                for (int pp = plo[i]; pp <= phi[i]; ++pp) g += sw1[c * k1 + x - pp * s1] * sdz[wave][c * L1 + pp];
            dh[(size_t)n * 128 + x] = g;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                   // dw1[c][k] += sum_p dz[c][p] h[p*s1+k]
            const int q = lane + 64 * i;
            if (q < 8 * k1) {
                float g = 0.f;
                for (int pp = 0; pp < L1; ++pp) g += sdz[wave][c1_[i] * L1 + pp] * sh[wave][pp * s1 + k1_[i]];
                sg[wave][q] += g;
            }
        }
        if (lane < 8) {                                                 // db1[c]
            float g = 0.f;
            for (int pp = 0; pp < L1; ++pp) g += sdz[wave][lane * L1 + pp];
            sg[wave][8 * k1 + lane] += g;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                   // dw2[c][k] += sum_j ddiff[j] a[c][j+k]
            const int q = lane + 64 * i;
            if (q < 8 * k2) {
                float g = 0.f;
                for (int jj = 0; jj < tw; ++jj) g += sd[wave][jj] * sz[wave][c2_[i] * L1 + jj + k2_[i]];
                sg[wave][8 * k1 + 8 + q] += g;
            }
        }
        if (lane == 0) {                                                // db2
            float g = 0.f;
            for (int jj = 0; jj < tw; ++jj) g += sd[wave][jj];
            sg[wave][8 * k1 + 8 + 8 * k2] += g;
        }
        wave_lds_sync();
    }
    __syncthreads();
    for (int q = threadIdx.x; q < GP; q += 256)
        gpart[(size_t)blockIdx.x * GP + q] = sg[0][q] + sg[1][q] + sg[2][q] + sg[3][q];
}

// The same backward with the read-out's shape as template arguments (the five (k1, s1, k2) the reference defines, models/mpnn.py:140-163):
// every inner loop has a compile-time trip count and constant LDS offsets, the (channel, position) pairs of the two convolutions are
// spread over all 64 lanes (the generic kernel above leaves 45 of them idle: one lane per conv1 position), and a lane keeps its share of
// the parameter gradients in registers across its nodes instead of read-modify-writing them in LDS per node.  Same partials layout.
// 1.6 ms -> see DESIGN.md 11.4 per launch at 4 x 256^2 nodes (it ran at 0.17 TB/s of its 268 MB).
template <int K1, int S1, int K2, int TW, bool SW>
__global__ void __launch_bounds__(256) k_mpnn_readout_bwd_t(const float* __restrict__ h, const float* __restrict__ dout, int N,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ dt,
                                                            float* __restrict__ dh, float* __restrict__ du_last, float* __restrict__ gpart) {
    constexpr int L1 = (128 - K1) / S1 + 1, GP = 8 * K1 + 8 + 8 * K2 + 1;
    constexpr int NA = (8 * L1 + 63) / 64;                 // rounds of the (channel, position) phase
    constexpr int NPP = (K1 + S1 - 1) / S1;                // conv1 positions that touch one input column, at most
    static_assert(L1 <= 64 && TW <= 64 && L1 - K2 + 1 == TW && 8 * K1 <= 128 && 8 * K2 <= 128, "read-out shape");
    __shared__ float sh[4][128 + 16], sz[4][8 * L1], sdz[4][8 * L1], sd[4][64];
    __shared__ float sw1[8 * K1], sw2[8 * K2], sb1[8];
    __shared__ float sg[4][GP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < 8 * K1; q += 256) sw1[q] = w1[q];
    for (int q = threadIdx.x; q < 8 * K2; q += 256) sw2[q] = w2[q];
    if (threadIdx.x < 8) sb1[threadIdx.x] = b1[threadIdx.x];
    if (lane < 16) sh[wave][128 + lane] = 0.f;             // (reads past column 127 of clamped positions)
    __syncthreads();
    int plo[2], phi[2];                                    // conv1 positions p touching h[x], x = lane + 64 i
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int x = lane + 64 * i;
        phi[i] = min(x / S1, L1 - 1);
        plo[i] = x - K1 + 1 <= 0 ? 0 : (x - K1 + S1) / S1;
    }
    float gw1[2] = {0.f, 0.f}, gw2[2] = {0.f, 0.f}, gb1 = 0.f, gb2 = 0.f;      // this lane's entries of the parameter gradients
    const float dt0 = dt[0];
    float* hs = sh[wave];
    float* zs = sz[wave];
    float* dzs = sdz[wave];
    float* ds = sd[wave];
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        hs[lane] = h[(size_t)n * 128 + lane];
        hs[lane + 64] = h[(size_t)n * 128 + lane + 64];
        float dsum = 0.f, dd = 0.f;
        if (lane < TW) {
            const float d = dout[(size_t)n * TW + lane];
            dsum = d;
            dd = d * dt0 * (float)(lane + 1);                           // d diff
        }
        ds[lane] = dd;                                                  // (zero beyond TW)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off);
        if (lane == 0) du_last[n] = dsum;                               // d u[n, tw-1]
        gb2 += dd;                                                      // db2 = sum_j ddiff[j]: lane j's share, reduced at the end
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < NA; ++r) {                                  // (channel c, conv1 position p) pairs over all lanes
            const int q = lane + 64 * r;
            if (q < 8 * L1) {
                const int c = q / L1, pp = q - c * L1;
                float z = sb1[c];
#pragma unroll
                for (int k = 0; k < K1; ++k) z = fmaf(sw1[c * K1 + k], hs[pp * S1 + k], z);
                float g = 0.f;                                          // d act(z)[c][p] = sum_k w2[c][k] ddiff[p - k]
#pragma unroll
                for (int k = 0; k < K2; ++k) {
                    const int jj = pp - k;
                    g = fmaf(sw2[c * K2 + k], (jj >= 0 && jj < TW) ? ds[jj & 63] : 0.f, g);
                }
                if (SW) {
                    const float sg_ = sigmoidf_fast(z);
                    zs[q] = z * sg_;
                    g *= sg_ * (1.0f + z * (1.0f - sg_));
                } else {
                    zs[q] = z;
                }
                dzs[q] = g;
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                   // dh[x] = sum_{c,p} w1[c][x - p s1] dz[c][p]
            const int x = lane + 64 * i;
            float g = 0.f;
#pragma unroll
            for (int t = 0; t < NPP; ++t) {
                const int pp = plo[i] + t;
                if (pp <= phi[i]) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) g = fmaf(sw1[c * K1 + x - pp * S1], dzs[c * L1 + pp], g);
                }
            }
            dh[(size_t)n * 128 + x] = g;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                   // dw1[c][k] += sum_p dz[c][p] h[p*s1+k]
            const int q = lane + 64 * i;
            if (q < 8 * K1) {
                const int c = q / K1, k = q - c * K1;
                float g = 0.f;
#pragma unroll
                for (int pp = 0; pp < L1; ++pp) g = fmaf(dzs[c * L1 + pp], hs[pp * S1 + k], g);
                gw1[i] += g;
            }
        }
        if (lane < 8) {                                                 // db1[c]
            float g = 0.f;
#pragma unroll
            for (int pp = 0; pp < L1; ++pp) g += dzs[lane * L1 + pp];
            gb1 += g;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                   // dw2[c][k] += sum_j ddiff[j] a[c][j+k]
            const int q = lane + 64 * i;
            if (q < 8 * K2) {
                const int c = q / K2, k = q - c * K2;
                float g = 0.f;
#pragma unroll
                for (int jj = 0; jj < TW; ++jj) g = fmaf(ds[jj], zs[c * L1 + jj + k], g);
                gw2[i] += g;
            }
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gb2 += __shfl_xor(gb2, off);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = lane + 64 * i;
        if (q < 8 * K1) sg[wave][q] = gw1[i];
        if (q < 8 * K2) sg[wave][8 * K1 + 8 + q] = gw2[i];
    }
    if (lane < 8) sg[wave][8 * K1 + lane] = gb1;
    if (lane == 0) sg[wave][8 * K1 + 8 + 8 * K2] = gb2;
    __syncthreads();
    for (int q = threadIdx.x; q < GP; q += 256)
        gpart[(size_t)blockIdx.x * GP + q] = sg[0][q] + sg[1][q] + sg[2][q] + sg[3][q];
}

// ------------------------------------------------------------------------------------ C entry points
#define S_ (hipStream_t) stream
static inline unsigned nb(long n, int b) { return (unsigned)((n + b - 1) / b); }
#define BEGIN() (void)hipGetLastError()
#define END() return (int)hipGetLastError()

extern "C" int magnet_swish_fwd_f32(const float* z, float* r, long n, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_swish_fwd, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, z, r, n);
    END();
}
extern "C" int magnet_swish_bwd_f32(const float* dr, const float* z, float* dz, long n, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_swish_bwd, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, dr, z, dz, n);
    END();
}
extern "C" int magnet_swish_bwd_amax_f32(const float* dr, const float* z, float* dz, long n, float* amax256, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_swish_bwd_amax, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, dr, z, dz, n, amax256);
    END();
}
extern "C" int magnet_axpy_f32(const float* a, const float* b, float alpha, float* out, long n, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_axpy, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, a, b, alpha, out, n);
    END();
}
extern "C" int magnet_mul_f32(const float* a, const float* b, float* out, long n, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_mul, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, a, b, out, n);
    END();
}
extern "C" int magnet_copy_cols_f32(const float* src, int ld_src, int scol, float* dst, int ld_dst, int dcol, int w, long rows,
                                    int accumulate, void* stream) {
    BEGIN();
    if (rows <= 0 || w <= 0) return 0;
    hipLaunchKernelGGL(k_copy_cols, dim3(nb(rows * w, 256)), dim3(256), 0, S_, src, ld_src, scol, dst, ld_dst, dcol, w, rows,
                       accumulate);
    END();
}
extern "C" int magnet_copy_cols_scaled_f32(const float* src, int ld_src, int scol, float* dst, int ld_dst, int dcol, int w, long rows,
                                           float scale, int accumulate, void* stream) {
    BEGIN();
    if (rows <= 0 || w <= 0) return 0;
    if ((w | scol | dcol | ld_src | ld_dst) & 3 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return -2;
    hipLaunchKernelGGL(k_copy_cols4, dim3(nb(rows * (w / 4), 256)), dim3(256), 0, S_, src, ld_src, scol, dst, ld_dst, dcol, w / 4, rows, scale,
                       accumulate);
    END();
}
extern "C" int magnet_edge_pre_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt, long n_edges,
                                   float* p, void* stream) {
    BEGIN();
    if (n_edges <= 0) return 0;
    hipLaunchKernelGGL(k_edge_pre, dim3(nb(n_edges * 32, 256)), dim3(256), 0, S_, A, Bm, src, tgt, n_edges, p);
    END();
}
extern "C" int magnet_edge_mean_bwd_f32(const float* dagg, const int32_t* tgt, const int32_t* rowptr, long n_edges, float* dm,
                                        void* stream) {
    BEGIN();
    if (n_edges <= 0) return 0;
    hipLaunchKernelGGL(k_edge_dm, dim3(nb(n_edges * 32, 256)), dim3(256), 0, S_, dagg, tgt, rowptr, n_edges, dm);
    END();
}
extern "C" int magnet_segment_sum128_f32(const float* x, const int32_t* rowptr, const int32_t* perm, int n_rows, float* out,
                                         void* stream) {
    BEGIN();
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(k_segment_sum128, dim3(nb(n_rows, 4)), dim3(256), 0, S_, x, rowptr, perm, n_rows, out);
    END();
}
// dW[128][kc] (+)= dY^T X[:, xcol : xcol+kc], db[128] (+)= column sums of dY (db may be null).
// workspace: nblk * (128 * kc + 128) + 128 * kc floats, nblk = min(768, ceil(M / 32)).
// general form: dY has n_out in {32, 64, 128} columns (weight rows), dW is [n_out][ldw]
extern "C" int magnet_wgrad_n_f32(const float* dY, int ldy, int n_out, const float* X, int ldx, int xcol, int kc, long M, float* dW,
                                  int ldw, int wcol, float* db, float* workspace, int accumulate, void* stream) {
    BEGIN();
    if (M <= 0) return 0;
    if (kc % 32 || kc < 32 || kc > 128 || (n_out != 32 && n_out != 64 && n_out != 128)) return -2;
    const int N = n_out;
    const int nblk = (int)((M + WG_SLAB - 1) / WG_SLAB < 768 ? (M + WG_SLAB - 1) / WG_SLAB : 768);
    float* part = workspace;
    float* bpart = db ? workspace + (size_t)nblk * N * kc : nullptr;
    static int use_fp32 = -1;        // MAGNET_WGRAD=fp32 keeps the fp32-MFMA kernel (A/B knob; the split kernel is the default)
    if (use_fp32 < 0) { const char* e = getenv("MAGNET_WGRAD"); use_fp32 = (e && e[0] == 'f') ? 1 : 0; }
    if (use_fp32) {
        switch (kc / 32) {
            case 1: hipLaunchKernelGGL(k_wgrad_partial<1>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
            case 2: hipLaunchKernelGGL(k_wgrad_partial<2>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
            case 3: hipLaunchKernelGGL(k_wgrad_partial<3>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
            default: hipLaunchKernelGGL(k_wgrad_partial<4>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
        }
    } else {
        switch (kc / 32) {
            case 1: hipLaunchKernelGGL(k_wgrad_split<1>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
            case 2: hipLaunchKernelGGL(k_wgrad_split<2>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
            case 3: hipLaunchKernelGGL(k_wgrad_split<3>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
            default: hipLaunchKernelGGL(k_wgrad_split<4>, dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart); break;
        }
    }
    // reduce into a scratch tile, then place it at dW[:, wcol : wcol+kc]
    float* tile = workspace + (size_t)nblk * (128 * kc + 128);
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb((long)N * kc, 32)), dim3(256), 0, S_, part, nblk, (long)N * kc, 1.0f, tile, 0);
    hipLaunchKernelGGL(k_copy_cols, dim3(nb((long)N * kc, 256)), dim3(256), 0, S_, tile, kc, 0, dW, ldw, wcol, kc, (long)N, accumulate);
    if (db) hipLaunchKernelGGL(k_reduce_partials, dim3(nb(N, 32)), dim3(256), 0, S_, bpart, nblk, (long)N, 1.0f, db, accumulate);
    END();
}
// dW[r][wcol + t kc + c] (+)= tiles[t][r][c]
__global__ void k_place_tiles(const float* __restrict__ tiles, int ntile, int N, int kc, float* __restrict__ dW, int ldw, int wcol,
                              int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)ntile * N * kc) return;
    const int t = (int)(i / ((long)N * kc)), r = (int)(i / kc) % N, c = (int)(i % kc);
    float* d = dW + (size_t)r * ldw + wcol + t * kc + c;
    *d = accumulate ? *d + tiles[i] : tiles[i];
}
// The same for `ntile` neighbouring column tiles of kc columns each, X[:, xcol + t kc ..) -> dW[:, wcol + t kc ..), in ONE launch of each
// kernel (gridDim.y = ntile): the nine 128-column taps of an EDSR convolution on a 64 x 64 map were nine 128-workgroup launches plus
// nine reductions and nine copies.  workspace: ntile * nblk * n_out * kc + nblk * n_out + ntile * n_out * kc floats.
extern "C" int magnet_wgrad_tiles_f32(const float* dY, int ldy, int n_out, const float* X, int ldx, int xcol, int kc, int ntile, long M,
                                      float* dW, int ldw, int wcol, float* db, float* workspace, int accumulate, void* stream) {
    BEGIN();
    if (M <= 0 || ntile <= 0) return 0;
    if (kc % 32 || kc < 32 || kc > 128 || (n_out != 32 && n_out != 64 && n_out != 128)) return -2;
    const int N = n_out;
    // row blocks (= partial sums per tile): ~768 workgroups over ALL tiles.  768 per tile made a six-tile launch 4 608 workgroups of 43
    // rows each at the reference's EDSR shape (32 768 pixels) and its reduction a 113 MB read: 89 + 83 us per convolution
    const int cap = ntile > 1 ? (768 / ntile > 32 ? 768 / ntile : 32) : 768;   // (x ntile <= 768: one round of three workgroups per CU)
    const int nblk = (int)((M + WG_SLAB - 1) / WG_SLAB < cap ? (M + WG_SLAB - 1) / WG_SLAB : cap);
    float* part = workspace;
    float* bpart = workspace + (size_t)ntile * nblk * N * kc;
    float* tiles = bpart + (size_t)nblk * N;
    const dim3 g(nblk, ntile);
    switch (kc / 32) {
        case 1: hipLaunchKernelGGL(k_wgrad_split<1>, g, dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, db ? bpart : nullptr); break;
        case 2: hipLaunchKernelGGL(k_wgrad_split<2>, g, dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, db ? bpart : nullptr); break;
        case 3: hipLaunchKernelGGL(k_wgrad_split<3>, g, dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, db ? bpart : nullptr); break;
        default: hipLaunchKernelGGL(k_wgrad_split<4>, g, dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, db ? bpart : nullptr); break;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb((long)N * kc, 32), ntile), dim3(256), 0, S_, part, nblk, (long)N * kc, 1.0f, tiles, 0);
    hipLaunchKernelGGL(k_place_tiles, dim3(nb((long)ntile * N * kc, 256)), dim3(256), 0, S_, tiles, ntile, N, kc, dW, ldw, wcol, accumulate);
    if (db) hipLaunchKernelGGL(k_reduce_partials, dim3(nb(N, 32)), dim3(256), 0, S_, bpart, nblk, (long)N, 1.0f, db, accumulate);
    END();
}
// Two column blocks of one Linear's weight gradient in ONE pass over dY: dW[0:128, wcol : wcol+128] (+)= dY^T X[:, xcol : xcol+128] and
// dW[0:128, wcol2 : wcol2+32] (+)= dY^T X2[:, x2col : x2col+32] (the MPNN node-level Linears over [h | node_in], models/mpnn.py:73-90),
// db (+)= colsum(dY) (may be null).  workspace: nblk * (128 * 160 + 128) + 128 * 160 floats, nblk = min(768, ceil(M / 32)).
extern "C" int magnet_wgrad_pair_f32(const float* dY, int ldy, const float* X, int ldx, int xcol, const float* X2, int ldx2, int x2col,
                                     long M, float* dW, int ldw, int wcol, int wcol2, float* db, float* workspace, int accumulate,
                                     void* stream) {
    BEGIN();
    if (M <= 0) return 0;
    constexpr int N = 128, KC = 160;
    const int nblk = (int)((M + WG_SLAB - 1) / WG_SLAB < 768 ? (M + WG_SLAB - 1) / WG_SLAB : 768);
    float* part = workspace;
    float* bpart = db ? workspace + (size_t)nblk * N * KC : nullptr;
    hipLaunchKernelGGL((k_wgrad_split<4, 1>), dim3(nblk), dim3(256), 0, S_, dY, ldy, N, X, ldx, xcol, M, part, bpart, X2, ldx2, x2col);
    float* tile = workspace + (size_t)nblk * (N * KC + N);
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb((long)N * KC, 32)), dim3(256), 0, S_, part, nblk, (long)N * KC, 1.0f, tile, 0);
    hipLaunchKernelGGL(k_copy_cols, dim3(nb((long)N * 128, 256)), dim3(256), 0, S_, tile, KC, 0, dW, ldw, wcol, 128, (long)N, accumulate);
    hipLaunchKernelGGL(k_copy_cols, dim3(nb((long)N * 32, 256)), dim3(256), 0, S_, tile, KC, 128, dW, ldw, wcol2, 32, (long)N, accumulate);
    if (db) hipLaunchKernelGGL(k_reduce_partials, dim3(nb(N, 32)), dim3(256), 0, S_, bpart, nblk, (long)N, 1.0f, db, accumulate);
    END();
}
// dW[0:128, wcol:wcol+kc] (+)= dY[M,128]^T X[M, xcol:xcol+kc], db[128] (+)= column sums of dY (db may be null).
// workspace: nblk * (128 * kc + 128) + 128 * kc floats, nblk = min(768, ceil(M / 32)).
extern "C" int magnet_wgrad_f32(const float* dY, int ldy, const float* X, int ldx, int xcol, int kc, long M, float* dW, int ldw,
                                int wcol, float* db, float* workspace, int accumulate, void* stream) {
    return magnet_wgrad_n_f32(dY, ldy, 128, X, ldx, xcol, kc, M, dW, ldw, wcol, db, workspace, accumulate, stream);
}
extern "C" int magnet_reduce_partials_f32(const float* partial, int nblk, long n, float scale, float* out, int accumulate,
                                          void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb(n, 32)), dim3(256), 0, S_, partial, nblk, n, scale, out, accumulate);
    END();
}
int magnet_launch_reduce_partials2(const float* pa, long na, float* oa, const float* pb, long nb_, float* ob, int nblk, float scale,
                                   int accumulate, void* stream) {
    BEGIN();
    if (na <= 0 || nb_ <= 0) return -2;
    hipLaunchKernelGGL(k_reduce_partials2, dim3(nb(na, 32) + nb(nb_, 32)), dim3(256), 0, S_, pa, na, oa, pb, nb_, ob, nblk, scale, accumulate,
                       nullptr, nullptr, nullptr, nullptr);
    END();
}
// the same for TWO Linears (two [nblk][na] / [nblk][nb] pairs) in one launch
int magnet_launch_reduce_partials2x2(const float* pa, long na, float* oa, const float* pb, long nb_, float* ob, const float* pa2, float* oa2,
                                     const float* pb2, float* ob2, int nblk, float scale, int accumulate, void* stream) {
    BEGIN();
    if (na <= 0 || nb_ <= 0) return -2;
    hipLaunchKernelGGL(k_reduce_partials2, dim3(nb(na, 32) + nb(nb_, 32), 2), dim3(256), 0, S_, pa, na, oa, pb, nb_, ob, nblk, scale, accumulate,
                       pa2, oa2, pb2, ob2);
    END();
}
// dx = InstanceNorm backward of dy given the normalised output yhat and the biased variance var [n_graphs][H]
// (both kept by the forward).  workspace: (2 * n_graphs * ceil(rows / 128) + 2 * n_graphs) * H floats.
extern "C" int magnet_instance_norm_bwd_f32(const float* dy, const float* yhat, const float* var, int n_graphs,
                                            int rows_per_graph, int H, float eps, float* dx, float* workspace, void* stream) {
    BEGIN();
    const long total = (long)n_graphs * rows_per_graph * H;
    if (total <= 0) return 0;
    if (H % 4) return -2;
    const int nchunks = (rows_per_graph + STAT_CHUNK - 1) / STAT_CHUNK;
    float* p1 = workspace;
    float* p2 = p1 + (size_t)n_graphs * nchunks * H;
    float* m1 = p2 + (size_t)n_graphs * nchunks * H;
    float* m2 = m1 + (size_t)n_graphs * H;
    hipLaunchKernelGGL(k_colstat2_partial, dim3(nchunks, n_graphs), dim3(128), 0, S_, dy, yhat, rows_per_graph, H, p1, p2);
    const float inv = 1.0f / (float)rows_per_graph;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb(H, 32), n_graphs), dim3(256), 0, S_, p1, nchunks, (long)H, inv, m1, 0);
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb(H, 32), n_graphs), dim3(256), 0, S_, p2, nchunks, (long)H, inv, m2, 0);
    hipLaunchKernelGGL(k_instnorm_bwd_apply, dim3(nb(total / 4, 256)), dim3(256), 0, S_, dy, yhat, m1, m2, var, rows_per_graph, H,
                       eps, total, dx);
    END();
}
// The same with the Swish backward of the producing layer in the apply pass: dx as above, dz = dx * Swish'(z) (z [rows][H]: the
// pre-activation whose Swish output was normalised), amax256: NULL or 256 floats zeroed by the caller that receive the block maxima of |dz|
// (magnet_swish_bwd_amax_f32's).  dx bit-identical to magnet_instance_norm_bwd_f32's, dz within a few ulp of magnet_swish_bwd(_amax)_f32's.
extern "C" int magnet_instance_norm_bwd_swish_f32(const float* dy, const float* yhat, const float* var, int n_graphs, int rows_per_graph,
                                                  int H, float eps, const float* z, float* dx, float* dz, float* amax256, float* workspace,
                                                  void* stream) {
    BEGIN();
    const long total = (long)n_graphs * rows_per_graph * H;
    if (total <= 0) return 0;
    if (H % 4) return -2;
    const int nchunks = (rows_per_graph + STAT_CHUNK - 1) / STAT_CHUNK;
    float* p1 = workspace;
    float* p2 = p1 + (size_t)n_graphs * nchunks * H;
    float* m1 = p2 + (size_t)n_graphs * nchunks * H;
    float* m2 = m1 + (size_t)n_graphs * H;
    hipLaunchKernelGGL(k_colstat2_partial, dim3(nchunks, n_graphs), dim3(128), 0, S_, dy, yhat, rows_per_graph, H, p1, p2);
    const float inv = 1.0f / (float)rows_per_graph;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb(H, 32), n_graphs), dim3(256), 0, S_, p1, nchunks, (long)H, inv, m1, 0);
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb(H, 32), n_graphs), dim3(256), 0, S_, p2, nchunks, (long)H, inv, m2, 0);
    hipLaunchKernelGGL(k_instnorm_bwd_apply_swish, dim3(nb(total / 4, 256)), dim3(256), 0, S_, dy, yhat, m1, m2, var, rows_per_graph, H,
                       eps, total, dx, z, dz, amax256);
    END();
}
// The same for graphs of UNEQUAL size (graph g = rows gptr[g] .. gptr[g + 1]; PyG's InstanceNorm(x, batch) takes any batch vector,
// models/mpnn.py:63,70): one workgroup per (graph, 32-channel window), 32 row groups x 32 channels, two passes over the graph's rows -
// the column means of dy and of dy * yhat in a fixed summation order, then dx = (dy - mean(dy) - yhat mean(dy yhat)) / sqrt(var + eps).
__global__ void __launch_bounds__(1024) k_instance_norm_ragged_bwd(const float* __restrict__ dy, const float* __restrict__ yhat,
                                                                   const float* __restrict__ var, const int32_t* __restrict__ gptr, int H,
                                                                   float eps, float* __restrict__ dx) {
    __shared__ float s1[32][33], s2[32][33];
    __shared__ float stat[2][32];
    const int g = blockIdx.x, cl = threadIdx.x & 31, c = blockIdx.y * 32 + cl, grp = threadIdx.x >> 5;
    const int r0 = gptr[g], nr = gptr[g + 1] - r0;
    if (nr <= 0) return;
    const float* dg = dy + (size_t)r0 * H;
    const float* yg = yhat + (size_t)r0 * H;
    float a1 = 0.f, a2 = 0.f;
    if (c < H)
        for (int r = grp; r < nr; r += 32) {
            const float d = dg[(size_t)r * H + c];
            a1 += d;
            a2 = fmaf(d, yg[(size_t)r * H + c], a2);
        }
    s1[grp][cl] = a1, s2[grp][cl] = a2;
    __syncthreads();
    if (grp == 0) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) t1 += s1[q][cl], t2 += s2[q][cl];
        stat[0][cl] = t1 / (float)nr, stat[1][cl] = t2 / (float)nr;
    }
    __syncthreads();
    if (c < H) {
        const float m1 = stat[0][cl], m2 = stat[1][cl], is = 1.0f / sqrtf(var[(size_t)g * H + c] + eps);
        float* xg = dx + (size_t)r0 * H;
        for (int r = grp; r < nr; r += 32) {
            const size_t o = (size_t)r * H + c;
            xg[o] = (dg[o] - m1 - yg[o] * m2) * is;
        }
    }
}
extern "C" int magnet_instance_norm_ragged_bwd_f32(const float* dy, const float* yhat, const float* var, const int32_t* gptr, int n_graphs,
                                                   int H, float eps, float* dx, void* stream) {
    BEGIN();
    if (n_graphs <= 0) return 0;
    hipLaunchKernelGGL(k_instance_norm_ragged_bwd, dim3(n_graphs, nb(H, 32)), dim3(1024), 0, S_, dy, yhat, var, gptr, H, eps, dx);
    END();
}
// gpart: [min(1024, ceil(n/4))][8*k1 + 8 + 8*k2 + 1] partial parameter gradients (reduce with magnet_reduce_partials_f32)
extern "C" int magnet_mpnn_readout_bwd_f32(const float* h, const float* dout, int n, int tw, const float* w1, const float* b1,
                                           int k1, int s1, const float* w2, int k2, int use_swish, const float* dt, float* dh,
                                           float* du_last, float* gpart, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    const int L1 = (128 - k1) / s1 + 1;
    if (L1 > 64 || tw > 64 || k1 > 16 || k2 > 16 || L1 - k2 + 1 != tw) return -2;
    const unsigned grid = nb(n, 4) < 1024u ? nb(n, 4) : 1024u;
#define RO_BWD(K1_, S1_, K2_, TW_, SW_)                                                                                            \
    if (k1 == K1_ && s1 == S1_ && k2 == K2_ && tw == TW_ && (use_swish != 0) == SW_) {                                             \
        hipLaunchKernelGGL((k_mpnn_readout_bwd_t<K1_, S1_, K2_, TW_, SW_>), dim3(grid), dim3(256), 0, S_, h, dout, n, w1, b1, w2, dt, dh, \
                           du_last, gpart);                                                                                        \
        END();                                                                                                                     \
    }
    // the read-outs the reference defines (models/mpnn.py:140-163; 1-D tw = 10 has no Swish, mpnn_2d.py has one)
    RO_BWD(16, 6, 10, 10, true) RO_BWD(16, 6, 10, 10, false) RO_BWD(16, 5, 8, 16, true) RO_BWD(15, 4, 10, 20, true)
    RO_BWD(16, 3, 14, 25, true) RO_BWD(12, 2, 10, 50, true)
#undef RO_BWD
    hipLaunchKernelGGL(k_mpnn_readout_bwd, dim3(grid), dim3(256), 0, S_, h, dout, n, tw, w1, b1, k1, s1, w2, k2, use_swish, dt,
                       dh, du_last, gpart);
    END();
}

// ---- fused Adam over one flat parameter buffer (torch.optim.Adam semantics, configure_optimizers models/mpnn.py:206-214:
// L2 weight decay added to the gradient, bias-corrected moments, eps added after the square root).
//   g' = g + wd p;  m += (g' - m)(1-b1);  v = b2 v + (1-b2) g'^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                       float lr_over_bc1, float beta1, float beta2, float eps, float wd, float sqrt_bc2) {
    EW_BODY(, )
    const f32x4 pv = ld4(p, i4, cnt), gv = ld4(g, i4, cnt), mv = ld4(m, i4, cnt), vv = ld4(v, i4, cnt);
    f32x4 po, mo, vo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float gg = fmaf(wd, pv[c], gv[c]);
        mo[c] = fmaf(gg - mv[c], 1.0f - beta1, mv[c]);              // torch: exp_avg.lerp_(grad, 1 - beta1)
        vo[c] = fmaf((1.0f - beta2) * gg, gg, beta2 * vv[c]);       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vo[c]) / sqrt_bc2 + eps;
        po[c] = pv[c] - lr_over_bc1 * (mo[c] / denom);
    }
    st4(p, i4, cnt, po);
    st4(m, i4, cnt, mo);
    st4(v, i4, cnt, vo);
}

extern "C" int magnet_adam_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    if (step < 1) return -2;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, p, g, m, v, n, (float)((double)lr / bc1), beta1, beta2,
                       eps, weight_decay, (float)sqrt(bc2));
    END();
}
