// MPNN read-out + Euler step (models/mpnn.py:197-201).  Own translation unit: the lane-per-node kernel is one fully
// unrolled block of ~3 400 fmas, which needs hipcc's pragma-unroll threshold raised (magnet_amd/build.py: EXTRA_FLAGS).
#include "common.cuh"

// ---------------------------------------------------------------------------- read-out + Euler step
// diff = Conv1d(8->1,k2)( [Swish]( Conv1d(1->8,k1,stride s1)(h[:, None, :]) ) ),
// out[n, k] = u[n, tw-1] + dt*(k+1) * diff[n, k]        (models/mpnn.py:197-201, conv specs :140-163)
// One wave per node; conv1 output (8 x L1) lives in LDS.
__global__ void __launch_bounds__(256) k_mpnn_readout(const float* __restrict__ h, const float* __restrict__ u, int N,
                                                      int tw, const float* __restrict__ w1, const float* __restrict__ b1,
                                                      int k1, int s1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, int k2, int use_swish,
                                                      const float* __restrict__ dt, float* __restrict__ out) {
    __shared__ float sh[4][128];
    __shared__ float sz[4][8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    const int L1 = (128 - k1) / s1 + 1;
    if (n < N) {
        sh[wave][lane] = h[(size_t)n * 128 + lane];
        sh[wave][lane + 64] = h[(size_t)n * 128 + lane + 64];
    }
    __syncthreads();
    if (n < N) {
        for (int o = lane; o < 8 * L1; o += 64) {
            const int c = o / L1, p = o % L1;
            float acc = b1[c];
            for (int k = 0; k < k1; ++k) acc += w1[c * k1 + k] * sh[wave][p * s1 + k];
            sz[wave][o] = use_swish ? act_swish(acc) : acc;
        }
    }
    __syncthreads();
    if (n < N && lane < tw) {
        float acc = b2[0];
        for (int c = 0; c < 8; ++c)
            for (int k = 0; k < k2; ++k) acc += w2[c * k2 + k] * sz[wave][c * L1 + lane + k];
        out[(size_t)n * tw + lane] = u[(size_t)n * tw + tw - 1] + dt[0] * (float)(lane + 1) * acc;
    }
}

// The same read-out with one LANE per node for the conv specs of models/mpnn.py:140-163 (compile-time k1 / s1 / k2 / tw):
// the node's 128-float row sits in registers, the weights are wave-uniform (scalar loads), the 8 x L1 hidden map is never
// stored - every hidden value goes straight into the (at most k2) outputs it feeds.  ~3 400 fmas per node at tw = 10 and
// nothing else; the wave-per-node version above spends most of its time on LDS traffic and loop overhead (0.215 ms at
// N = 262 144; this one: see DESIGN.md 3.4).
template <int K1, int S1, int K2, int TW, bool SWISH>
__global__ void __launch_bounds__(256) k_mpnn_readout_rows(const float* __restrict__ h, const float* __restrict__ u, int N,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ dt, float* __restrict__ out) {
    constexpr int L1 = (128 - K1) / S1 + 1;
    static_assert(L1 == TW + K2 - 1, "conv2 must produce exactly tw outputs");
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float hr[128];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(h + (size_t)n * 128 + 4 * q);
        hr[4 * q] = v[0], hr[4 * q + 1] = v[1], hr[4 * q + 2] = v[2], hr[4 * q + 3] = v[3];
    }
    float o[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) o[t] = b2[0];
#pragma unroll
    for (int p = 0; p < L1; ++p)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float z = b1[c];
#pragma unroll
            for (int k = 0; k < K1; ++k) z = fmaf(w1[c * K1 + k], hr[p * S1 + k], z);
            if (SWISH) z = act_swish(z);
#pragma unroll
            for (int t = 0; t < TW; ++t)
                if (p - t >= 0 && p - t < K2) o[t] = fmaf(w2[c * K2 + p - t], z, o[t]);
        }
    const float last = u[(size_t)n * TW + TW - 1], d = dt[0];
#pragma unroll
    for (int t = 0; t < TW; ++t) out[(size_t)n * TW + t] = last + d * (float)(t + 1) * o[t];
}

// The lane-per-node read-out with the rows fetched through LDS: a lane reading its own 512-byte row touches 64 different 128-byte lines
// per load instruction (every line is requested eight times; 92 us for 134 MB at 4 x 256^2 nodes).  Here a wave fetches its 64 rows 32 columns
// at a time with whole-line loads (lane l: row 8 i + l / 8, 16-byte piece l % 8), passes them through a wave-private 64 x 36-float LDS tile
// and every lane picks up its own row from there.  NORM: the InstanceNorm of the last hop (models/mpnn.py:70) is applied to the row on the
// way ((h - mean[g]) / sqrt(var[g] + eps), g = node / rows_per_graph), so that pass over h disappears as well.
#define RO_TS 36
template <int K1, int S1, int K2, int TW, bool SWISH, bool NORM>
__global__ void __launch_bounds__(256) k_mpnn_readout_lines(const float* __restrict__ h, const float* __restrict__ u, int N,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ dt, const float* __restrict__ mean,
                                                            const float* __restrict__ var, float eps, int n_graphs, int rows_per_graph,
                                                            float* __restrict__ out) {
    constexpr int L1 = (128 - K1) / S1 + 1;
    static_assert(L1 == TW + K2 - 1, "conv2 must produce exactly tw outputs");
    __shared__ __attribute__((aligned(16))) float tiles[4][64 * RO_TS];
    __shared__ __attribute__((aligned(16))) float st[2][NORM ? 8 : 1][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (NORM) {
        for (int i = threadIdx.x; i < n_graphs * 128; i += 256) {
            st[0][i >> 7][i & 127] = mean[i];
            st[1][i >> 7][i & 127] = 1.0f / sqrtf(var[i] + eps);
        }
        __syncthreads();
    }
    float* tl = tiles[wave];
    const long n0 = (long)blockIdx.x * 256 + wave * 64;
    if (n0 >= N) return;
    const long n = n0 + lane;
    const int gid = NORM ? (int)((n < N ? n : (long)N - 1) / rows_per_graph) : 0;
    float hr[128];
    f32x4 v[2][8];
    auto fetch = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long r = n0 + 8 * i + (lane >> 3);
            r = r < N ? r : (long)N - 1;
            v[buf][i] = *reinterpret_cast<const f32x4*>(h + r * 128 + c * 32 + 4 * (lane & 7));
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) fetch(c + 1, (c + 1) & 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(tl + (8 * i + (lane >> 3)) * RO_TS + 4 * (lane & 7)) = v[c & 1][i];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 t = *reinterpret_cast<const f32x4*>(tl + lane * RO_TS + 4 * q);
            if (NORM) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(&st[0][gid][c * 32 + 4 * q]);
                const f32x4 r = *reinterpret_cast<const f32x4*>(&st[1][gid][c * 32 + 4 * q]);
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = (t[k] - m[k]) * r[k];
            }
            hr[c * 32 + 4 * q] = t[0], hr[c * 32 + 4 * q + 1] = t[1], hr[c * 32 + 4 * q + 2] = t[2], hr[c * 32 + 4 * q + 3] = t[3];
        }
    }
    if (n >= N) return;
    float o[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) o[t] = b2[0];
#pragma unroll
    for (int p = 0; p < L1; ++p)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float z = b1[c];
#pragma unroll
            for (int k = 0; k < K1; ++k) z = fmaf(w1[c * K1 + k], hr[p * S1 + k], z);
            if (SWISH) z = act_swish(z);
#pragma unroll
            for (int t = 0; t < TW; ++t)
                if (p - t >= 0 && p - t < K2) o[t] = fmaf(w2[c * K2 + p - t], z, o[t]);
        }
    const float last = u[(size_t)n * TW + TW - 1], d = dt[0];
#pragma unroll
    for (int t = 0; t < TW; ++t) out[(size_t)n * TW + t] = last + d * (float)(t + 1) * o[t];
}

#define LAUNCH_CHECK() return (int)hipGetLastError()

// magnet_mpnn_readout_f32 with the InstanceNorm of the last hop applied to h on load (mean / var [n_graphs][128]: the statistics of
// magnet_instance_norm_stats_f32; mean == NULL: h is used as it is).  Returns -2 for shapes without a lane-per-node kernel (conv specs
// other than models/mpnn.py:140-163's, n < 8192, more than 8 graphs): the caller then normalises first and calls magnet_mpnn_readout_f32.
extern "C" int magnet_mpnn_readout_norm_f32(const float* h, const float* mean, const float* var, float eps, int n_graphs,
                                            int rows_per_graph, const float* u, int n, int tw, const float* w1, const float* b1,
                                            int k1, int s1, const float* w2, const float* b2, int k2, int use_swish,
                                            const float* dt, float* out, void* stream) {
    (void)hipGetLastError();
    if (n <= 0) return 0;
    if (n < 8192) return -2;
    if (mean && (n_graphs < 1 || n_graphs > 8 || rows_per_graph < 1 || !var)) return -2;
#define READOUT_LINES(K1_, S1_, K2_, TW_, SW_, NM_)                                                                                      \
    hipLaunchKernelGGL((k_mpnn_readout_lines<K1_, S1_, K2_, TW_, SW_, NM_>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h, \
                       u, n, w1, b1, w2, b2, dt, mean, var, eps, n_graphs, rows_per_graph, out)
#define READOUT_CASE2(K1_, S1_, K2_, TW_)                                                                                                \
    if (k1 == K1_ && s1 == S1_ && k2 == K2_ && tw == TW_) {                                                                              \
        if (use_swish && mean) READOUT_LINES(K1_, S1_, K2_, TW_, true, true);                                                            \
        else if (use_swish) READOUT_LINES(K1_, S1_, K2_, TW_, true, false);                                                              \
        else if (mean) READOUT_LINES(K1_, S1_, K2_, TW_, false, true);                                                                   \
        else READOUT_LINES(K1_, S1_, K2_, TW_, false, false);                                                                            \
        LAUNCH_CHECK();                                                                                                                  \
    }
    READOUT_CASE2(16, 6, 10, 10)
    READOUT_CASE2(16, 5, 8, 16)
    READOUT_CASE2(15, 4, 10, 20)
    READOUT_CASE2(16, 3, 14, 25)
#undef READOUT_CASE2
#undef READOUT_LINES
    return -2;
}

extern "C" int magnet_mpnn_readout_f32(const float* h, const float* u, int n, int tw, const float* w1, const float* b1,
                                       int k1, int s1, const float* w2, const float* b2, int k2, int use_swish,
                                       const float* dt, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n <= 0) return 0;
    const int L1 = (128 - k1) / s1 + 1;
    if (L1 > 64 || tw > 64 || L1 - k2 + 1 != tw) return -2;
    {   // large n, the reference's conv specs: rows through LDS as whole lines
        const int rc = magnet_mpnn_readout_norm_f32(h, nullptr, nullptr, 0.f, 0, 0, u, n, tw, w1, b1, k1, s1, w2, b2, k2, use_swish, dt, out,
                                                    stream);
        if (rc != -2) return rc;
    }
    // lane-per-node kernels: one long serial block per lane - right when there are nodes enough to fill the chip; below that
    // (the launch-bound 1-D configs) the wave-per-node kernel finishes sooner (n = 100: 46 us vs the time of a few waves)
#define READOUT_CASE(K1_, S1_, K2_, TW_)                                                                                  \
    if (n >= 8192 && k1 == K1_ && s1 == S1_ && k2 == K2_ && tw == TW_) {                                                               \
        if (use_swish) hipLaunchKernelGGL((k_mpnn_readout_rows<K1_, S1_, K2_, TW_, true>), dim3((n + 255) / 256), dim3(256), 0, \
                                          (hipStream_t)stream, h, u, n, w1, b1, w2, b2, dt, out);                          \
        else hipLaunchKernelGGL((k_mpnn_readout_rows<K1_, S1_, K2_, TW_, false>), dim3((n + 255) / 256), dim3(256), 0,    \
                                (hipStream_t)stream, h, u, n, w1, b1, w2, b2, dt, out);                                    \
        LAUNCH_CHECK();                                                                                                   \
    }
    READOUT_CASE(16, 6, 10, 10)     // models/mpnn.py:140-163 / mpnn_2d.py:140-163; tw = 50 (59 x 8 hidden values) stays generic
    READOUT_CASE(16, 5, 8, 16)
    READOUT_CASE(15, 4, 10, 20)
    READOUT_CASE(16, 3, 14, 25)
#undef READOUT_CASE
    hipLaunchKernelGGL(k_mpnn_readout, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, u, n, tw, w1, b1, k1,
                       s1, w2, b2, k2, use_swish, dt, out);
    LAUNCH_CHECK();
}

