// Backward building blocks of the generic fused-MLP op and of the MAgNet-specific kernels (training of MAgNet[GNN],
// BASELINE.json configs[4]; the reference differentiates models/magnet_gnn.py:312-376 with torch autograd).
// Everything here is correctness-first and materialising: the fused forward kernels stay in use for the forward pass,
// the backward recomputes a chain's activations layer by layer (magnet_mlp_* with one Linear) and walks it back with
// these kernels + magnet_wgrad_f32 + the transposed-weight GEMMs.
#include "common.cuh"
#define MAGNET_SEG_RUNS 9          // runs per launch of k_segment_sum_cols_runs (the taps of a 3 x 3 convolution)

#define S_ (hipStream_t) stream
static inline unsigned nb(long n, int b) { return (unsigned)((n + b - 1) / b); }
#define BEGIN() (void)hipGetLastError()
#define END() return (int)hipGetLastError()

// X0[row][kt*32 + c] = the first-layer input the fused kernel streams (gathered / concatenated / differenced sources)
__global__ void k_materialize_ktiles(const MagnetMlpArgs a, float* __restrict__ x0, int ldx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (row, kt, quad of 4 columns)
    const long total = (long)a.M * a.nkt * 8;
    if (i >= total) return;
    const int q = (int)(i & 7);
    const int kt = (int)((i >> 3) % a.nkt);
    const int row = (int)(i / (8L * a.nkt));
    const MagnetKTile& d = a.kt[kt];
    f32x4 v = {0, 0, 0, 0};
    for (int s = 0; s < d.n; ++s) {
        const MagnetSrc& src = d.s[s];
        const long r = src.idx ? (long)src.idx[row] : (long)(row / src.rdiv);
        v += *reinterpret_cast<const f32x4*>(src.p + r * src.ld + src.col + 4 * q) * src.scale;
    }
    *reinterpret_cast<f32x4*>(x0 + (size_t)row * ldx + kt * 32 + 4 * q) = v;
}

// dz = y > 0 ? dy : 0     (y = ReLU output kept from the recomputation)
__global__ void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz, long n) {
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i4), v = *reinterpret_cast<const f32x4*>(y + i4);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = v[c] > 0.f ? g[c] : 0.f;
        *reinterpret_cast<f32x4*>(dz + i4) = o;
    } else {
        for (long k = i4; k < n; ++k) dz[k] = y[k] > 0.f ? dy[k] : 0.f;
    }
}

// LayerNorm backward over rows of D <= 256 columns (ld floats apart): zhat = (z - mean) / sqrt(var + eps),
// g = dy * gamma, dz = (g - mean(g) - zhat * mean(g * zhat)) / sqrt(var + eps); per-workgroup partial sums of
// dgamma = sum dy * zhat and dbeta = sum dy ([grid][2][D], reduced by magnet_reduce_partials_f32).  One wave per row.
__global__ void __launch_bounds__(256) k_layernorm_bwd(const float* __restrict__ z, int ldz, const float* __restrict__ dy, int ldd,
                                                       const float* __restrict__ gamma, long rows, int D, float eps,
                                                       float* __restrict__ dz, int ldo, float* __restrict__ part,
                                                       float* __restrict__ amax /* [grid] block maxima of |dz|, or nullptr */,
                                                       const int32_t* __restrict__ tgt /* nullptr, or: row r's dy is the backward of a CSR
                                                       mean - dy[tgt[r]] / deg(tgt[r]) (k_edge_mean_bwd_d fused into the load) */,
                                                       const int32_t* __restrict__ rowptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float sg[4][2][256];
    float dzmax = 0.f;
    float ag[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};              // this lane's columns lane, lane+64, ...
    const float invD = 1.0f / (float)D;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        float zv[4], dv[4], s = 0.f;
        long rd = r;
        float dsc = 1.0f;
        if (tgt) {
            const int t = tgt[r];
            rd = t;
            dsc = 1.0f / (float)(rowptr[t + 1] - rowptr[t]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            zv[k] = c < D ? z[r * ldz + c] : 0.f;
            dv[k] = c < D ? dy[rd * ldd + c] * dsc : 0.f;
            s += zv[k];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s * invD;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            const float t = c < D ? zv[k] - mean : 0.f;
            v += t * t;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        const float rstd = 1.0f / sqrtf(v * invD + eps);
        float m1 = 0.f, m2 = 0.f, zh[4], g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            zh[k] = c < D ? (zv[k] - mean) * rstd : 0.f;
            g[k] = c < D ? dv[k] * gamma[c] : 0.f;
            m1 += g[k];
            m2 += g[k] * zh[k];
            ag[k] += dv[k] * zh[k];
            ab[k] += dv[k];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m1 += __shfl_xor(m1, off);
            m2 += __shfl_xor(m2, off);
        }
        m1 *= invD;
        m2 *= invD;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            if (c < D) {
                const float v = (g[k] - m1 - zh[k] * m2) * rstd;
                dz[r * ldo + c] = v;
                dzmax = fmaxf(dzmax, fabsf(v));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sg[wave][0][lane + 64 * k] = ag[k];
        sg[wave][1][lane + 64 * k] = ab[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int w = c / D, col = c % D;
        part[(size_t)blockIdx.x * 2 * D + c] = sg[0][w][col] + sg[1][w][col] + sg[2][w][col] + sg[3][w][col];
    }
    if (amax) {                                  // what the next kernel (k_linear_bwd) needs to pick its power-of-two scale: no pass of its own
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, off));
        __syncthreads();
        if (lane == 0) sg[0][0][wave] = dzmax;
        __syncthreads();
        if (threadIdx.x == 0) amax[blockIdx.x] = fmaxf(fmaxf(sg[0][0][0], sg[0][0][1]), fmaxf(sg[0][0][2], sg[0][0][3]));
    }
}

// The same for D in {32, 64, 128} with 16-byte accesses: a row takes LPR = D / 4 lanes (one float4 each), a wave works on 64 / LPR rows at
// once.  The one-row-per-wave kernel above moves 256 bytes per load instruction at D = 64 and has one row per wave in flight: 2.6 TB/s
// on the [2.2 M, 64] arrays of MAgNet[CNN]'s chains.  Same outputs (other summation order inside a row and across rows).
template <int LPR>
__global__ void __launch_bounds__(256) k_layernorm_bwd_v4(const float* __restrict__ z, int ldz, const float* __restrict__ dy, int ldd,
                                                          const float* __restrict__ gamma, long rows, float eps, float* __restrict__ dz,
                                                          int ldo, float* __restrict__ part, float* __restrict__ amax,
                                                          const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr) {
    constexpr int D = 4 * LPR, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, c4 = lane % LPR;
    __shared__ float sg[4][2][128];
    const f32x4 gm = {gamma[4 * c4], gamma[4 * c4 + 1], gamma[4 * c4 + 2], gamma[4 * c4 + 3]};   // (a parameter: a view into the optimizer's flat
                                                                                              // buffer, 4-byte aligned only)
    f32x4 ag = {0, 0, 0, 0}, ab = {0, 0, 0, 0};
    float dzmax = 0.f;
    const float invD = 1.0f / (float)D;
    for (long g0 = ((long)blockIdx.x * 4 + wave) * RPW; g0 < rows; g0 += (long)gridDim.x * 4 * RPW) {
        const long r = g0 + sub;
        const bool ok = r < rows;
        const long rc = ok ? r : rows - 1;
        long rd = rc;
        float dsc = 1.0f;
        if (tgt) {
            const int t = tgt[rc];
            rd = t;
            dsc = 1.0f / (float)(rowptr[t + 1] - rowptr[t]);
        }
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + rc * ldz + 4 * c4);
        f32x4 dv = *reinterpret_cast<const f32x4*>(dy + rd * ldd + 4 * c4);
        dv *= ok ? dsc : 0.f;                                               // (rows past the end: no contribution to dgamma / dbeta)
        float s = (zv[0] + zv[1]) + (zv[2] + zv[3]);
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s * invD;
        const f32x4 zc = zv - mean;
        float v = (zc[0] * zc[0] + zc[1] * zc[1]) + (zc[2] * zc[2] + zc[3] * zc[3]);
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
        const float rstd = 1.0f / sqrtf(v * invD + eps);
        const f32x4 zh = zc * rstd, g = dv * gm;
        float m1 = (g[0] + g[1]) + (g[2] + g[3]);
        float m2 = (g[0] * zh[0] + g[1] * zh[1]) + (g[2] * zh[2] + g[3] * zh[3]);
        ag += dv * zh;
        ab += dv;
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) {
            m1 += __shfl_xor(m1, off);
            m2 += __shfl_xor(m2, off);
        }
        m1 *= invD;
        m2 *= invD;
        const f32x4 o = (g - m1 - zh * m2) * rstd;
        if (ok) {
            *reinterpret_cast<f32x4*>(dz + r * ldo + 4 * c4) = o;
            dzmax = fmaxf(fmaxf(dzmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        }
    }
    // the wave's RPW row groups hold partial sums of the same columns: combine them, then the four waves (as above)
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ag[c] += __shfl_xor(ag[c], off);
            ab[c] += __shfl_xor(ab[c], off);
        }
    if (sub == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sg[wave][0][4 * c4 + c] = ag[c];
            sg[wave][1][4 * c4 + c] = ab[c];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int w = c / D, col = c % D;
        part[(size_t)blockIdx.x * 2 * D + c] = sg[0][w][col] + sg[1][w][col] + sg[2][w][col] + sg[3][w][col];
    }
    if (amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, off));
        __syncthreads();
        if (lane == 0) sg[0][0][wave] = dzmax;
        __syncthreads();
        if (threadIdx.x == 0) amax[blockIdx.x] = fmaxf(fmaxf(sg[0][0][0], sg[0][0][1]), fmaxf(sg[0][0][2], sg[0][0][3]));
    }
}

// dm[e][0:D] = dagg[tgt[e]][0:D] / deg(tgt[e])      (backward of the CSR mean, any width D % 4 == 0)
__global__ void k_edge_mean_bwd_d(const float* __restrict__ dagg, const int32_t* __restrict__ tgt,
                                  const int32_t* __restrict__ rowptr, long E, int D, float* __restrict__ dm, int ldm) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q4 = D / 4;
    if (i >= E * q4) return;
    const long e = i / q4;
    const int c = (int)(i % q4) * 4;
    const int t = tgt[e];
    const float inv = 1.0f / (float)(rowptr[t + 1] - rowptr[t]);
    *reinterpret_cast<f32x4*>(dm + e * ldm + c) = *reinterpret_cast<const f32x4*>(dagg + (size_t)t * D + c) * inv;
}

// out[r][ocol : ocol+W] (+)= scale * sum_{k in row r} x[perm ? perm[k] : k][xcol : xcol+W]   (scatter-add of a column
// block of row gradients to the rows they were gathered from; W % 4 == 0).  One wave per output row.
__global__ void __launch_bounds__(256) k_segment_sum_cols(const float* __restrict__ x, int ldx, int xcol, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ perm, int N, int W, float scale,
                                                          float* __restrict__ out, int ldo, int ocol, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int s = rowptr[r], e = rowptr[r + 1];
    const int q4 = W / 4;                                             // float4 columns
    const int per = 64 / (q4 < 64 ? q4 : 64);                         // rows summed concurrently by the wave
    const int sub = lane / q4, c4 = lane % q4;
    for (int c0 = 0; c0 < q4; c0 += 64) {                            // (W <= 256: one pass)
        f32x4 acc = {0, 0, 0, 0};
        const int cc = c0 + c4;
        // (as k_segment_sum128: the permuted positions of up to 64 entries of the row come from ONE load and are handed out with
        //  ds_bpermute, four row loads in flight per lane group - not an index load in front of every row load; same entries per
        //  lane group in the same order)
        const bool on = sub < per && cc < q4;
        for (int base = s; base < e; base += 64) {
            const int cnt = min(64, e - base);
            const int mine = lane < cnt ? (perm ? perm[base + lane] : base + lane) : 0;
            for (int k0 = 0; k0 < cnt; k0 += 4 * per) {
                f32x4 v[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + sub + u * per;
                    ok[u] = on && kk < cnt;
                    const long row = __shfl(mine, ok[u] ? kk : 0);
                    v[u] = *reinterpret_cast<const f32x4*>(x + row * ldx + xcol + 4 * (cc < q4 ? cc : 0));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] += ok[u] ? v[u][c] : 0.f;
            }
        }
        // combine the `per` partial sums (lanes c4 + q4 * sub)
        for (int off = q4; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], off);
        }
        if (sub == 0 && cc < q4) {
            float* o = out + (size_t)r * ldo + ocol + 4 * cc;
            f32x4 v = acc * scale;
            if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = v;
        }
    }
}

// The same for up to MAGNET_SEG_RUNS gathers of the SAME source rows at once (the taps of a convolution: one inverse CSR per tap, each
// with its own column block of x): out[r] (+)= scale * sum_u sum_{k in row r of run u} x[perm_u[k]][xcol_u ..].  One launch instead of one
// per tap, and - what makes it faster, not just fewer launches - the first entry of every run is fetched before any of them is consumed
// (a wave is an in-order machine: run after run it would pay rowptr -> perm -> row as three dependent round trips NINE times).
// Runs are added in index order: deterministic.
struct SegRuns {
    const int32_t* rowptr[MAGNET_SEG_RUNS];
    const int32_t* perm[MAGNET_SEG_RUNS];
    int32_t xcol[MAGNET_SEG_RUNS];
    int32_t n;
};
__global__ void __launch_bounds__(256) k_segment_sum_cols_runs(const float* __restrict__ x, int ldx, const SegRuns runs, int N, int W,
                                                               float scale, float* __restrict__ out, int ldo, int ocol, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int q4 = W / 4;
    const int per = 64 / (q4 < 64 ? q4 : 64);
    const int sub = lane / q4, c4 = lane % q4;
    const bool on = sub < per && c4 < q4;                               // (W <= 256: one pass over the float4 columns)
    int k0[MAGNET_SEG_RUNS], e[MAGNET_SEG_RUNS];
#pragma unroll
    for (int u = 0; u < MAGNET_SEG_RUNS; ++u) {
        const bool have = u < runs.n;
        k0[u] = have ? runs.rowptr[u][r] + sub : 0;
        e[u] = have ? runs.rowptr[u][r + 1] : 0;
    }
    long row[MAGNET_SEG_RUNS];
#pragma unroll
    for (int u = 0; u < MAGNET_SEG_RUNS; ++u) row[u] = (on && k0[u] < e[u]) ? (runs.perm[u] ? (long)runs.perm[u][k0[u]] : (long)k0[u]) : -1;
    f32x4 part[MAGNET_SEG_RUNS];
#pragma unroll
    for (int u = 0; u < MAGNET_SEG_RUNS; ++u) {
        part[u] = f32x4{0, 0, 0, 0};
        if (row[u] >= 0) part[u] = *reinterpret_cast<const f32x4*>(x + row[u] * ldx + runs.xcol[u] + 4 * c4);
    }
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < MAGNET_SEG_RUNS; ++u) {
        if (on && u < runs.n)                                           // further entries of this row in run u (rare for a tap index)
            for (int k = k0[u] + per; k < e[u]; k += per) {
                const long rw = runs.perm[u] ? runs.perm[u][k] : k;
                part[u] += *reinterpret_cast<const f32x4*>(x + rw * ldx + runs.xcol[u] + 4 * c4);
            }
        acc += part[u];
    }
    for (int off = q4; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], off);
    }
    if (sub == 0 && c4 < q4) {
        float* o = out + (size_t)r * ldo + ocol + 4 * c4;
        f32x4 v = acc * scale;
        if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
}

// ---- backward of k_inr_knn_blend (magnet_kernels.hip; models/magnet_gnn.py:254-280).  Forward:
// z[q, i, c] = om0 * l0 + om1 * l1,  l_j = base[2q + j][c] + wv[c] * vals[nn2[2q + j]][i] + wt[c] * t[b][i],
// om_j = the normalised geometric blend weights (functions of the coordinates only).
// Outputs: dbase[2q + j][c] = om_j sum_i dz;  dvq[2q + j][i] = om_j sum_c dz wv[c]  (per (query, neighbour); the caller
// scatter-adds it to the LR nodes);  per-workgroup partials of dwv[c] = sum dz (om0 v0 + om1 v1), dwt[c] = sum dz t.
__device__ __forceinline__ void blend_weights(float w0, float w1, int mode, int L, float& o0, float& o1) {
    float a0, a1;
    if (mode == 0) { a0 = w1; a1 = w0; }                             // 'area': weights swapped
    else if (mode == 1) { a0 = 1.f / w0; a1 = 1.f / w1; }            // 'knn'
    else { const float p = 1.f - L * w0, q = 1.f - L * w1; a0 = p * p * p; a1 = q * q * q; }   // 'sph'
    o0 = a0 / (a0 + a1);
    o1 = a1 / (a0 + a1);
}
__global__ void __launch_bounds__(256) k_inr_knn_blend_bwd(const float* __restrict__ dz, const float* __restrict__ w,
                                                           const float* __restrict__ vals, const int32_t* __restrict__ nn2,
                                                           const float* __restrict__ t, int t_ld, const float* __restrict__ wv,
                                                           int L, int Nq, long nq_total, int T, int C, int mode,
                                                           float* __restrict__ dbase, float* __restrict__ dvq,
                                                           float* __restrict__ part /* [grid][2][C] */) {
    // one wave per query, lanes over channels (C <= 256); T <= 64 time steps kept in LDS per wave
    __shared__ float sv[4][2][64], sdv[4][2][64], sp[4][2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float awv[4] = {0, 0, 0, 0}, awt[4] = {0, 0, 0, 0};
    for (long q = (long)blockIdx.x * 4 + wave; q < nq_total; q += (long)gridDim.x * 4) {
        const int b = (int)(q / Nq);
        float o0, o1;
        blend_weights(w[2 * q], w[2 * q + 1], mode, L, o0, o1);
        const int n0 = nn2[2 * q], n1 = nn2[2 * q + 1];
        if (lane < T) {
            sv[wave][0][lane] = vals[(size_t)n0 * T + lane];
            sv[wave][1][lane] = vals[(size_t)n1 * T + lane];
        }
        wave_lds_sync();
        float db[4] = {0, 0, 0, 0};
        for (int i = 0; i < T; ++i) {
            const float tt = t[(size_t)b * t_ld + i];
            const float vmix = o0 * sv[wave][0][i] + o1 * sv[wave][1][i];
            float dvi = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    const float g = dz[((size_t)q * T + i) * C + c];
                    db[k] += g;
                    awv[k] += g * vmix;
                    awt[k] += g * tt;
                    dvi += g * wv[c];
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dvi += __shfl_xor(dvi, off);
            if (lane == 0) {
                sdv[wave][0][i] = o0 * dvi;
                sdv[wave][1][i] = o1 * dvi;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                dbase[(size_t)(2 * q) * C + c] = o0 * db[k];
                dbase[(size_t)(2 * q + 1) * C + c] = o1 * db[k];
            }
        }
        wave_lds_sync();
        if (lane < T) {
            dvq[(size_t)(2 * q) * T + lane] = sdv[wave][0][lane];
            dvq[(size_t)(2 * q + 1) * T + lane] = sdv[wave][1][lane];
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sp[wave][0][lane + 64 * k] = awv[k];
        sp[wave][1][lane + 64 * k] = awt[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c / C, col = c % C;
        part[(size_t)blockIdx.x * 2 * C + c] = sp[0][which][col] + sp[1][which][col] + sp[2][which][col] + sp[3][which][col];
    }
}

// ---- backward of k_euler_update: out[b][k][node] = last[r] + (t[b][T+k] - t[b][T-1]) * ret[r][k],  r = b*n + node
__global__ void k_euler_update_bwd(const float* __restrict__ dout, const float* __restrict__ t, int t_ld, int B, int n, int T,
                                   int T_out, float* __restrict__ dret, int ld, float* __restrict__ dlast) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (long)B * n) return;
    const int b = (int)(r / n), node = (int)(r % n);
    float s = 0.f;
    for (int k = 0; k < T_out; ++k) {
        const float g = dout[((size_t)b * T_out + k) * n + node];
        s += g;
        dret[r * ld + k] = (t[(size_t)b * t_ld + T + k] - t[(size_t)b * t_ld + T - 1]) * g;
    }
    dlast[r] = s;
}

// ------------------------------------------------------------------------------------ C entry points
extern "C" int magnet_mlp_materialize_input_f32(const MagnetMlpArgs* args, float* x0, int ldx, void* stream) {
    BEGIN();
    if (!args || args->M <= 0) return args && args->M == 0 ? 0 : -1;
    if (args->nkt < 1 || args->nkt > MAGNET_MAX_KT || ldx < args->nkt * 32) return -2;
    hipLaunchKernelGGL(k_materialize_ktiles, dim3(nb((long)args->M * args->nkt * 8, 256)), dim3(256), 0, S_, *args, x0, ldx);
    END();
}
// Host only: the k-tile descriptors of `ntaps` gathered taps of ONE source block - tap t, 32-column block k is
// src[idx[t * n_rows + row]][32 k .. 32 k + 32) - written into `a` (nkt, kt[]; nothing else is touched).  A convolution as a
// gathered-tap Linear has up to 36 of these; filling them field by field from Python cost 35 us per launch, 30 ms of a magnet_cnn_2d
// training step at the reference's batch size.
extern "C" int magnet_mlp_args_set_taps(MagnetMlpArgs* a, const float* src, int ld, int width, const int32_t* idx, long n_rows,
                                        int ntaps) {
    if (!a || !src || !idx || width % 32 || width < 32 || ntaps < 1 || ntaps * (width / 32) > MAGNET_MAX_KT || ld < width || ld % 4) return -2;
    const int w32 = width / 32;
    a->nkt = ntaps * w32;
    for (int t = 0; t < ntaps; ++t)
        for (int k = 0; k < w32; ++k) {
            MagnetKTile& kt = a->kt[t * w32 + k];
            kt.n = 1;
            kt.s[0].p = src;
            kt.s[0].idx = idx + (size_t)t * n_rows;
            kt.s[0].ld = ld;
            kt.s[0].col = 32 * k;
            kt.s[0].scale = 1.0f;
            kt.s[0].rdiv = 1;
            kt.s[1] = MagnetSrc{};
        }
    return 0;
}
extern "C" int magnet_relu_bwd_f32(const float* dy, const float* y, float* dz, long n, void* stream) {
    BEGIN();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_relu_bwd, dim3(nb((n + 3) / 4, 256)), dim3(256), 0, S_, dy, y, dz, n);
    END();
}
// part: [min(2048, ceil(rows/4))][2][D] partial (dgamma | dbeta) sums.  (2048 workgroups = 8 waves per SIMD: a wave's row is one
// dependent load -> reduce -> store chain, so the kernel is as fast as the number of rows in flight; 1024 left narrow rows (D = 64:
// 256 bytes per load) at 3.2 TB/s)
extern "C" int magnet_layernorm_bwd_f32(const float* z, int ldz, const float* dy, int ldd, const float* gamma, long rows, int D,
                                        float eps, float* dz, int ldo, float* part, void* stream) {
    BEGIN();
    if (rows <= 0) return 0;
    if (D < 1 || D > 256) return -2;
    const unsigned grid = nb(rows, 4) < 2048u ? nb(rows, 4) : 2048u;
    hipLaunchKernelGGL(k_layernorm_bwd, dim3(grid), dim3(256), 0, S_, z, ldz, dy, ldd, gamma, rows, D, eps, dz, ldo, part, (float*)nullptr,
                       (const int32_t*)nullptr, (const int32_t*)nullptr);
    END();
}
// the same, and amax [min(2048, ceil(rows/4))] = per-workgroup maxima of |dz| (the amax_in of magnet_linear_bwd128_f32 / 64); tgt /
// rowptr != NULL: the rows are CSR-ordered edges and dy [n_nodes][ldd] is the gradient of their per-target MEAN - row r takes
// dy[tgt[r]] / deg(tgt[r]) (magnet_edge_mean_bwd_cols_f32 fused into the load: the [rows][D] array in between is never written)
extern "C" int magnet_layernorm_bwd_amax_f32(const float* z, int ldz, const float* dy, int ldd, const float* gamma, long rows, int D,
                                             float eps, float* dz, int ldo, float* part, float* amax, const int32_t* tgt,
                                             const int32_t* rowptr, void* stream) {
    BEGIN();
    if (rows <= 0) return 0;
    if (D < 1 || D > 256 || (tgt && !rowptr)) return -2;
    const unsigned grid = nb(rows, 4) < 2048u ? nb(rows, 4) : 2048u;
    const bool v4 = (D == 32 || D == 64 || D == 128) && !((ldz | ldd | ldo) & 3) &&
                    !(((uintptr_t)z | (uintptr_t)dy | (uintptr_t)dz) & 15);
    if (v4 && D == 32)
        hipLaunchKernelGGL(k_layernorm_bwd_v4<8>, dim3(grid), dim3(256), 0, S_, z, ldz, dy, ldd, gamma, rows, eps, dz, ldo, part, amax, tgt, rowptr);
    else if (v4 && D == 64)
        hipLaunchKernelGGL(k_layernorm_bwd_v4<16>, dim3(grid), dim3(256), 0, S_, z, ldz, dy, ldd, gamma, rows, eps, dz, ldo, part, amax, tgt, rowptr);
    else if (v4)
        hipLaunchKernelGGL(k_layernorm_bwd_v4<32>, dim3(grid), dim3(256), 0, S_, z, ldz, dy, ldd, gamma, rows, eps, dz, ldo, part, amax, tgt, rowptr);
    else
        hipLaunchKernelGGL(k_layernorm_bwd, dim3(grid), dim3(256), 0, S_, z, ldz, dy, ldd, gamma, rows, D, eps, dz, ldo, part, amax, tgt, rowptr);
    END();
}
extern "C" int magnet_edge_mean_bwd_cols_f32(const float* dagg, const int32_t* tgt, const int32_t* rowptr, long n_edges, int D,
                                             float* dm, int ldm, void* stream) {
    BEGIN();
    if (n_edges <= 0) return 0;
    if (D % 4) return -2;
    hipLaunchKernelGGL(k_edge_mean_bwd_d, dim3(nb(n_edges * (D / 4), 256)), dim3(256), 0, S_, dagg, tgt, rowptr, n_edges, D, dm, ldm);
    END();
}
extern "C" int magnet_segment_sum_cols_runs_f32(const float* x, int ldx, const int32_t* const* rowptrs, const int32_t* const* perms,
                                                const int32_t* xcols, int n_runs, int n_rows, int width, float scale, float* out,
                                                int ldo, int ocol, int accumulate, void* stream) {
    BEGIN();
    if (n_rows <= 0 || width <= 0 || n_runs <= 0) return 0;
    if (n_runs > MAGNET_SEG_RUNS || width % 4 || width > 256 || (64 % (width / 4 < 64 ? width / 4 : 64))) return -2;
    SegRuns runs;
    for (int u = 0; u < MAGNET_SEG_RUNS; ++u) {
        runs.rowptr[u] = u < n_runs ? rowptrs[u] : nullptr;
        runs.perm[u] = u < n_runs ? perms[u] : nullptr;
        runs.xcol[u] = u < n_runs ? xcols[u] : 0;
    }
    runs.n = n_runs;
    hipLaunchKernelGGL(k_segment_sum_cols_runs, dim3(nb(n_rows, 4)), dim3(256), 0, S_, x, ldx, runs, n_rows, width, scale, out, ldo, ocol,
                       accumulate);
    END();
}
extern "C" int magnet_segment_sum_cols_f32(const float* x, int ldx, int xcol, const int32_t* rowptr, const int32_t* perm,
                                           int n_rows, int width, float scale, float* out, int ldo, int ocol, int accumulate,
                                           void* stream) {
    BEGIN();
    if (n_rows <= 0 || width <= 0) return 0;
    if (width % 4 || width > 256 || (64 % (width / 4 < 64 ? width / 4 : 64))) return -2;   // float4 columns must divide the wave
    hipLaunchKernelGGL(k_segment_sum_cols, dim3(nb(n_rows, 4)), dim3(256), 0, S_, x, ldx, xcol, rowptr, perm, n_rows, width, scale,
                       out, ldo, ocol, accumulate);
    END();
}
// part: [min(1024, ceil(nq/4))][2][C] partial (dwv | dwt) sums
extern "C" int magnet_inr_knn_blend_bwd_f32(const float* dz, const float* w, const float* vals, const int32_t* nn2,
                                            const float* t, int t_ld, const float* wv, int n_graphs, int L, int Nq, int T, int C,
                                            int mode, float* dbase, float* dvq, float* part, void* stream) {
    BEGIN();
    const long nq = (long)n_graphs * Nq;
    if (nq <= 0) return 0;
    if (T > 64 || C > 256) return -2;
    const unsigned grid = nb(nq, 4) < 1024u ? nb(nq, 4) : 1024u;
    hipLaunchKernelGGL(k_inr_knn_blend_bwd, dim3(grid), dim3(256), 0, S_, dz, w, vals, nn2, t, t_ld, wv, L, Nq, nq, T, C, mode,
                       dbase, dvq, part);
    END();
}
extern "C" int magnet_euler_update_bwd_f32(const float* dout, const float* t, int t_ld, int n_graphs, int nodes_per_graph, int T,
                                           int T_out, float* dret, int ld_ret, float* dlast, void* stream) {
    BEGIN();
    const long rows = (long)n_graphs * nodes_per_graph;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_euler_update_bwd, dim3(nb(rows, 256)), dim3(256), 0, S_, dout, t, t_ld, n_graphs, nodes_per_graph, T, T_out,
                       dret, ld_ret, dlast);
    END();
}

// ---- backward of k_inr_grid_blend (magnet_kernels.hip): z = sum_s p_s * om_s with om_s = area[opposite(s)] / sum area
// (functions of the coordinates only)  ->  dp[((bq*S + s)*T + i)][c] = dz[(bq*T + i)][c] * om_s
__global__ void k_inr_grid_blend_bwd(const float* __restrict__ dz, const float* __restrict__ area, int T, int C, int S,
                                     long total /* = n_rows * S * T * C */, float* __restrict__ dp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int ti = (int)((i / C) % T);
    const long rs = i / ((long)C * T);                                 // bq*S + s
    const int s = (int)(rs % S);
    const long bq = rs / S;
    float om;
    if (S == 2) {
        const float a0 = area[bq * 2], a1 = area[bq * 2 + 1];
        om = (s == 0 ? a1 : a0) / (a1 + a0);
    } else {
        const float a0 = area[bq * 4], a1 = area[bq * 4 + 1], a2 = area[bq * 4 + 2], a3 = area[bq * 4 + 3];
        const float tot = ((a0 + a1) + a2) + a3;
        om = (s == 0 ? a3 : s == 1 ? a2 : s == 2 ? a1 : a0) / tot;
    }
    dp[i] = dz[(bq * T + ti) * C + c] * om;
}
extern "C" int magnet_inr_grid_blend_bwd_f32(const float* dz, const float* area, int n_rows, int T, int C, int nshift, float* dp,
                                             void* stream) {
    BEGIN();
    const long total = (long)n_rows * nshift * T * C;
    if (total <= 0) return 0;
    if (nshift != 2 && nshift != 4) return -2;
    hipLaunchKernelGGL(k_inr_grid_blend_bwd, dim3(nb(total, 256)), dim3(256), 0, S_, dz, area, T, C, nshift, total, dp);
    END();
}
