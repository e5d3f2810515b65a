// MAgNet[GNN] / MAgNet[CNN] feature builders, Euler update and the two INR decoders' gather /
// blend stages.  The MLPs between them run in magnet_mlp_f32 (mlp_kernels.hip).
#include "common.cuh"

// node features [u (T) | x (d) | t_last[k mod B] | 0-pad]  (models/magnet_gnn.py:298-302; the
// repeat(N,1) tile gives node k the time of sample k mod B - quirk Q5, reproduced on purpose)
__global__ void k_node_features(const float* __restrict__ u, const float* __restrict__ x,
                                const float* __restrict__ t_last, int B, long n_total, int T, int d, int ld,
                                float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total * ld) return;
    const long n = i / ld;
    const int c = (int)(i % ld);
    float v = 0.f;
    if (c < T) v = u[n * T + c];
    else if (c < T + d) v = x[n * d + (c - T)];
    else if (c == T + d) v = t_last[n % B];
    out[i] = v;
}

// out[b][k][i] = last[b*n + i] + (t[b][T+k] - t[b][T-1]) * ret[(b*n + i)*ld + k]
// (models/magnet_gnn.py:359-369)
__global__ void k_euler_update(const float* __restrict__ ret, int ld, const float* __restrict__ last,
                               const float* __restrict__ t, int t_ld, int B, int n, int T, int T_out,
                               float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T_out * n) return;
    const int node = (int)(i % n);
    const int k = (int)((i / n) % T_out);
    const int b = (int)(i / ((long)n * T_out));
    const float delta = t[(size_t)b * t_ld + T + k] - t[(size_t)b * t_ld + T - 1];
    const size_t r = (size_t)b * n + node;
    out[i] = last[r] + delta * ret[r * ld + k];
}

// ---- k-NN INR (models/magnet_gnn.py:247-268): per query and j in {0,1} (only the two nearest are
// blended, Q9): gather row of the neighbour, rel = X[nb] - hr, w = ||rel||_2 ** 2
__global__ void k_inr_knn_prepare(const float* __restrict__ lr, const float* __restrict__ hr,
                                  const int32_t* __restrict__ nn, int k, int d, long nq_total,
                                  int32_t* __restrict__ nn2, float* __restrict__ rel, float* __restrict__ w) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;   // r = q*2 + j
    if (r >= nq_total * 2) return;
    const long q = r >> 1;
    const int j = (int)(r & 1);
    const int nb = nn[q * k + j];
    nn2[r] = nb;
    float s = 0.f;
    for (int c = 0; c < 32; ++c) {
        float v = 0.f;
        if (c < d) {
            v = lr[(size_t)nb * d + c] - hr[q * d + c];
            s = __fadd_rn(s, __fmul_rn(v, v));
        }
        rel[r * 32 + c] = v;
    }
    const float nrm = sqrtf(s);
    w[r] = nrm * nrm;
}

// z[(q, i)][c] = blend_j( base[(q, j)][c] + wv[c] * vals[nb_j][i] + wt[c] * t[b][i] )
__global__ void k_inr_knn_blend(const float* __restrict__ base, const float* __restrict__ w,
                                const float* __restrict__ vals, const int32_t* __restrict__ nn2,
                                const float* __restrict__ t, int t_ld, const float* __restrict__ wv,
                                const float* __restrict__ wt, int L, int Nq, int T, int C, int mode, long total,
                                float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int ti = (int)((i / C) % T);
    const long q = i / ((long)C * T);
    const int b = (int)(q / Nq);
    const float tt = t[(size_t)b * t_ld + ti];
    const float l0 = base[(2 * q) * C + c] + wv[c] * vals[(size_t)nn2[2 * q] * T + ti] + wt[c] * tt;
    const float l1 = base[(2 * q + 1) * C + c] + wv[c] * vals[(size_t)nn2[2 * q + 1] * T + ti] + wt[c] * tt;
    float w0 = w[2 * q], w1 = w[2 * q + 1];
    float r;
    if (mode == 0) {
        r = (l0 * w1 + l1 * w0) / (w1 + w0);                    // 'area': weights swapped
    } else {
        if (mode == 1) { w0 = 1.f / w0; w1 = 1.f / w1; }          // 'knn'
        else { const float a = 1.f - L * w0, bb = 1.f - L * w1; w0 = a * a * a; w1 = bb * bb * bb; }   // 'sph'
        r = (l0 * w0 + l1 * w1) / (w1 + w0);
    }
    z[i] = r;
}

// the same with C % 4 == 0: one thread per (query, time step, 4 channels), 16-byte loads and stores, the per-(query, step) scalars once per
// thread; every element goes through the expression above, bit for bit (335 MB written per call at 256^2 queries: 307 -> see DESIGN.md 9.9)
__global__ void k_inr_knn_blend4(const float* __restrict__ base, const float* __restrict__ w,
                                 const float* __restrict__ vals, const int32_t* __restrict__ nn2,
                                 const float* __restrict__ t, int t_ld, const float* __restrict__ wv,
                                 const float* __restrict__ wt, int L, int Nq, int T, int C4, int mode, long total4,
                                 float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % C4);
    const int ti = (int)((i / C4) % T);
    const long q = i / ((long)C4 * T);
    const int b = (int)(q / Nq);
    const float tt = t[(size_t)b * t_ld + ti];
    const float v0 = vals[(size_t)nn2[2 * q] * T + ti], v1 = vals[(size_t)nn2[2 * q + 1] * T + ti];
    const f32x4 b0 = reinterpret_cast<const f32x4*>(base)[(2 * q) * C4 + c], b1 = reinterpret_cast<const f32x4*>(base)[(2 * q + 1) * C4 + c];
    const f32x4 wv4 = reinterpret_cast<const f32x4*>(wv)[c], wt4 = reinterpret_cast<const f32x4*>(wt)[c];
    float w0 = w[2 * q], w1 = w[2 * q + 1];
    if (mode == 1) { w0 = 1.f / w0; w1 = 1.f / w1; }
    else if (mode == 2) { const float a = 1.f - L * w0, bb = 1.f - L * w1; w0 = a * a * a; w1 = bb * bb * bb; }
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float l0 = b0[k] + wv4[k] * v0 + wt4[k] * tt;
        const float l1 = b1[k] + wv4[k] * v1 + wt4[k] * tt;
        r[k] = mode == 0 ? (l0 * w1 + l1 * w0) / (w1 + w0) : (l0 * w0 + l1 * w1) / (w1 + w0);
    }
    reinterpret_cast<f32x4*>(z)[i] = r;
}

// ---- grid INR (models/magnet_cnn.py:249-276, models/magnet_cnn_2d.py:248-283): per query and
// shift: nearest cell (grid_sample nearest / border / align_corners=False), rel, cell*W, area,
// and per input step the tail [q_inp_i | rel | cell*W | t_i | 0-pad] + the gather row index.
__device__ __forceinline__ int nearest_cell(float c, int W) {
    float u = ((c + 1.f) * (float)W - 1.f) / 2.f;
    u = fminf((float)(W - 1), fmaxf(u, 0.f));
    return (int)rintf(u);                                         // round half to even == nearbyint
}
// Rows written 16 bytes per lane, lanes along the row: thread = (row (r, i), column quad); the per-(query, shift) values are recomputed by
// the 8 T threads that share them (a few flops) - one thread per (query, shift) writing T x 128 contiguous bytes at a 128-byte lane stride
// (64 partially written lines per store instruction) took 238 us at 256^2 queries.
__global__ void k_inr_grid_prepare(const float* __restrict__ coords, const float* __restrict__ cell,
                                   const float* __restrict__ x_t, const float* __restrict__ t, int t_ld, int B,
                                   int Nq, int T, int W, int d, int32_t* __restrict__ cell_idx,
                                   float* __restrict__ tail, float* __restrict__ area) {
    const int S = d == 2 ? 4 : 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;  // gid = ((r * T) + i) * 8 + quad, r = (b*Nq + q)*S + s
    const int quad = (int)(gid & 7);
    const long row = gid >> 3;
    if (row >= (long)B * Nq * S * T) return;
    const int i = (int)(row % T);
    const long r = row / T;
    const int s = (int)(r % S);
    const long bq = r / S;
    const int b = (int)(bq / Nq);
    const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
    const float c0 = (float)(-1.0 + 1.0 / (double)W), c1 = (float)(2.0 / (double)W);
    const long cells = d == 2 ? (long)W * W : W;
    int idx[2] = {0, 0};
    float rel[2] = {0.f, 0.f}, cw[2] = {0.f, 0.f};
    for (int a = 0; a < d; ++a) {
        const int v = d == 2 ? (a == 0 ? (s >> 1) * 2 - 1 : (s & 1) * 2 - 1) : s * 2 - 1;   // (-,-),(-,+),(+,-),(+,+)
        const float off = (float)((double)v / (double)W + 1e-6);
        const float co = coords[bq * d + a];
        const float c = fminf(hi, fmaxf(lo, co + off));
        idx[a] = nearest_cell(c, W);
        const float qc = __fadd_rn(c0, __fmul_rn(c1, (float)idx[a]));
        rel[a] = (co - qc) * (float)W;
        cw[a] = cell[bq * d + a] * (float)W;
    }
    const long flat = d == 2 ? (long)idx[0] * W + idx[1] : idx[0];
    if (quad == 0 && i == 0) area[r] = d == 2 ? fabsf(rel[0] * rel[1]) + 1e-9f : fabsf(rel[0]);
    if (quad == 0) cell_idx[row] = (int32_t)((long)b * cells + flat);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * quad + k;
        float v = 0.f;
        if (c == 0) v = x_t[((size_t)b * T + i) * cells + flat];
        else if (c <= d) v = rel[c - 1];
        else if (c <= 2 * d) v = cw[c - 1 - d];
        else if (c == 1 + 2 * d) v = t[(size_t)b * t_ld + i];
        o[k] = v;
    }
    *reinterpret_cast<f32x4*>(tail + row * 32 + 4 * quad) = o;
}

// The 2-D prepare for the decoder whose epilogue does the area blend itself (EPI_QUADBLEND): the four shifts of a (query, step) are
// CONSECUTIVE rows - row = ((b Nq + q) T + i) 4 + s - and every row carries its blend weight area[opposite shift] / sum of the four areas
// (models/magnet_cnn_2d.py:285-290).  Same per-row values as k_inr_grid_prepare.
__global__ void k_inr_grid_prepare_blend(const float* __restrict__ coords, const float* __restrict__ cell,
                                         const float* __restrict__ x_t, const float* __restrict__ t, int t_ld, int B, int Nq,
                                         int T, int W, int32_t* __restrict__ cell_idx, float* __restrict__ tail,
                                         float* __restrict__ blend_w) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;  // gid = row * 8 + quad
    const int quad = (int)(gid & 7);
    const long row = gid >> 3;
    if (row >= (long)B * Nq * T * 4) return;
    const int s = (int)(row & 3);
    const int i = (int)((row >> 2) % T);
    const long bq = (row >> 2) / T;
    const int b = (int)(bq / Nq);
    const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
    const float c0 = (float)(-1.0 + 1.0 / (double)W), c1 = (float)(2.0 / (double)W);
    const long cells = (long)W * W;
    const float co[2] = {coords[bq * 2], coords[bq * 2 + 1]};
    float ar[4], rel_s[2] = {0.f, 0.f};
    long flat_s = 0;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
        int idx[2];
        float rel[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int v = a == 0 ? (ss >> 1) * 2 - 1 : (ss & 1) * 2 - 1;
            const float off = (float)((double)v / (double)W + 1e-6);
            const float c = fminf(hi, fmaxf(lo, co[a] + off));
            idx[a] = nearest_cell(c, W);
            const float qc = __fadd_rn(c0, __fmul_rn(c1, (float)idx[a]));
            rel[a] = (co[a] - qc) * (float)W;
        }
        ar[ss] = fabsf(rel[0] * rel[1]) + 1e-9f;
        if (ss == s) {
            rel_s[0] = rel[0], rel_s[1] = rel[1];
            flat_s = (long)idx[0] * W + idx[1];
        }
    }
    if (quad == 0) {
        const float tot = ((ar[0] + ar[1]) + ar[2]) + ar[3];
        blend_w[row] = ar[3 - s] / tot;
        cell_idx[row] = (int32_t)((long)b * cells + flat_s);
    }
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * quad + k;
        float v = 0.f;
        if (c == 0) v = x_t[((size_t)b * T + i) * cells + flat_s];
        else if (c <= 2) v = rel_s[c - 1];
        else if (c <= 4) v = cell[bq * 2 + (c - 3)] * (float)W;
        else if (c == 5) v = t[(size_t)b * t_ld + i];
        o[k] = v;
    }
    *reinterpret_cast<f32x4*>(tail + row * 32 + 4 * quad) = o;
}

// z[(bq, i)][c] = sum_s p[((bq*S + s)*T + i)][c] * area[bq*S + opposite(s)] / sum_s area   (2D)
//               = (p0*a1 + p1*a0) / (a1 + a0)                                               (1D)
__global__ void k_inr_grid_blend(const float* __restrict__ p, const float* __restrict__ area, int T, int C, int S,
                                 long total, float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int ti = (int)((i / C) % T);
    const long bq = i / ((long)C * T);
    if (S == 2) {
        const float a0 = area[bq * 2], a1 = area[bq * 2 + 1];
        const float p0 = p[((bq * 2 + 0) * T + ti) * C + c], p1 = p[((bq * 2 + 1) * T + ti) * C + c];
        z[i] = (p0 * a1 + p1 * a0) / (a1 + a0);
    } else {
        const float a0 = area[bq * 4], a1 = area[bq * 4 + 1], a2 = area[bq * 4 + 2], a3 = area[bq * 4 + 3];
        const float tot = ((a0 + a1) + a2) + a3;
        float r = 0.f;
        r = r + p[((bq * 4 + 0) * T + ti) * C + c] * (a3 / tot);
        r = r + p[((bq * 4 + 1) * T + ti) * C + c] * (a2 / tot);
        r = r + p[((bq * 4 + 2) * T + ti) * C + c] * (a1 / tot);
        r = r + p[((bq * 4 + 3) * T + ti) * C + c] * (a0 / tot);
        z[i] = r;
    }
}
// the 2-D blend with C % 4 == 0: one thread per (query, time step, 4 channels) - the four weights (one division each) are computed
// once per thread instead of once per element, 16-byte loads and stores; the arithmetic per element is the expression above, bit for bit
__global__ void k_inr_grid_blend4(const float* __restrict__ p, const float* __restrict__ area, int T, int C4, long total4,
                                  float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % C4);
    const int ti = (int)((i / C4) % T);
    const long bq = i / ((long)C4 * T);
    const f32x4 a = *reinterpret_cast<const f32x4*>(area + bq * 4);
    const float tot = ((a[0] + a[1]) + a[2]) + a[3];
    const float w0 = a[3] / tot, w1 = a[2] / tot, w2 = a[1] / tot, w3 = a[0] / tot;
    const f32x4* pp = reinterpret_cast<const f32x4*>(p);
    const f32x4 p0 = pp[((bq * 4 + 0) * T + ti) * C4 + c], p1 = pp[((bq * 4 + 1) * T + ti) * C4 + c];
    const f32x4 p2 = pp[((bq * 4 + 2) * T + ti) * C4 + c], p3 = pp[((bq * 4 + 3) * T + ti) * C4 + c];
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = 0.f;
        v = v + p0[k] * w0;
        v = v + p1[k] * w1;
        v = v + p2[k] * w2;
        v = v + p3[k] * w3;
        r[k] = v;
    }
    reinterpret_cast<f32x4*>(z)[i] = r;
}

// ------------------------------------------------------------------------------------ C entry points
#define S_ (hipStream_t) stream
#define LAUNCH_CHECK() return (int)hipGetLastError()
static inline unsigned nb(long n, int b) { return (unsigned)((n + b - 1) / b); }

extern "C" int magnet_node_features_f32(const float* u, const float* x, const float* t_last, int n_graphs,
                                        int nodes_per_graph, int T, int d, int ld, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    const long n = (long)n_graphs * nodes_per_graph;
    if (n <= 0) return 0;
    if (T + d + 1 > ld) return -2;
    hipLaunchKernelGGL(k_node_features, dim3(nb(n * ld, 256)), dim3(256), 0, S_, u, x, t_last, n_graphs, n, T, d, ld,
                       out);
    LAUNCH_CHECK();
}

extern "C" int magnet_euler_update_f32(const float* ret, int ld_ret, const float* last, const float* t, int t_ld,
                                       int n_graphs, int nodes_per_graph, int T, int T_out, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    const long total = (long)n_graphs * nodes_per_graph * T_out;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_euler_update, dim3(nb(total, 256)), dim3(256), 0, S_, ret, ld_ret, last, t, t_ld, n_graphs,
                       nodes_per_graph, T, T_out, out);
    LAUNCH_CHECK();
}

extern "C" int magnet_inr_knn_prepare_f32(const float* lr_coords, const float* hr_coords, const int32_t* nn, int k,
                                          int d, int n_graphs, int L, int Nq, int32_t* nn2, float* rel, float* w,
                                          void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    const long nq = (long)n_graphs * Nq;
    if (nq <= 0) return 0;
    if (k < 2 || d > 3) return -2;
    (void)L;
    hipLaunchKernelGGL(k_inr_knn_prepare, dim3(nb(nq * 2, 256)), dim3(256), 0, S_, lr_coords, hr_coords, nn, k, d, nq,
                       nn2, rel, w);
    LAUNCH_CHECK();
}

extern "C" int magnet_inr_knn_blend_f32(const float* base, const float* w, const float* vals, const int32_t* nn2,
                                        const float* t, int t_ld, const float* wv, const float* wt, int n_graphs,
                                        int L, int Nq, int T, int C, int mode, float* z, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    const long total = (long)n_graphs * Nq * T * C;
    if (total <= 0) return 0;
    if (C % 4 == 0)
        hipLaunchKernelGGL(k_inr_knn_blend4, dim3(nb(total / 4, 256)), dim3(256), 0, S_, base, w, vals, nn2, t, t_ld, wv, wt, L,
                           Nq, T, C / 4, mode, total / 4, z);
    else
        hipLaunchKernelGGL(k_inr_knn_blend, dim3(nb(total, 256)), dim3(256), 0, S_, base, w, vals, nn2, t, t_ld, wv, wt, L,
                           Nq, T, C, mode, total, z);
    LAUNCH_CHECK();
}

extern "C" int magnet_inr_grid_prepare_f32(const float* coords, const float* cell, const float* x_t, const float* t,
                                           int t_ld, int n_graphs, int Nq, int T, int W, int d, int32_t* cell_idx,
                                           float* tail, float* area, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (d != 1 && d != 2) return -2;
    const long rows = (long)n_graphs * Nq * (d == 2 ? 4 : 2);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_inr_grid_prepare, dim3(nb(rows * T * 8, 256)), dim3(256), 0, S_, coords, cell, x_t, t, t_ld, n_graphs,
                       Nq, T, W, d, cell_idx, tail, area);
    LAUNCH_CHECK();
}

extern "C" int magnet_inr_grid_prepare_blend_f32(const float* coords, const float* cell, const float* x_t, const float* t, int t_ld,
                                                 int n_graphs, int Nq, int T, int W, int32_t* cell_idx, float* tail, float* blend_w,
                                                 void* stream) {
    (void)hipGetLastError();
    const long rows = (long)n_graphs * Nq * T * 4;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_inr_grid_prepare_blend, dim3(nb(rows * 8, 256)), dim3(256), 0, S_, coords, cell, x_t, t, t_ld, n_graphs, Nq, T,
                       W, cell_idx, tail, blend_w);
    LAUNCH_CHECK();
}

extern "C" int magnet_inr_grid_blend_f32(const float* p, const float* area, int n_rows, int T, int C, int nshift,
                                         float* z, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    const long total = (long)n_rows * T * C;
    if (total <= 0) return 0;
    if (nshift != 2 && nshift != 4) return -2;
    if (nshift == 4 && C % 4 == 0)
        hipLaunchKernelGGL(k_inr_grid_blend4, dim3(nb(total / 4, 256)), dim3(256), 0, S_, p, area, T, C / 4, total / 4, z);
    else
        hipLaunchKernelGGL(k_inr_grid_blend, dim3(nb(total, 256)), dim3(256), 0, S_, p, area, T, C, nshift, total, z);
    LAUNCH_CHECK();
}

// ---- EDSR convolutions (models/backbones/edsr.py:32-60) as gathered-tap GEMMs --------------------------------------
// A k x k (or k) "same" convolution over a channels-last feature map [P = B*H*W rows, C] is one Linear over the
// concatenation of the k^d shifted neighbours of every pixel.  tap[t][p] = row of the neighbour of pixel p for tap t
// (dy, dx = t / k - k/2, t % k - k/2), or P - the index of an all-zero row appended to the map - outside the image
// (zero padding).  The fused MLP kernel then gathers its k tiles through these indices: an implicit GEMM, no im2col.
__global__ void k_conv_tap_index(int B, int H, int W, int k, int d, int32_t* __restrict__ tap) {
    const long P = (long)B * H * W;
    const int ntaps = d == 2 ? k * k : k;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * ntaps) return;
    const int t = (int)(i / P);
    const long p = i % P;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const int dy = d == 2 ? t / k - k / 2 : 0, dx = (d == 2 ? t % k : t) - k / 2;
    const int yy = y + dy, xx = x + dx;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    tap[i] = in ? (int32_t)(p + (long)dy * W + dx) : (int32_t)P;
}

extern "C" int magnet_conv_tap_index(int n_images, int H, int W, int k, int d, int32_t* tap, void* stream) {
    (void)hipGetLastError();
    if ((d != 1 && d != 2) || !(k & 1) || (d == 1 && H != 1)) return -2;
    const long total = (long)n_images * H * W * (d == 2 ? k * k : k);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_conv_tap_index, dim3(nb(total, 256)), dim3(256), 0, S_, n_images, H, W, k, d, tap);
    LAUNCH_CHECK();
}

// ---- rollout feedback of MAgNet[CNN]: F.interpolate(mode='(bi)linear', align_corners=False) of the predicted HR field
// back onto the LR grid (models/magnet_cnn.py:456-461, models/magnet_cnn_2d.py:466-472).  Same arithmetic as ATen's
// upsample_(bi)linear kernels: src = (dst + 0.5) * (in / out) - 0.5 clamped at 0, the two neighbours along each axis.
__device__ __forceinline__ void lin_coeff(int dst, float scale, int in_size, int& i0, int& step, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    step = i0 < in_size - 1 ? 1 : 0;
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}
__global__ void k_resize_linear(const float* __restrict__ x, int n_images, int Hi, int Wi, int Ho, int Wo, int d,
                                float* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)n_images * Ho * Wo;
    if (i >= total) return;
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
    const long img = i / ((long)Ho * Wo);
    const float* src = x + img * Hi * Wi;
    int w0, ws;
    float wl0, wl1;
    lin_coeff(xo, (float)Wi / (float)Wo, Wi, w0, ws, wl0, wl1);
    if (d == 1) {
        y[i] = wl0 * src[w0] + wl1 * src[w0 + ws];
        return;
    }
    int h0, hs;
    float hl0, hl1;
    lin_coeff(yo, (float)Hi / (float)Ho, Hi, h0, hs, hl0, hl1);
    const float* r0 = src + (long)h0 * Wi;
    const float* r1 = src + (long)(h0 + hs) * Wi;
    y[i] = hl0 * (wl0 * r0[w0] + wl1 * r0[w0 + ws]) + hl1 * (wl0 * r1[w0] + wl1 * r1[w0 + ws]);
}

extern "C" int magnet_resize_linear_f32(const float* x, int n_images, int Hi, int Wi, int Ho, int Wo, int d, float* y,
                                        void* stream) {
    (void)hipGetLastError();
    if ((d != 1 && d != 2) || (d == 1 && (Hi != 1 || Ho != 1))) return -2;
    const long total = (long)n_images * Ho * Wo;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_resize_linear, dim3(nb(total, 256)), dim3(256), 0, S_, x, n_images, Hi, Wi, Ho, Wo, d, y);
    LAUNCH_CHECK();
}
