// MPNN / MPNN_2d specific kernels (hidden width fixed at 128 by the reference's read-out
// convolution, models/mpnn.py:140-163).
#include "common.cuh"
#include "edge_tile.cuh"

// ---------------------------------------------------------------------------- node input
// node_in[n, :] = [ u[n, 0:tw] | pos[n, 1] / L[0..dpos) | pos[n, 0] / tmax | 0 ... ]   (ld = KIN)
// Restates models/mpnn.py:180-191 (and the (x/Lx, x/Ly) broadcast of models/mpnn_2d.py:179, Q4).
__global__ void k_mpnn_node_input(const float* __restrict__ u, const float* __restrict__ pos, int pos_ld,
                                  const float* __restrict__ Lvec, const float* __restrict__ tmax, int N, int tw,
                                  int dpos, int kin, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * kin) return;
    const int n = (int)(i / kin), c = (int)(i % kin);
    float v = 0.f;
    if (c < tw) v = u[(size_t)n * tw + c];
    else if (c < tw + dpos) v = pos[(size_t)n * pos_ld + 1] / Lvec[c - tw];
    else if (c == tw + dpos) v = pos[(size_t)n * pos_ld] / tmax[0];
    out[i] = v;
}

// ---------------------------------------------------------------------------- range / finiteness status
// The fp16x3 GEMM mode represents every fp32 value with two fp16 pieces, so a value of magnitude >= 65504 silently becomes
// inf (common.cuh).  The models run this check on the arrays that enter the GEMM chains (node features) and on the arrays
// that leave a forward (the predicted window): status |= range_bit where limit <= |x| < inf, |= nonfinite_bit where x is
// inf / NaN - unless the word already carries `unless_bit` (stream order: a free-running rollout feeds a non-finite
// prediction back as the next input, which must not be mistaken for non-finite data).  Sticky (atomicOr), read by the host
// once per rollout.
__global__ void k_status_check(const float* __restrict__ x, long n, float limit, int range_bit, int nonfinite_bit,
                               int unless_bit, int* __restrict__ status) {
    if (unless_bit && (__builtin_nontemporal_load(status) & unless_bit)) nonfinite_bit = 0;
    int flag = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = fabsf(x[i]);
        if (!(a < limit)) flag |= (a <= 3.402823466e+38f) ? range_bit : nonfinite_bit;      // NaN fails both compares
    }
    if (__ballot(flag != 0) == 0) return;
    if (flag) atomicOr(status, flag);
}

// ---------------------------------------------------------------------------- fused edge kernel
// For every CSR edge e = (i <- j):  m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2), then the mean of
// m_e over each target row.  A / Bm are the per-node halves of message_net_1 (exact algebra:
// W1*[x_i, x_j, u_i-u_j, p_i-p_j, v_i] + b1 = (W1a x_i + W1c u_i + W1d p_i + W1e v_i + b1)
// + (W1b x_j - W1c u_j - W1d p_j)), so the [E, 2H+tw+d+1] concat and the [E, H] messages of
// models/mpnn.py:73-79 + the scatter-mean of :46 never touch HBM.
template <int PF, int ABL = 0>
__global__ void __launch_bounds__(256) k_mpnn_edge(const float* __restrict__ A, const float* __restrict__ Bm,
                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                   const int32_t* __restrict__ rowptr, int E, MagnetLayer W2,
                                                   float* __restrict__ agg, float* __restrict__ partials) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    const int ntiles = (E + 31) / 32;
    const int e_raw = tile * 32 + (lane & 31);
    const int e = e_raw < E ? e_raw : E - 1;
    const int ti = tgt[e], sj = src[e];
    const float* pa = A + (size_t)ti * 128 + 4 * half;
    const float* pb = Bm + (size_t)sj * 128 + 4 * half;
    f32x16 in[4], out[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a, b;
            if (ABL & 1) {          // ablation: no gather (synthetic operands, still per-lane distinct)
                a = f32x4{0.001f * lane, 0.002f * kt, 0.003f * g, 0.1f};
                b = f32x4{0.5f, 0.25f * ti, 0.125f, 0.0625f * sj};
            } else {
                a = *reinterpret_cast<const f32x4*>(pa + kt * 32 + 8 * g);
                b = *reinterpret_cast<const f32x4*>(pb + kt * 32 + 8 * g);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) in[kt][4 * g + c] = (ABL & 4) ? (a[c] + b[c]) : act_swish(a[c] + b[c]);
        }
    dense_layer<4, 4, PF>(in, out, W2, lane);
    if (!(ABL & 4)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) apply_act<ACT_SWISH>(out[t]);
    }
    if (ABL & 2) {                  // ablation: no segment-mean epilogue (keep the result live)
        float sacc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc += out[t][i];
        if (sacc == 123.456f) agg[tile] = sacc;
        return;
    }
    segmean_epilogue_dpp<4>(out, tile_meta(tile < ntiles ? tile : 0, E, tgt, rowptr, lane & 31), tile < ntiles ? E : 0, agg, partials, lane);
}

// ---------------------------------------------------------------------------- fused edge kernel, v2
// Same arithmetic as k_mpnn_edge, restructured after profiling (profiles/r1):
//  * message_net_2's packed weights (64 KiB) live in LDS, loaded once per persistent workgroup, so the
//    MFMA stream no longer pulls 64 KiB per 32-edge tile through L1/L2 (7.5 TB/s of L2 reads before);
//  * the segment mean is a segmented inclusive scan across the 32 edge lanes done with DPP
//    (row_shr 1/2/4/8 inside 16-lane rows + row_bcast:15 across them) - no LDS slab, no barrier -
//    and only the last lane of every run of equal targets stores;
//  * persistent grid: NW waves per workgroup, 2 workgroups per CU, tiles strided over the grid.

template <int NW, int ABL = 0>
__global__ void __launch_bounds__(NW * 64, (NW == 12 ? 3 : 2)) k_mpnn_edge2(const float* __restrict__ A, const float* __restrict__ Bm,
                                                        const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                        const int32_t* __restrict__ rowptr, int E, MagnetLayer W2,
                                                        float* __restrict__ agg, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) f32x4 wl[16 * 4 * 64];          // 64 KiB: W2 in fragment order
    for (int i = threadIdx.x; i < 16 * 4 * 64; i += NW * 64) wl[i] = W2.w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int ntiles = (E + 31) / 32;
    for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += gridDim.x * NW) {
        const int e0 = tile * 32;
        const int e_raw = e0 + j;
        const bool e_ok = e_raw < E;
        const int e = e_ok ? e_raw : E - 1;
        const int ti = tgt[e], sj = src[e];
        const int tnext = (e_raw + 1 < E && j < 31) ? tgt[e_raw + 1] : -2;
        const float* pa = A + (size_t)ti * 128 + 4 * half;
        const float* pb = Bm + (size_t)sj * 128 + 4 * half;
        f32x16 in[4], out[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a, b;
                if (ABL & 1) {
                    a = f32x4{0.001f * lane, 0.002f * kt, 0.003f * g, 0.1f};
                    b = f32x4{0.5f, 0.25f * ti, 0.125f, 0.0625f * sj};
                } else {
                    a = *reinterpret_cast<const f32x4*>(pa + kt * 32 + 8 * g);
                    b = *reinterpret_cast<const f32x4*>(pb + kt * 32 + 8 * g);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) in[kt][4 * g + c] = (ABL & 4) ? (a[c] + b[c]) : act_swish(a[c] + b[c]);
            }
        init_bias<4>(out, W2.b, half);
        {   // 16 (kt, nt) blocks; fragments for block b+1 are read from LDS while block b's MFMAs issue.
            // The sched_barriers stop hipcc from hoisting all 64 ds_read_b128 to the top (256 VGPRs + spills).
            f32x4 w[2][4];
            const f32x4* wp = wl + lane;
            w[0][0] = wp[0]; w[0][1] = wp[64]; w[0][2] = wp[128]; w[0][3] = wp[192];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                if (b + 1 < 16) {
                    const f32x4* wn = wl + (b + 1) * 256 + lane;
                    w[(b + 1) & 1][0] = wn[0]; w[(b + 1) & 1][1] = wn[64];
                    w[(b + 1) & 1][2] = wn[128]; w[(b + 1) & 1][3] = wn[192];
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_block(w[b & 1], in[b >> 2], out[b & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ABL & 2) {
            float sacc = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc += out[tt][i];
            if (sacc == 123.456f) agg[tile] = sacc;
            continue;
        }
        // ---- segmented scan over the 32 edge lanes (targets are non-decreasing along j)
        const int t = e_ok ? ti : (-16 - j);                                   // padding lanes: singleton runs
        const int jr = j & 15;
        // the DPP moves must execute with every lane enabled: evaluate them before (not inside) the
        // lane predicates - a short-circuit '&&' lets hipcc sink them under an exec mask, and a DPP
        // read from a disabled lane returns 0
        const int t1 = dpp_i<DPP_SHR1, 0xF>(t), t2 = dpp_i<DPP_SHR2, 0xF>(t), t4 = dpp_i<DPP_SHR4, 0xF>(t);
        const int t8 = dpp_i<DPP_SHR8, 0xF>(t), tb = dpp_i<DPP_BCAST15, 0xA>(t);
        const float m1 = ((jr >= 1) & (t1 == t)) ? 1.f : 0.f;
        const float m2 = ((jr >= 2) & (t2 == t)) ? 1.f : 0.f;
        const float m4 = ((jr >= 4) & (t4 == t)) ? 1.f : 0.f;
        const float m8 = ((jr >= 8) & (t8 == t)) ? 1.f : 0.f;
        const float mb = ((j >= 16) & (tb == t)) ? 1.f : 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float x = act_swish(out[nt][i]);
                x += m1 * dpp_f<DPP_SHR1, 0xF>(x);
                x += m2 * dpp_f<DPP_SHR2, 0xF>(x);
                x += m4 * dpp_f<DPP_SHR4, 0xF>(x);
                x += m8 * dpp_f<DPP_SHR8, 0xF>(x);
                x += mb * dpp_f<DPP_BCAST15, 0xA>(x);
                out[nt][i] = x;
            }
        if (e_ok && tnext != ti) {                                             // last edge of a run: holds the run's sum
            const int s = rowptr[ti], en = rowptr[ti + 1];
            const bool complete = (s >= e0) && (en <= e0 + 32);
            const float dn = complete ? (float)(en - s) : 1.f;
            float* dst = complete ? agg + (size_t)ti * 128 : partials + ((size_t)tile * 2 + (s < e0 ? 0 : 1)) * 128;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {out[nt][4 * g], out[nt][4 * g + 1], out[nt][4 * g + 2], out[nt][4 * g + 3]};
                    if (complete) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = v[c] / dn;
                    }
                    *reinterpret_cast<f32x4*>(dst + nt * 32 + 8 * g + 4 * half) = v;
                }
        }
    }
}

// ---------------------------------------------------------------------------- per-graph InstanceNorm
// y = (x - mean_g) / sqrt(var_g + eps), biased variance per graph and channel, one pass over x
// (PyG InstanceNorm defaults; models/mpnn.py:63,70).  Graphs are contiguous row blocks of
// `rows_per_graph` rows.  Deterministic: chunk partials are combined in a fixed order.
#define NORM_CHUNK 64
#define NORM_SUB 4                                                  // 64-row chunks per partial
// One pass over x for the InstanceNorm statistics: a thread holds its column of a 64-row chunk in registers (64
// independent loads in flight) and emits the chunk's column sum and its sum of squares CENTRED ON THE CHUNK'S OWN MEAN;
// k_colstat_final merges the chunks with the pairwise-variance identity  M2 = sum_c [ M2_c + n_c (mean_c - mean)^2 ]
// - as free of cancellation as the two-pass algorithm, at half its HBM traffic and three launches instead of five.
__global__ void __launch_bounds__(128) k_colstat_partial(const float* __restrict__ x, int rows_per_graph, int H,
                                                         float* __restrict__ part_sum, float* __restrict__ part_m2) {
    // one workgroup = NORM_SUB consecutive 64-row chunks, merged on the fly with the same pairwise identity: four times
    // fewer partials for k_colstat_final (which runs on n_graphs x H/32 workgroups only)
    const int g = blockIdx.y, sup = blockIdx.x, nsup = gridDim.x;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float n_a = 0.f, s_a = 0.f, m2_a = 0.f;
        for (int sub = 0; sub < NORM_SUB; ++sub) {
            const int r0 = (sup * NORM_SUB + sub) * NORM_CHUNK, cnt = min(rows_per_graph - r0, NORM_CHUNK);
            if (cnt <= 0) break;
            const float* col = x + ((size_t)g * rows_per_graph + r0) * H + c;
            float v[NORM_CHUNK];
#pragma unroll
            for (int r = 0; r < NORM_CHUNK; ++r) v[r] = r < cnt ? col[(size_t)r * H] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < NORM_CHUNK; ++r) s += v[r];
            const float n_b = (float)cnt, m = s / n_b;
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < NORM_CHUNK; ++r) {
                const float d = r < cnt ? v[r] - m : 0.f;
                m2 += d * d;
            }
            if (sub == 0) {
                n_a = n_b, s_a = s, m2_a = m2;
            } else {
                const float delta = m - s_a / n_a, n = n_a + n_b;
                m2_a += m2 + delta * delta * (n_a * n_b / n);
                s_a += s;
                n_a = n;
            }
        }
        part_sum[((size_t)g * nsup + sup) * H + c] = s_a;
        part_m2[((size_t)g * nsup + sup) * H + c] = m2_a;
    }
}
// One workgroup per (graph, 32-channel window): 32 groups of 32 lanes each sum a strided subset of the chunk
// partials (128-byte coalesced reads), then group 0 combines the groups in a fixed order (deterministic);
// mean first, then the merged centred second moment (biased variance).
__global__ void __launch_bounds__(1024) k_colstat_final(const float* __restrict__ part_sum, const float* __restrict__ part_m2,
                                                        int nchunks, int rows_per_graph, int H, float* __restrict__ mean,
                                                        float* __restrict__ var, int chunk_rows = NORM_CHUNK * NORM_SUB) {
    __shared__ float sm[32][33];
    __shared__ float mu[32];
    const int g = blockIdx.x, cl = threadIdx.x & 31, c = blockIdx.y * 32 + cl, grp = threadIdx.x >> 5;
    const float n_rows = (float)rows_per_graph;
    float acc = 0.f;
    if (c < H)
        for (int k = grp; k < nchunks; k += 32) acc += part_sum[((size_t)g * nchunks + k) * H + c];
    sm[grp][cl] = acc;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        mu[cl] = tot / n_rows;
        if (c < H) mean[(size_t)g * H + c] = mu[cl];
    }
    __syncthreads();
    const float m = mu[cl];
    acc = 0.f;
    if (c < H)
        for (int k = grp; k < nchunks; k += 32) {
            const float nc = (float)min(chunk_rows, rows_per_graph - k * chunk_rows);
            const float d = part_sum[((size_t)g * nchunks + k) * H + c] / nc - m;
            acc += part_m2[((size_t)g * nchunks + k) * H + c] + nc * d * d;
        }
    __syncthreads();
    sm[grp][cl] = acc;
    __syncthreads();
    if (grp == 0 && c < H) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        var[(size_t)g * H + c] = tot / n_rows;
    }
}
// Small graphs (rows_per_graph <= NORM_SMALL_ROWS, the launch-bound 1-D configs): statistics AND normalisation in ONE launch.
// One workgroup per (graph, 32-channel window), 32 row groups x 32 channels; two passes over the graph's rows (the second
// one hits L1/L2): mean, then the centred second moment - fixed summation order, deterministic.
#define NORM_SMALL_ROWS 2048
__global__ void __launch_bounds__(1024) k_instance_norm_small(const float* __restrict__ x, int rows_per_graph, int H, float eps,
                                                              float* __restrict__ y, float* __restrict__ mean,
                                                              float* __restrict__ var) {
    __shared__ float sm[32][33];
    __shared__ float stat[2][32];
    const int g = blockIdx.x, cl = threadIdx.x & 31, c = blockIdx.y * 32 + cl, grp = threadIdx.x >> 5;
    const float n_rows = (float)rows_per_graph;
    const float* xg = x + (size_t)g * rows_per_graph * H;
    float acc = 0.f;
    if (c < H)
        for (int r = grp; r < rows_per_graph; r += 32) acc += xg[(size_t)r * H + c];
    sm[grp][cl] = acc;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        stat[0][cl] = tot / n_rows;
    }
    __syncthreads();
    const float m = stat[0][cl];
    acc = 0.f;
    if (c < H)
        for (int r = grp; r < rows_per_graph; r += 32) {
            const float d = xg[(size_t)r * H + c] - m;
            acc += d * d;
        }
    __syncthreads();
    sm[grp][cl] = acc;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        stat[1][cl] = tot / n_rows;
        if (c < H) {
            mean[(size_t)g * H + c] = m;
            var[(size_t)g * H + c] = stat[1][cl];
        }
    }
    __syncthreads();
    if (c < H) {
        const float v = stat[1][cl];
        float* yg = y + (size_t)g * rows_per_graph * H;
        for (int r = grp; r < rows_per_graph; r += 32) yg[(size_t)r * H + c] = (xg[(size_t)r * H + c] - m) / sqrtf(v + eps);
    }
}
// Graphs of UNEQUAL size in one batch (PyG's InstanceNorm(x, batch) takes any batch vector, models/mpnn.py:63,70; every caller of
// the reference builds equal graphs, so this is the generality path, not the fast one): the one-launch kernel above with the
// graph's rows taken from gptr[g] .. gptr[g + 1].  An empty graph writes nothing (its mean / var rows are set to 0 / 0).
__global__ void __launch_bounds__(1024) k_instance_norm_ragged(const float* __restrict__ x, const int32_t* __restrict__ gptr, int H,
                                                               float eps, float* __restrict__ y, float* __restrict__ mean,
                                                               float* __restrict__ var) {
    __shared__ float sm[32][33];
    __shared__ float stat[2][32];
    const int g = blockIdx.x, cl = threadIdx.x & 31, c = blockIdx.y * 32 + cl, grp = threadIdx.x >> 5;
    const int r0 = gptr[g], nr = gptr[g + 1] - r0;
    if (nr <= 0) {
        if (grp == 0 && c < H) mean[(size_t)g * H + c] = 0.f, var[(size_t)g * H + c] = 0.f;
        return;
    }
    const float n_rows = (float)nr;
    const float* xg = x + (size_t)r0 * H;
    float acc = 0.f;
    if (c < H)
        for (int r = grp; r < nr; r += 32) acc += xg[(size_t)r * H + c];
    sm[grp][cl] = acc;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        stat[0][cl] = tot / n_rows;
    }
    __syncthreads();
    const float m = stat[0][cl];
    acc = 0.f;
    if (c < H)
        for (int r = grp; r < nr; r += 32) {
            const float d = xg[(size_t)r * H + c] - m;
            acc += d * d;
        }
    __syncthreads();
    sm[grp][cl] = acc;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        stat[1][cl] = tot / n_rows;
        if (c < H) {
            mean[(size_t)g * H + c] = m;
            var[(size_t)g * H + c] = stat[1][cl];
        }
    }
    __syncthreads();
    if (c < H) {
        const float v = stat[1][cl];
        float* yg = y + (size_t)r0 * H;
        for (int r = grp; r < nr; r += 32) yg[(size_t)r * H + c] = (xg[(size_t)r * H + c] - m) / sqrtf(v + eps);
    }
}
__global__ void k_norm_apply(const float* __restrict__ x, const float* __restrict__ mean,
                             const float* __restrict__ var, int rows_per_graph, int H, float eps, long total,
                             float* __restrict__ y) {
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= total) return;
    const int c = (int)(i4 % H);
    const int g = (int)((i4 / H) / rows_per_graph);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i4);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + (size_t)g * H + c);
    const f32x4 vv = *reinterpret_cast<const f32x4*>(var + (size_t)g * H + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (xv[k] - mv[k]) / sqrtf(vv[k] + eps);
    *reinterpret_cast<f32x4*>(y + i4) = o;
}

// ------------------------------------------------------------------------------------ C entry points
#define LAUNCH_CHECK() return (int)hipGetLastError()

// Development builds (-DMAGNET_DEV_VARIANTS, `python -m magnet_amd.build --dev`) keep the earlier kernel generations and
// their ablations (which produce INVALID results) selectable at run time for A/B measurements through
// magnet_debug_set_variant (include/magnet_hip_dev.h).  The product library contains none of it: no knob, no symbol.
#ifdef MAGNET_DEV_VARIANTS
extern int g_edge6_ablation;
extern int g_edge9_ablation;
extern int g_edge_ws_ablation;
static int g_edge_variant = 102;   // persistent v2 kernel, 8 waves / workgroup
// 102 = k_mpnn_edge2<8> (default), 0 = first-generation k_mpnn_edge, 1x/11x = ablations (results invalid)
extern "C" int magnet_debug_set_variant(int which, int value) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (which == 0) g_edge_variant = value;
    if (which == 1) g_edge6_ablation = value;
    if (which == 2) g_edge9_ablation = value;
    if (which == 3) g_edge_ws_ablation = value;    // k_mpnn_edge_ws role ablations (mpnn_edge_ws.hip)      // k_mpnn_edge32t ablation flags (mpnn_edge32t.hip)
    return 0;
}
#endif

extern "C" int magnet_mpnn_node_input_f32(const float* u, const float* pos, int pos_ld, const float* Lvec,
                                          const float* tmax, int n, int tw, int dpos, int kin, float* out,
                                          void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n <= 0) return 0;
    if (tw + dpos + 1 > kin) return -2;
    const long total = (long)n * kin;
    hipLaunchKernelGGL(k_mpnn_node_input, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u,
                       pos, pos_ld, Lvec, tmax, n, tw, dpos, kin, out);
    LAUNCH_CHECK();
}

extern "C" int magnet_status_check_f32(const float* x, long n, float limit, int range_bit, int nonfinite_bit, int unless_bit,
                                       int* status, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n <= 0) return 0;
    const long blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_status_check, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, x,
                       n, limit, range_bit, nonfinite_bit, unless_bit, status);
    LAUNCH_CHECK();
}

extern "C" int magnet_segmean_fixup_f32(const int32_t* rowptr, int n_rows, int d, const float* partials, float* agg, void* stream);

extern "C" int magnet_mpnn_edge_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                    const int32_t* rowptr, int n_edges, int n_rows, const void* w2_packed, const float* b2,
                                    float* agg, float* partials, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_edges <= 0) return magnet_segmean_fixup_f32(rowptr, n_rows, 128, partials, agg, stream);   // all rows empty: zeros
    MagnetLayer L{reinterpret_cast<const f32x4*>(w2_packed), b2};
    const int tiles = (n_edges + 31) / 32;
    const dim3 g((tiles + 3) / 4), b(256);
    hipStream_t s = (hipStream_t)stream;
#ifdef MAGNET_DEV_VARIANTS
    switch (g_edge_variant) {
        case 1: hipLaunchKernelGGL(k_mpnn_edge<1>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 2: hipLaunchKernelGGL(k_mpnn_edge<2>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 110: case 111: case 112: case 113: {
            static int ncu2 = 0;
            if (!ncu2) { hipDeviceProp_t prop; int dev; hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev); ncu2 = prop.multiProcessorCount; }
            const int nw = 8; const dim3 gg(min((tiles + nw - 1) / nw, 2 * ncu2)), bb(nw * 64);
            if (g_edge_variant == 110) hipLaunchKernelGGL((k_mpnn_edge2<8, 1>), gg, bb, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials);
            else if (g_edge_variant == 111) hipLaunchKernelGGL((k_mpnn_edge2<8, 2>), gg, bb, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials);
            else if (g_edge_variant == 112) hipLaunchKernelGGL((k_mpnn_edge2<8, 3>), gg, bb, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials);
            else hipLaunchKernelGGL((k_mpnn_edge2<8, 7>), gg, bb, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials);
            break;
        }
        case 100: case 101: case 102: case 103: {
            // persistent v2: 2 workgroups per CU (64 KiB LDS each)
            hipDeviceProp_t prop;
            static int ncu = 0;
            if (!ncu) { int dev; hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev); ncu = prop.multiProcessorCount; }
            if (g_edge_variant == 103) { const int nw = 12; hipLaunchKernelGGL(k_mpnn_edge2<12>, dim3(min((tiles + nw - 1) / nw, ncu)), dim3(nw * 64), 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); }
            else if (g_edge_variant == 100) { const int nw = 4; hipLaunchKernelGGL(k_mpnn_edge2<4>, dim3(min((tiles + nw - 1) / nw, 2 * ncu)), dim3(nw * 64), 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); }
            else if (g_edge_variant == 101) { const int nw = 6; hipLaunchKernelGGL(k_mpnn_edge2<6>, dim3(min((tiles + nw - 1) / nw, 2 * ncu)), dim3(nw * 64), 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); }
            else { const int nw = 8; hipLaunchKernelGGL(k_mpnn_edge2<8>, dim3(min((tiles + nw - 1) / nw, 2 * ncu)), dim3(nw * 64), 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); }
            break;
        }
        case 3: hipLaunchKernelGGL(k_mpnn_edge<3>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 23: hipLaunchKernelGGL((k_mpnn_edge<3, 7>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 22: hipLaunchKernelGGL((k_mpnn_edge<2, 7>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 21: hipLaunchKernelGGL((k_mpnn_edge<1, 7>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 10: hipLaunchKernelGGL((k_mpnn_edge<0, 1>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 11: hipLaunchKernelGGL((k_mpnn_edge<0, 2>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 12: hipLaunchKernelGGL((k_mpnn_edge<0, 3>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 13: hipLaunchKernelGGL((k_mpnn_edge<0, 7>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        case 14: hipLaunchKernelGGL((k_mpnn_edge<0, 4>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
        default: hipLaunchKernelGGL(k_mpnn_edge<0>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials); break;
    }
#else
    {   // persistent k_mpnn_edge2: 2 workgroups of 8 waves per CU (64 KiB LDS each)
        (void)g; (void)b;
        static int ncu = 0;
        if (!ncu) { hipDeviceProp_t prop; int dev; hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev); ncu = prop.multiProcessorCount; }
        const int nw = 8;
        hipLaunchKernelGGL(k_mpnn_edge2<8>, dim3(min((tiles + nw - 1) / nw, 2 * ncu)), dim3(nw * 64), 0, s, A, Bm, src, tgt, rowptr, n_edges, L, agg, partials);
    }
#endif
    const int rc = (int)hipGetLastError();
    return rc ? rc : magnet_segmean_fixup_f32(rowptr, n_rows, 128, partials, agg, stream);
}


// statistics only: mean at workspace + n_graphs * nchunks * H, biased variance right behind it ([n_graphs][H] each); the apply
// step is fused into the consumer (magnet_mpnn_node_halves_f32).  Same workspace layout as magnet_instance_norm_f32.
extern "C" int magnet_instance_norm_stats_f32(const float* x, int n_graphs, int rows_per_graph, int H, float* workspace,
                                              void* stream) {
    (void)hipGetLastError();
    if (n_graphs <= 0 || rows_per_graph <= 0) return 0;
    if (H % 4) return -2;
    const int nchunks = (rows_per_graph + NORM_CHUNK - 1) / NORM_CHUNK;
    float* part = workspace;
    float* mean = part + (size_t)n_graphs * nchunks * H;
    float* var = mean + (size_t)n_graphs * H;
    float* part_m2 = var + (size_t)n_graphs * H;
    hipStream_t s = (hipStream_t)stream;
    const int nsup = (nchunks + NORM_SUB - 1) / NORM_SUB;
    hipLaunchKernelGGL(k_colstat_partial, dim3(nsup, n_graphs), dim3(128), 0, s, x, rows_per_graph, H, part, part_m2);
    hipLaunchKernelGGL(k_colstat_final, dim3(n_graphs, (H + 31) / 32), dim3(1024), 0, s, part, part_m2, nsup,
                       rows_per_graph, H, mean, var);
    LAUNCH_CHECK();
}
// mean / biased variance [n_graphs][H] from per-chunk partials (column sums and squares centred on the chunk's own column means,
// chunk_rows rows each, chunks of a graph consecutive: part[(g * nchunks + k) * H + c]) - the merge step of the InstanceNorm statistics
// for a producer that emits the partials itself (magnet_mpnn_node_update_f32)
extern "C" int magnet_colstat_final_f32(const float* part_sum, const float* part_m2, int n_graphs, int rows_per_graph, int chunk_rows,
                                        int H, float* mean, float* var, void* stream) {
    (void)hipGetLastError();
    if (n_graphs <= 0 || rows_per_graph <= 0) return 0;
    if (chunk_rows <= 0 || rows_per_graph % chunk_rows) return -2;
    hipLaunchKernelGGL(k_colstat_final, dim3(n_graphs, (H + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part_sum, part_m2,
                       rows_per_graph / chunk_rows, rows_per_graph, H, mean, var, chunk_rows);
    LAUNCH_CHECK();
}
extern "C" int magnet_instance_norm_ragged_f32(const float* x, const int32_t* gptr, int n_graphs, int H, float eps, float* y,
                                               float* mean, float* var, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_graphs <= 0) return 0;
    hipLaunchKernelGGL(k_instance_norm_ragged, dim3(n_graphs, (H + 31) / 32), dim3(1024), 0, (hipStream_t)stream, x, gptr, H, eps, y,
                       mean, var);
    LAUNCH_CHECK();
}

// workspace: (2 * n_graphs * nchunks + 2 * n_graphs) * H floats, nchunks = ceil(rows_per_graph / 64)
extern "C" int magnet_instance_norm_f32(const float* x, int n_graphs, int rows_per_graph, int H, float eps, float* y,
                                        float* workspace, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_graphs <= 0 || rows_per_graph <= 0) return 0;
    if (H % 4) return -2;
    const int nchunks = (rows_per_graph + NORM_CHUNK - 1) / NORM_CHUNK;
    float* part = workspace;
    float* mean = part + (size_t)n_graphs * nchunks * H;
    float* var = mean + (size_t)n_graphs * H;
    hipStream_t s = (hipStream_t)stream;
    if (rows_per_graph <= NORM_SMALL_ROWS) {
        hipLaunchKernelGGL(k_instance_norm_small, dim3(n_graphs, (H + 31) / 32), dim3(1024), 0, s, x, rows_per_graph, H, eps, y, mean,
                           var);
        LAUNCH_CHECK();
    }
    float* part_m2 = var + (size_t)n_graphs * H;
    const int nsup = (nchunks + NORM_SUB - 1) / NORM_SUB;     // the workspace keeps its documented per-64-row layout; fewer slots are used
    hipLaunchKernelGGL(k_colstat_partial, dim3(nsup, n_graphs), dim3(128), 0, s, x, rows_per_graph, H, part, part_m2);
    hipLaunchKernelGGL(k_colstat_final, dim3(n_graphs, (H + 31) / 32), dim3(1024), 0, s, part, part_m2, nsup,
                       rows_per_graph, H, mean, var);
    const long total = (long)n_graphs * rows_per_graph * H;
    hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, x, mean, var,
                       rows_per_graph, H, eps, total, y);
    LAUNCH_CHECK();
}
