// Neighbour search, CSR construction and the standalone CSR segment-mean ("scatter mean").
//
// Search semantics restate torch_cluster's CUDA kernels (not vendored in the reference; called at
// models/mpnn.py:245, models/magnet_gnn.py:247,293): per query, scan the query's graph segment in
// ascending index, keep candidates with dist^2 < r^2 (strict), stop after `cap` hits; knn keeps an
// insertion-sorted top-k with strict '>' (lower index wins ties).  dist^2 is accumulated
// coordinate by coordinate with separately rounded multiply and add (no FMA contraction), the same
// arithmetic the oracle uses.
#include "common.cuh"

__device__ __forceinline__ float dist2(const float* __restrict__ x, int a, const float* __restrict__ yq, int dim) {
    float d = 0.f;
    for (int k = 0; k < dim; ++k) {
        const float t = __fsub_rn(yq[k], x[(size_t)a * dim + k]);
        d = __fadd_rn(d, __fmul_rn(t, t));
    }
    return d;
}

__device__ __forceinline__ int graph_of(const int32_t* __restrict__ gptr, int ng, int q) {
    int g = 0;
    while (g + 1 < ng && q >= gptr[g + 1]) ++g;
    return g;
}

// one wave per query; 64 candidates per step in ascending index order
__global__ void __launch_bounds__(256) k_radius(const float* __restrict__ x, int dim, int N,
                                                const int32_t* __restrict__ gptr, int ng, float r2, int cap,
                                                int32_t* __restrict__ nbr, int32_t* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= N) return;
    const int g = graph_of(gptr, ng, q);
    const int s = gptr[g], e = gptr[g + 1];
    float yq[3];
    for (int k = 0; k < dim; ++k) yq[k] = x[(size_t)q * dim + k];
    int count = 0;
    for (int base = s; base < e && count < cap; base += 64) {
        const int c = base + lane;
        const bool hit = (c < e) && (dist2(x, c < e ? c : s, yq, dim) < r2);
        const unsigned long long mask = __ballot(hit);
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < cap) nbr[(size_t)q * cap + pos] = c;
        count += __popcll(mask);
    }
    if (lane == 0) cnt[q] = count < cap ? count : cap;
}

// ---- accelerated search with identical results: per 64-node chunk bounding boxes let a query skip
// chunks that cannot contain a hit.  The lower bound uses the same monotone fp32 operation sequence
// as dist2 (|q - x| >= gap, rounding is monotone), so lb >= r2 proves no point of the chunk passes
// d2 < r2 - the visited candidates, their order and the early exit are unchanged.
__global__ void __launch_bounds__(256) k_chunk_bbox(const float* __restrict__ x, int dim,
                                                    const int32_t* __restrict__ gptr, const int32_t* __restrict__ cptr,
                                                    int ng, int nchunks, float* __restrict__ bbox) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    int g = 0;
    while (g + 1 < ng && c >= cptr[g + 1]) ++g;
    const int s = gptr[g] + (c - cptr[g]) * 64, e = min(gptr[g + 1], s + 64);
    const int i = s + lane < e ? s + lane : e - 1;
    for (int k = 0; k < dim; ++k) {
        float lo = x[(size_t)i * dim + k], hi = lo;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if (lane == 0) { bbox[((size_t)c * 3 + k) * 2] = lo; bbox[((size_t)c * 3 + k) * 2 + 1] = hi; }
    }
}

// One box per 64 CHUNKS (4 096 nodes; super-chunk s = chunks 64 s .. 64 s + 63 of the global chunk numbering - it may straddle two
// graphs, a box over more points is still a valid lower bound) behind the chunk boxes: bbox[(nchunks + s) * 6 ..].
__global__ void __launch_bounds__(256) k_super_bbox(int nchunks, int nsuper, float* __restrict__ bbox) {
    const int lane = threadIdx.x & 63;
    const int sc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sc >= nsuper) return;
    const int c = sc * 64 + lane;
    for (int k = 0; k < 3; ++k) {
        float lo = c < nchunks ? bbox[((size_t)c * 3 + k) * 2] : INFINITY, hi = c < nchunks ? bbox[((size_t)c * 3 + k) * 2 + 1] : -INFINITY;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if (lane == 0) { bbox[((size_t)(nchunks + sc) * 3 + k) * 2] = lo; bbox[((size_t)(nchunks + sc) * 3 + k) * 2 + 1] = hi; }
    }
}

// Two levels (round 5): a query first tests the boxes of the super-chunks its graph touches (a 65 536-node graph: 16 of them, one
// lane each) and only inside a super-chunk that passes the 64 chunk boxes - before, sixteen rounds of 64 chunk tests per query were
// most of the kernel (0.33 ms per 4 x 256^2 build).  Chunks are still visited in ascending order with the same early exit: the
// same neighbours in the same order.
__global__ void __launch_bounds__(256) k_radius_bbox(const float* __restrict__ x, int dim, int N,
                                                     const int32_t* __restrict__ gptr, const int32_t* __restrict__ cptr,
                                                     int ng, int nchunks, const float* __restrict__ bbox, float r2, int cap,
                                                     int32_t* __restrict__ nbr, int32_t* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= N) return;
    const int g = graph_of(gptr, ng, q);
    const int s = gptr[g], e = gptr[g + 1];
    const int c0 = cptr[g], c1 = cptr[g + 1];
    float yq[3];
    for (int k = 0; k < dim; ++k) yq[k] = x[(size_t)q * dim + k];
    auto below = [&](int box) {                         // lower bound of the squared distance to box `box` < r2 ?
        float lb = 0.f;
        for (int k = 0; k < dim; ++k) {
            const float lo = bbox[((size_t)box * 3 + k) * 2], hi = bbox[((size_t)box * 3 + k) * 2 + 1];
            const float gap = fmaxf(fmaxf(__fsub_rn(lo, yq[k]), __fsub_rn(yq[k], hi)), 0.f);
            lb = __fadd_rn(lb, __fmul_rn(gap, gap));
        }
        return lb < r2;
    };
    int count = 0;
    const int S0 = c0 >> 6, S1 = (c1 + 63) >> 6;
    for (int sb = S0; sb < S1 && count < cap; sb += 64) {
        const int sc = sb + lane;
        unsigned long long sm = __ballot(sc < S1 && below(nchunks + sc));
        while (sm && count < cap) {
            const int sbit = __ffsll((long long)sm) - 1;
            sm &= sm - 1;
            const int cb = (sb + sbit) << 6;            // chunks cb .. cb + 63, clipped to this graph's
            const int c = cb + lane;
            unsigned long long cm = __ballot(c >= c0 && c < c1 && below(c));
            while (cm && count < cap) {
                const int bit = __ffsll((long long)cm) - 1;
                cm &= cm - 1;
                const int base = s + (cb + bit - c0) * 64;
                const int i = base + lane;
                const bool hit = (i < e) && (dist2(x, i < e ? i : s, yq, dim) < r2);
                const unsigned long long mask = __ballot(hit);
                const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && pos < cap) nbr[(size_t)q * cap + pos] = i;
                count += __popcll(mask);
            }
        }
    }
    if (lane == 0) cnt[q] = count < cap ? count : cap;
}

// kept[q] = number of neighbours kept for query q (self dropped when !loop)
__global__ void k_kept(const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int N, int cap, int loop,
                       int32_t* __restrict__ kept) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > N) return;
    if (q == N) { kept[N] = 0; return; }
    int c = cnt[q];
    if (!loop)
        for (int k = 0; k < cnt[q]; ++k) c -= (nbr[(size_t)q * cap + k] == q);
    kept[q] = c;
}

// in-place exclusive scan of n ints by one 1024-thread block: 4096-int chunks, 16 coalesced bytes per thread, the next
// chunk's load in flight while the current one is scanned (wave scan by __shfl_up, 16 wave totals through LDS)
__global__ void __launch_bounds__(1024) k_exclusive_scan(int32_t* __restrict__ v, int n, long array_stride = 0) {
    v += (size_t)blockIdx.x * array_stride;                // one block per array (magnet_launch_exclusive_scans)
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nch = (n + 4095) / 4096;
    // 16-byte accesses only from a 16-byte aligned array (the stride of a multi-array launch need not be a multiple of four ints)
    const bool vec = (reinterpret_cast<uintptr_t>(v) & 15) == 0;
    auto load = [&](int ch, int (&x)[4]) {
        const int b = ch * 4096 + t * 4;
        if (vec && b + 3 < n) {
            const int4 q = *reinterpret_cast<const int4*>(v + b);
            x[0] = q.x, x[1] = q.y, x[2] = q.z, x[3] = q.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = b + c < n ? v[b + c] : 0;
        }
    };
    int x[4], y[4] = {0, 0, 0, 0};
    int carry = 0;
    if (nch > 0) load(0, x);
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) load(ch + 1, y);
        const int s = x[0] + x[1] + x[2] + x[3];
        int inc = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(inc, off);
            if (lane >= off) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int ws = wsum[w];
            before += w < wave ? ws : 0;
            total += ws;
        }
        __syncthreads();
        int run = carry + before + inc - s;
        const int b = ch * 4096 + t * 4;
        if (vec && b + 3 < n) {
            int4 q;
            q.x = run, q.y = run + x[0], q.z = run + x[0] + x[1], q.w = run + x[0] + x[1] + x[2];
            *reinterpret_cast<int4*>(v + b) = q;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (b + c < n) v[b + c] = run;
                run += x[c];
            }
        }
        carry += total;
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = y[c];
    }
}

// query-major edge list: found[qptr[q] + k], query[...] = q   (== torch_cluster / PyG edge order)
// Half a wave per query: 32 lanes read 32 consecutive candidates (coalesced), the kept ones are ranked with a ballot and
// written next to each other; a thread per query read and wrote with a stride of `cap` ints (0.31 ms for 8.6 M pairs).
__global__ void __launch_bounds__(256) k_emit_pairs(const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt,
                                                    const int32_t* __restrict__ qptr, int N, int cap, int loop,
                                                    int32_t* __restrict__ found, int32_t* __restrict__ query) {
    const int lane = threadIdx.x & 63, sub = lane & 31, hi = lane >> 5;
    const int q = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + hi;
    const bool q_ok = q < N;
    const int c = q_ok ? cnt[q] : 0;
    int o = q_ok ? qptr[q] : 0;
    const int cmax = max(__shfl(c, 0), __shfl(c, 32));            // both halves run the same number of rounds
    for (int k0 = 0; k0 < cmax; k0 += 32) {
        const int k = k0 + sub;
        const int f = (k < c) ? nbr[(size_t)q * cap + k] : -1;
        const bool keep = (k < c) && (loop || f != q);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const unsigned mh = (unsigned)(hi ? (m >> 32) : m);       // this query's 32 lanes
        if (keep) {
            const int pos = o + __builtin_popcount(mh & ((1u << sub) - 1u));
            found[pos] = f;
            query[pos] = q;
        }
        o += __builtin_popcount(mh);
    }
}

// ---- generic CSR-by-target of an arbitrary (src, tgt) edge list, stable in edge order
__global__ void k_count_tgt(const int32_t* __restrict__ tgt, int E, int32_t* __restrict__ deg) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&deg[tgt[e]], 1);
}
__global__ void k_fill_tgt(const int32_t* __restrict__ tgt, int E, const int32_t* __restrict__ rowptr,
                           int32_t* __restrict__ cursor, int32_t* __restrict__ eid_unsorted) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int t = tgt[e];
    eid_unsorted[rowptr[t] + atomicAdd(&cursor[t], 1)] = e;
}
// one wave per row: rank-sort the row's edge ids ascending, then emit (perm, src, tgt) in CSR order.  The rank sort is
// O(d^2) per row: rows longer than SORT_BIG_ROW are left to k_sort_big_rows.
#define SORT_BIG_ROW 1024
__global__ void __launch_bounds__(256) k_sort_rows(const int32_t* __restrict__ rowptr, int N,
                                                   const int32_t* __restrict__ eid_unsorted,
                                                   const int32_t* __restrict__ src_in, int32_t* __restrict__ perm,
                                                   int32_t* __restrict__ src_out, int32_t* __restrict__ tgt_out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int s = rowptr[r], d = rowptr[r + 1] - s;
    if (d > SORT_BIG_ROW) return;
    for (int i = lane; i < d; i += 64) {
        const int v = eid_unsorted[s + i];
        int rank = 0;
        for (int k = 0; k < d; ++k) rank += (eid_unsorted[s + k] < v);
        perm[s + rank] = v;
        src_out[s + rank] = src_in[v];
        tgt_out[s + rank] = r;
    }
}
// Long rows (the inverse CSR of a gather index can have rows of 10^4..10^5 entries: every query of an INR decoder that
// reads one coarse node; the rank sort then ran for 16 ms on one wave): one workgroup per row, least-significant-bit-first
// binary radix sort of the edge ids - per bit a stable block-wide partition (count the zeros, then ordered 256-element
// chunks ranked with ballots), ping-ponging between the row's segments of eid_unsorted and perm.  O(d log E).
__global__ void __launch_bounds__(256) k_sort_big_rows(const int32_t* __restrict__ rowptr, int N, int nbits,
                                                       int32_t* __restrict__ eid_unsorted,
                                                       const int32_t* __restrict__ src_in, int32_t* __restrict__ perm,
                                                       int32_t* __restrict__ src_out, int32_t* __restrict__ tgt_out) {
    __shared__ int wz[4];
    __shared__ int ztotal;
    const int r = blockIdx.x;
    const int s = rowptr[r], d = rowptr[r + 1] - s;
    if (d <= SORT_BIG_ROW) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* a = eid_unsorted + s;
    int32_t* b = perm + s;
    for (int bit = 0; bit < nbits; ++bit) {
        int z = 0;
        for (int i = tid; i < d; i += 256) z += !((a[i] >> bit) & 1);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off);
        if (lane == 0) wz[wave] = z;
        __syncthreads();
        if (tid == 0) ztotal = wz[0] + wz[1] + wz[2] + wz[3];
        __syncthreads();
        const int Z = ztotal;
        int zrun = 0, orun = 0;
        for (int c0 = 0; c0 < d; c0 += 256) {
            const int i = c0 + tid, nvalid = min(256, d - c0);
            const bool valid = i < d;
            const int v = valid ? a[i] : 0;
            const bool isz = valid && !((v >> bit) & 1);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(isz);
            const int zpre_w = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            __syncthreads();                                  // wz of the previous chunk has been read by everybody
            if (lane == 0) wz[wave] = __builtin_popcountll(m);
            __syncthreads();
            int zbefore = zpre_w, zchunk = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                zbefore += w < wave ? wz[w] : 0;
                zchunk += wz[w];
            }
            if (valid) {
                if (isz) b[zrun + zbefore] = v;
                else b[Z + orun + (tid - zbefore)] = v;      // valid lanes are 0 .. nvalid-1: ones before me = tid - zeros before me
            }
            zrun += zchunk;
            orun += nvalid - zchunk;
        }
        __syncthreads();
        int32_t* t = a;
        a = b;
        b = t;
    }
    // `a` holds the sorted ids (either buffer); emit in place / across
    for (int i = tid; i < d; i += 256) {
        const int v = a[i];                                   // in place when `a` is the row's perm segment: one thread per index
        perm[s + i] = v;
        src_out[s + i] = src_in[v];
        tgt_out[s + i] = r;
    }
}

__global__ void k_i64_to_i32(const int64_t* __restrict__ a, int n, int32_t* __restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = (int32_t)a[i];
}
__global__ void k_pairs_to_i64(const int32_t* __restrict__ r0, const int32_t* __restrict__ r1, int E,
                               int64_t* __restrict__ ei) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) {
        ei[i] = r0[i];
        ei[(size_t)E + i] = r1[i];
    }
}

// ---- k nearest x for every y, same graph, thread per query, k <= 8
__global__ void k_knn(const float* __restrict__ x, const int32_t* __restrict__ xptr, const float* __restrict__ y,
                      const int32_t* __restrict__ yptr, int ng, int dim, int M, int k, int32_t* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M) return;
    const int g = graph_of(yptr, ng, q);
    float bd[8];
    int bi[8];
    for (int i = 0; i < 8; ++i) { bd[i] = 1e10f; bi[i] = -1; }
    float yq[3];
    for (int c = 0; c < dim; ++c) yq[c] = y[(size_t)q * dim + c];
    for (int n = xptr[g]; n < xptr[g + 1]; ++n) {
        const float d = dist2(x, n, yq, dim);
#pragma unroll
        for (int e1 = 0; e1 < 8; ++e1) {
            if (e1 < k && bd[e1] > d) {
#pragma unroll
                for (int e2 = 7; e2 > 0; --e2)
                    if (e2 > e1 && e2 < k) { bd[e2] = bd[e2 - 1]; bi[e2] = bi[e2 - 1]; }
                bd[e1] = d;
                bi[e1] = n;
                break;
            }
        }
    }
    for (int i = 0; i < k; ++i) out[(size_t)q * k + i] = bi[i];
}

// The same result with one WAVE per query: lane l scans the candidates s + l, s + l + 64, ... of the query's graph into its own sorted
// list (the insertion of k_knn), then the wave takes the k smallest heads in (distance, index) order - exactly the list one thread
// builds in ascending candidate order (strict ">" there keeps the lower index first among equal distances).  At the reference's training
// shapes (8 x 256 queries against 1 024 nodes each) the thread-per-query kernel is 16 workgroups of sequential 1 024-candidate scans:
// 1.2 ms, 6 % of a magnet_gnn training step; this one is a few microseconds.
__global__ void __launch_bounds__(256) k_knn_wave(const float* __restrict__ x, const int32_t* __restrict__ xptr,
                                                  const float* __restrict__ y, const int32_t* __restrict__ yptr, int ng, int dim, int M,
                                                  int k, int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= M) return;
    const int g = graph_of(yptr, ng, q);
    float bd[8];
    int bi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bd[i] = 1e10f; bi[i] = -1; }
    float yq[3];
    for (int c = 0; c < dim; ++c) yq[c] = y[(size_t)q * dim + c];
    const int e = xptr[g + 1];
    for (int n = xptr[g] + lane; n < e; n += 64) {
        const float d = dist2(x, n, yq, dim);
#pragma unroll
        for (int e1 = 0; e1 < 8; ++e1) {
            if (e1 < k && bd[e1] > d) {
#pragma unroll
                for (int e2 = 7; e2 > 0; --e2)
                    if (e2 > e1 && e2 < k) { bd[e2] = bd[e2 - 1]; bi[e2] = bi[e2 - 1]; }
                bd[e1] = d;
                bi[e1] = n;
                break;
            }
        }
    }
    for (int r = 0; r < k; ++r) {
        float md = bd[0];
        int mi = bi[0];
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const float od = __shfl_xor(md, off);
            const int oi = __shfl_xor(mi, off);
            if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
        }
        if (lane == 0) out[(size_t)q * k + r] = mi;
        if (bd[0] == md && bi[0] == mi) {                       // the winner's list moves up (empty heads are all (1e10, -1): any of them)
#pragma unroll
            for (int e2 = 0; e2 < 7; ++e2) { bd[e2] = bd[e2 + 1]; bi[e2] = bi[e2 + 1]; }
            bd[7] = 1e10f;
            bi[7] = -1;
        }
    }
}

// ---- standalone CSR segment mean: out[r] = mean of msg[rowptr[r] .. rowptr[r+1]) (0 when empty).
// HBM-bound: every message row is read exactly once with 16 B per lane, LPR = H/4 lanes per row,
// 64/LPR edges per wave pass, one wave per target row; partial sums meet in a __shfl_xor tree.
// Replaces torch_scatter's atomic scatter-mean behind PyG aggr='mean' (models/mpnn.py:46,
// models/magnet_gnn.py:54).  Algorithmic bytes: E*H*4 + N*H*4 + 4*(N+1).
template <int H>
__global__ void __launch_bounds__(256) k_segment_mean(const float* __restrict__ msg,
                                                      const int32_t* __restrict__ rowptr, int N,
                                                      float* __restrict__ out) {
    constexpr int LPR = H / 4;          // lanes per message row
    constexpr int EPP = 64 / LPR;       // edges per pass
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int s = rowptr[r], e = rowptr[r + 1];
    const int sub = lane / LPR, col = (lane % LPR) * 4;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
    int i = s + sub;
    for (; i + 3 * EPP < e; i += 4 * EPP) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(msg + (size_t)i * H + col);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(msg + (size_t)(i + EPP) * H + col);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(msg + (size_t)(i + 2 * EPP) * H + col);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(msg + (size_t)(i + 3 * EPP) * H + col);
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; i < e; i += EPP) a0 += *reinterpret_cast<const f32x4*>(msg + (size_t)i * H + col);
    f32x4 a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] += __shfl_xor(a[c], off);
    }
    if (sub == 0) {
        const float dn = (float)max(e - s, 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = a[c] / dn;
        *reinterpret_cast<f32x4*>(out + (size_t)r * H + col) = a;
    }
}

// ------------------------------------------------------------------------------------ C entry points
#define S_ (hipStream_t) stream
#define LAUNCH_CHECK() return (int)hipGetLastError()
static inline unsigned nb(long n, int b) { return (unsigned)((n + b - 1) / b); }

extern "C" int magnet_radius_search_f32(const float* x, int dim, int n_nodes, const int32_t* graph_ptr, int n_graphs,
                                        float r2, int cap, int32_t* nbr, int32_t* cnt, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_nodes <= 0) return 0;
    if (dim < 1 || dim > 3 || cap < 1) return -2;
    hipLaunchKernelGGL(k_radius, dim3(nb(n_nodes, 4)), dim3(256), 0, S_, x, dim, n_nodes, graph_ptr, n_graphs, r2, cap,
                       nbr, cnt);
    LAUNCH_CHECK();
}

// Same result as magnet_radius_search_f32, ~10x faster on large graphs.  chunk_ptr[g] = first 64-node
// chunk of graph g (chunks never straddle graphs), bbox_scratch: (n_chunks + ceil(n_chunks / 64)) * 6 floats.
extern "C" int magnet_radius_search_bbox_f32(const float* x, int dim, int n_nodes, const int32_t* graph_ptr,
                                             const int32_t* chunk_ptr, int n_graphs, int n_chunks, float r2, int cap,
                                             float* bbox_scratch, int32_t* nbr, int32_t* cnt, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_nodes <= 0) return 0;
    if (dim < 1 || dim > 3 || cap < 1) return -2;
    hipLaunchKernelGGL(k_chunk_bbox, dim3(nb(n_chunks, 4)), dim3(256), 0, S_, x, dim, graph_ptr, chunk_ptr, n_graphs,
                       n_chunks, bbox_scratch);
    const int n_super = (n_chunks + 63) / 64;
    hipLaunchKernelGGL(k_super_bbox, dim3(nb(n_super, 4)), dim3(256), 0, S_, n_chunks, n_super, bbox_scratch);
    hipLaunchKernelGGL(k_radius_bbox, dim3(nb(n_nodes, 4)), dim3(256), 0, S_, x, dim, n_nodes, graph_ptr, chunk_ptr,
                       n_graphs, n_chunks, bbox_scratch, r2, cap, nbr, cnt);
    LAUNCH_CHECK();
}

// in-place exclusive scans of `narrays` arrays of n ints, `array_stride` ints apart, one 1024-thread block each (C++ linkage: the
// tile plan of mpnn_edge_rows.hip scans its three per-row counts in one launch)
int magnet_launch_exclusive_scans(int32_t* v, int n, int narrays, long array_stride, void* stream) {
    if (n > 0 && narrays > 0) hipLaunchKernelGGL(k_exclusive_scan, dim3(narrays), dim3(1024), 0, S_, v, n, array_stride);
    return (int)hipGetLastError();
}

// qptr[n_nodes + 1] <- exclusive scan of the kept-neighbour counts; qptr[n_nodes] = E
extern "C" int magnet_radius_offsets(const int32_t* nbr, const int32_t* cnt, int n_nodes, int cap, int loop,
                                     int32_t* qptr, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    hipLaunchKernelGGL(k_kept, dim3(nb(n_nodes + 1, 256)), dim3(256), 0, S_, nbr, cnt, n_nodes, cap, loop, qptr);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, S_, qptr, n_nodes + 1);
    LAUNCH_CHECK();
}

extern "C" int magnet_radius_emit_pairs(const int32_t* nbr, const int32_t* cnt, const int32_t* qptr, int n_nodes,
                                        int cap, int loop, int32_t* found, int32_t* query, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_nodes <= 0) return 0;
    hipLaunchKernelGGL(k_emit_pairs, dim3(nb(n_nodes, 8)), dim3(256), 0, S_, nbr, cnt, qptr, n_nodes, cap, loop,
                       found, query);
    LAUNCH_CHECK();
}

// CSR by target, stable in input edge order.  rowptr[n+1], cursor[n] (scratch), eid_scratch[E],
// perm/src_out/tgt_out[E].  rowptr and cursor must be zero on entry.
extern "C" int magnet_csr_by_target(const int32_t* src, const int32_t* tgt, int n_edges, int n_nodes, int32_t* rowptr,
                                    int32_t* cursor, int32_t* eid_scratch, int32_t* perm, int32_t* src_out,
                                    int32_t* tgt_out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_edges > 0) hipLaunchKernelGGL(k_count_tgt, dim3(nb(n_edges, 256)), dim3(256), 0, S_, tgt, n_edges, rowptr);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, S_, rowptr, n_nodes + 1);
    if (n_edges > 0) {
        hipLaunchKernelGGL(k_fill_tgt, dim3(nb(n_edges, 256)), dim3(256), 0, S_, tgt, n_edges, rowptr, cursor,
                           eid_scratch);
        hipLaunchKernelGGL(k_sort_rows, dim3(nb(n_nodes, 4)), dim3(256), 0, S_, rowptr, n_nodes, eid_scratch, src, perm,
                           src_out, tgt_out);
        int nbits = 1;
        while (nbits < 31 && (1 << nbits) < n_edges) ++nbits;
        hipLaunchKernelGGL(k_sort_big_rows, dim3(n_nodes), dim3(256), 0, S_, rowptr, n_nodes, nbits, eid_scratch, src, perm,
                           src_out, tgt_out);
    }
    LAUNCH_CHECK();
}

extern "C" int magnet_i64_to_i32(const int64_t* a, int n, int32_t* b, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_i64_to_i32, dim3(nb(n, 256)), dim3(256), 0, S_, a, n, b);
    LAUNCH_CHECK();
}
extern "C" int magnet_pairs_to_edge_index(const int32_t* row0, const int32_t* row1, int n_edges, int64_t* edge_index,
                                          void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_edges <= 0) return 0;
    hipLaunchKernelGGL(k_pairs_to_i64, dim3(nb(n_edges, 256)), dim3(256), 0, S_, row0, row1, n_edges, edge_index);
    LAUNCH_CHECK();
}

extern "C" int magnet_knn_f32(const float* x, const int32_t* x_ptr, const float* y, const int32_t* y_ptr, int n_graphs,
                              int dim, int n_queries, int k, int32_t* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_queries <= 0) return 0;
    if (k < 1 || k > 8 || dim < 1 || dim > 3) return -2;
    hipLaunchKernelGGL(k_knn_wave, dim3(nb(n_queries, 4)), dim3(256), 0, S_, x, x_ptr, y, y_ptr, n_graphs, dim, n_queries, k, out);
    LAUNCH_CHECK();
}

extern "C" int magnet_segment_mean_f32(const float* msg, const int32_t* rowptr, int n_rows, int h, float* out,
                                       void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (n_rows <= 0) return 0;
    const dim3 grid(nb(n_rows, 4)), block(256);
    switch (h) {
        case 32: hipLaunchKernelGGL(k_segment_mean<32>, grid, block, 0, S_, msg, rowptr, n_rows, out); break;
        case 64: hipLaunchKernelGGL(k_segment_mean<64>, grid, block, 0, S_, msg, rowptr, n_rows, out); break;
        case 128: hipLaunchKernelGGL(k_segment_mean<128>, grid, block, 0, S_, msg, rowptr, n_rows, out); break;
        case 256: hipLaunchKernelGGL(k_segment_mean<256>, grid, block, 0, S_, msg, rowptr, n_rows, out); break;
        default: return -2;
    }
    LAUNCH_CHECK();
}

// Measurement utility (bench.py `roofline.clock_mhz`, no reference counterpart): ONE wave samples the shader clock for
// ~wall_us microseconds on `stream` while other streams run the workload; out[0] = shader-clock ticks (s_memtime),
// out[1] = constant-rate ticks (s_memrealtime).  MHz = out[0] / out[1] * wall_khz / 1000.
__global__ void __launch_bounds__(64) k_clock_probe(long long* __restrict__ out, long long wall_ticks) {
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    long long w;
    do {
        __builtin_amdgcn_s_sleep(64);
        w = wall_clock64();
    } while (w - w0 < wall_ticks);
    const long long c1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w - w0;
    }
}

extern "C" int magnet_clock_probe(long long* out, int wall_us, int* wall_khz, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    int dev = 0, khz = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) return -2;
    if (wall_khz) *wall_khz = khz;
    if (wall_us <= 0 || wall_us > 2000000) return -2;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, S_, out, (long long)khz * wall_us / 1000);
    LAUNCH_CHECK();
}
