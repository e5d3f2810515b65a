// Fused backward of the MPNN edge block, generation 3 (k_mpnn_edge_bwd_co): the workgroup's eight waves CO-OPERATE on one 32-edge tile per
// step and every wave owns 16 FEATURES of everything - the construction of k_linear_bwd_r (linear_bwd.hip) applied to the three GEMMs of
// this block.  Same arguments and outputs as k_mpnn_edge_bwd_wp (generation 2, mpnn_edge_bwd_wp.hip).
// Reference: torch autograd through GNN_Layer.message / aggr='mean', models/mpnn.py:73-79,46.
//
// Why (profiles/r5/pmc_edge_bwd_wp_sq_counters.txt): generation 2 keeps the whole [128 x 128] fp32 accumulator of dW2 in
// every wave - 256 registers, hence ONE wave per SIMD, which executes an instruction only 50 % of the time (its own waits).  Here
//  * wave w owns features 16 w .. 16 w + 15: rows a of q = W2 s1 (GEMM 1), rows b of ds1 = W2^T dq (GEMM 2), rows a of dW2 = dq^T s1; its
//    A fragments of W2 and of W2^T (4 k tiles x 2 planes each: 32 + 32 registers, read once from the standard fp16x3 packs) and its
//    [16 x 128] slab of dW2 (32 registers) are resident - 96 registers, no weights in LDS, two waves per SIMD;
//  * everything element-wise happens in the C/D layout of v_mfma_f32_16x16x32_f16 - lane (n16, q4) = edges 2 n16 and 2 n16 + 1 of the tile,
//    features 16 w + 4 q4 .. + 3: p = A[i] + Bm[j] (64-byte pieces of the rows: the eight waves together read every row once), s1,
//    Swish'(p) (kept in 8 registers until the dp phase: no parking block), dagg[i] / deg, dq, dp, and the row sums dA over the 16-lane DPP
//    rows (24 instructions per wave and tile instead of 640);
//  * only the fp16 planes of s1 and of G dq go through LDS, split ONCE by the wave that formed them, row-major [edge][position] (LR_ST of
//    linear_bwd.hip: 288-byte rows): the B operands of GEMM 1 / GEMM 2 are one ds_read_b128 per k slot and plane, the operands of the
//    weight gradient (lane = feature, registers = edges) come out of ds_read_b64_tr_b16;
//  * ONE workgroup barrier per tile: the dq planes are double buffered, the s1 planes have THREE images - the next tile's s1 is staged in
//    front of the tile's barrier, which publishes it together with the tile's dq (two barriers on two images: 3.83 ms against 3.72);
//  * the older wave of a SIMD wins every arbitration (priority, then age) and would idle ~2 000 cycles at the barrier while its partner
//    finishes alone: waves 4 .. 7 run at priority 1, waves 0 .. 3 at 2 behind the barrier and at 0 from the end of the matrix part
//    (CO_FLIP_PRIO; 3.59 ms against 3.68 with a static priority, phase traces in profiles/r6/edge_bwd_co_trace_*.txt).
// The kernel issues ~670 instructions per wave and tile (8 waves: 5 400 per tile; generation 2: 4 900 in one wave) and fills about 75 % of
// its SIMDs' issue slots (generation 2: 50 %) - DESIGN.md 15.1 has the measurements.
// Arithmetic as generation 2: fp16 split with one accumulator (W planes 2^11 hw, 2^11 lw), G = one power of two per launch for dq (serves
// GEMM 2 and the weight gradient), fixed summation orders (bit-identical repeats).
// (the plane words this kernel forms in inline asm go to LDS, never straight into an MFMA: the two wait states that MAGNET_MIX_TAIL appends
//  for hipcc's blind spot - DESIGN.md 10.10 - are eight wasted issue slots per tile here; hazard_check.py polices the build as always)
#define MAGNET_MIX_TAIL ""
#include "common.cuh"
#include "split_single.cuh"

#define CO_ST 144                       // halfwords per LDS row: 288 B (see LR_ST in linear_bwd.hip)
#ifndef CO_PACKED
#define CO_PACKED 1                     // element-wise arithmetic on whole f32x4 vectors (v_pk_*_f32) / 0: value by value
#endif
#ifndef CO_ONE_BARRIER
#define CO_ONE_BARRIER 1                // 1: three s1 images, the next tile's s1 staged in front of the tile's only barrier; 0: two barriers
#endif
#ifndef CO_FLIP_PRIO
#define CO_FLIP_PRIO 1
#endif
#ifndef CO_YOUNG_PRIO
#define CO_YOUNG_PRIO 0
#endif
#ifndef CO_ABL
#define CO_ABL 0                        // development ablations (timing only, results invalid): 1 no barriers, 2 no dW2 MFMAs, 4 no GEMM 1 / 2
#endif                                  //   MFMAs, 8 no dp stores, 16 no transcendentals, 32 gathers from 64 fixed rows (cache resident), 64 no LDS reads

#ifdef CO_TRACE                         // development: s_memtime stamps of workgroup 0's waves at the phase boundaries of 32 steps
__device__ long long g_co_trace[8][32][10];
extern "C" int magnet_debug_co_trace(long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_co_trace), sizeof(g_co_trace)); }
#define CO_STAMP(k) do { if (tr_on) tr[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CO_STAMP(k) do { } while (0)
#endif

namespace {

typedef short co_s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(1))) const char* co_gconst;
typedef __attribute__((address_space(1))) char* co_gbytes;

// One fp16-plane image of a tile: [plane][edge row][position]; within every block of 16 features, feature 8 hi + 4 kg + c sits at position
// 8 kg + 4 hi + c (the k order of the standard pack's A fragments; linear_bwd.hip LrBuf)
typedef unsigned short CoImg[2][32][CO_ST];

__device__ __forceinline__ int co_sw(int i) { return (i & 3) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ f32x4 co_ldg4(const float* base, unsigned off) {
    return *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>((co_gconst)(base) + off);
}
__device__ __forceinline__ void co_stg4(float* base, unsigned off, const f32x4& v) {
    *reinterpret_cast<__attribute__((address_space(1))) f32x4*>((co_gbytes)(base) + off) = v;
}
__device__ __forceinline__ int co_ldi(const int32_t* base, unsigned off) {
    return *reinterpret_cast<__attribute__((address_space(1))) const int*>((co_gconst)(base) + off);
}
__device__ __forceinline__ u32x2 co_tr(const unsigned short* p) {
    if (CO_ABL & 64) return u32x2{(unsigned)(size_t)p, 0x3c003c00u};
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) co_s16x4*)(p)));
}
__device__ __forceinline__ float co_dot_ones(unsigned w, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), one, acc, false);
}
__device__ __forceinline__ float co_sigmoid(float x) {
    if (CO_ABL & 16) return x * 0.25f;
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
// 1 / (1 + 2^x) of four values (x = -log2(e) * the sigmoid's argument)
__device__ __forceinline__ f32x4 co_sigmoid4(const f32x4& x) {
    if (CO_ABL & 16) return x * 0.25f;
    f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(x[i]);
    e = e + 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
    return e;
}
// 1 / (1 + 2^-x): the negation rides in v_exp_f32's source modifier
__device__ __forceinline__ f32x4 co_sigmoid4n(const f32x4& x) {
    if (CO_ABL & 16) return x * -0.25f;
    f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(-x[i]);
    e = e + 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
    return e;
}
__device__ __forceinline__ void co_barrier() {
    if (!(CO_ABL & 1)) wg_barrier();
}
#define DPP_SHL1 0x101

struct CoIdx {                          // one tile's indices of this lane: edges 2 n16 (A) and 2 n16 + 1 (B), clamped to the last edge
    int tA, tB, sA, sB;
};

// acc[rb] += 2^11 W x for the wave's 16 output features x the tile's edges 16 rb + n16: ALL sixteen operand reads of the phase are issued
// first (hipcc's own order waited lgkmcnt(0) two reads after the barrier and chained up to twelve dependent MFMAs on one accumulator),
// the MFMAs then alternate between the two accumulators
__device__ __forceinline__ void co_gemm(const unsigned short (&img)[2][32][CO_ST], const f16x8 (&wh)[4], const f16x8 (&wl)[4], f32x4 (&acc)[2],
                                        int n16, int q4) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                                          // two batches of eight reads (sixteen at once: 64 registers, spills)
        f16x8 zh[2][2], zl[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int kt = 2 * hf + k2;
                if (CO_ABL & 64) { zh[k2][rb] = wh[kt], zl[k2][rb] = wl[(kt + rb) & 3]; continue; }
                zh[k2][rb] = *reinterpret_cast<const f16x8*>(&img[0][16 * rb + n16][32 * kt + 8 * q4]);
                zl[k2][rb] = *reinterpret_cast<const f16x8*>(&img[1][16 * rb + n16][32 * kt + 8 * q4]);
            }
#ifdef CO_GEMM_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kt = 2 * hf + k2;
            if (CO_ABL & 4) {
                acc[0][0] += (float)zh[k2][0][0] + (float)zl[k2][0][1], acc[1][0] += (float)zh[k2][1][0] + (float)zl[k2][1][1];
                continue;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {                                  // (the chain's first product starts from the inline constant 0:
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};                      //  a zeroed accumulator variable costs four v_mov per block)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[kt], zh[k2][rb], kt == 0 ? zero : acc[rb], 0, 0, 0);
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kt], zl[k2][rb], acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kt], zh[k2][rb], acc[rb], 0, 0, 0);
        }
    }
}

}  // namespace

// AB_L2E: the rows of A / Bm carry the factor log2(e) (the forward's halves for the exp2-based Swish: A' + Bm' = log2(e) p)
template <bool AB_L2E>
__global__ void __launch_bounds__(512, 1) k_mpnn_edge_bwd_co(const float* __restrict__ A, const float* __restrict__ Bm,
                                                             const float* __restrict__ dagg, const int32_t* __restrict__ src,
                                                             const int32_t* __restrict__ tgt, const int32_t* __restrict__ rowptr,
                                                             int E, const u32x4* __restrict__ w2s, const u32x4* __restrict__ w2ts,
                                                             const float* __restrict__ b2, float* __restrict__ dp_out,
                                                             float* __restrict__ dA, float* __restrict__ partials,
                                                             const float* __restrict__ amax_part, int n_amax,
                                                             float* __restrict__ wpart, float* __restrict__ bpart) {
    __shared__ __attribute__((aligned(16))) CoImg s1b[CO_ONE_BARRIER ? 3 : 2];    // s1 planes of tiles t, t + 1 (, t + 2)
    __shared__ __attribute__((aligned(16))) CoImg dqb[2];                         // G dq planes
    __shared__ float gsc[8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, q4 = lane >> 4;
    float G, invG;
    {
        float m = 0.f;
        for (int i = threadIdx.x; i < n_amax; i += 512) m = fmaxf(m, amax_part[i]);
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) gsc[wave] = m;
        __syncthreads();
        m = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) m = fmaxf(m, gsc[w]);
        const unsigned eb = (__float_as_uint(m) >> 23) & 0xFFu;               // max|dagg| in [2^(eb-127), 2^(eb-126))
        const bool ok = eb >= 14u && eb < 255u;
        G = ok ? __uint_as_float((267u - eb) << 23) : 1.0f;                   // 2^(140-eb): G * 1.1 max|dagg| < 2^15
        invG = ok ? __uint_as_float((eb - 13u) << 23) : 1.0f;
    }
    // ---- the wave's 16 rows of W2 (GEMM 1: q features) and of W2^T (GEMM 2: s1 features) as A fragments: lane = row n16, k group q4 =
    // the pack's k slots (kt, q4 >> 1, q4 & 1)
    f16x8 w1h[4], w1l[4], w2h[4], w2l[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int idx = (((kt * 4 + (wave >> 1)) * 2 + (q4 >> 1)) * 2) * 64 + 16 * (wave & 1) + n16 + 32 * (q4 & 1);
        w1h[kt] = __builtin_bit_cast(f16x8, scale_plane_2048(w2s[idx]));
        w1l[kt] = __builtin_bit_cast(f16x8, w2s[idx + 64]);
        w2h[kt] = __builtin_bit_cast(f16x8, scale_plane_2048(w2ts[idx]));
        w2l[kt] = __builtin_bit_cast(f16x8, w2ts[idx + 64]);
    }
    const unsigned fcol = 16u * (unsigned)wave + 4u * (unsigned)q4;           // this lane's first feature
    f32x4 bq, bql;                                                            // b2 of its four q features, and - log2(e) b2: the bias joins q in the
#pragma unroll                                                                // two fused multiply-adds that take 2^11 q apart (the MFMAs start from 0)
    for (int i = 0; i < 4; ++i) bq[i] = b2[fcol + i], bql[i] = b2[fcol + i] * -1.4426950408889634f;
    const int ppos = 16 * wave + 8 * (q4 & 1) + 4 * (q4 >> 1);                // where its four features sit in an LDS row
    // ---- the workgroup's contiguous range of tiles (consecutive workgroups of an XCD take consecutive ranges)
    const int ntiles = (E + 31) / 32;
    const int nb8 = gridDim.x / 8;
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tile0 = vb * tpw;
    const int nsteps = __builtin_amdgcn_readfirstlane(max(0, min(tpw, ntiles - tile0)));

    // every per-lane address is a wave-uniform base (SGPR pair, made opaque) + a 32-bit byte offset (DESIGN.md 10 rule 1)
    auto load_idx = [&](int tile_) {
        CoIdx x;
        const int ea = min(min(tile_, ntiles - 1) * 32 + 2 * n16, E - 1), eb = min(ea + 1, E - 1);
        const int32_t *pt = tgt, *ps = src;
        asm volatile("" : "+s"(pt), "+s"(ps));
        x.tA = co_ldi(pt, 4u * (unsigned)ea), x.tB = co_ldi(pt, 4u * (unsigned)eb);
        x.sA = co_ldi(ps, 4u * (unsigned)ea), x.sB = co_ldi(ps, 4u * (unsigned)eb);
        if (CO_ABL & 32) x.tA &= 63, x.tB &= 63, x.sA &= 63, x.sB &= 63;
        return x;
    };
    // Rows in flight: pa / pb = A[i], Bm[j] of the tile whose s1 is produced in THIS iteration (tile + 1) - re-requested for tile + 2 as
    // soon as the production has read them (two barriers: a second set qa / qb, requested at the top of the iteration) -, gv / rs / re =
    // dagg[i] and the row bounds of the tile whose dq is formed in the next iteration.
    f32x4 pa[2], pb[2], qa[2], qb[2], gv[2];
    int rs[2], re[2];
    auto load_ab = [&](const CoIdx& x, f32x4 (&a_)[2], f32x4 (&b_)[2]) {
        const float *ba = A, *bb = Bm;
        asm volatile("" : "+s"(ba), "+s"(bb));
        a_[0] = co_ldg4(ba, (unsigned)x.tA * 512u + 4u * fcol), a_[1] = co_ldg4(ba, (unsigned)x.tB * 512u + 4u * fcol);
        b_[0] = co_ldg4(bb, (unsigned)x.sA * 512u + 4u * fcol), b_[1] = co_ldg4(bb, (unsigned)x.sB * 512u + 4u * fcol);
    };
    auto load_g = [&](const CoIdx& x) {
        const float* pg = dagg;
        const int32_t* pr = rowptr;
        asm volatile("" : "+s"(pg), "+s"(pr));
        gv[0] = co_ldg4(pg, (unsigned)x.tA * 512u + 4u * fcol), gv[1] = co_ldg4(pg, (unsigned)x.tB * 512u + 4u * fcol);
        rs[0] = co_ldi(pr, 4u * (unsigned)x.tA), re[0] = co_ldi(pr, 4u * (unsigned)x.tA + 4u);
        rs[1] = co_ldi(pr, 4u * (unsigned)x.tB), re[1] = co_ldi(pr, 4u * (unsigned)x.tB + 4u);
    };
    f32x4 vone = {1.0f, 1.0f, 1.0f, 1.0f}, vmone = {-1.0f, -1.0f, -1.0f, -1.0f};
    asm volatile("" : "+v"(vone), "+v"(vmone));                               // (in registers: hipcc scalarises packed arithmetic on literals / negations)
    // ---- phase A of a tile: s1 = Swish(p) as fp16 planes into LDS buffer `B`, Swish'(p) into `sp` (registers)
    auto produce = [&](CoImg& S, const f32x4 (&a_)[2], const f32x4 (&b_)[2], f32x4 (&sp)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            // (whole-vector arithmetic: v_pk_add / v_pk_mul / v_pk_fma_f32 take two of the lane's four values per instruction)
#if CO_PACKED
            const f32x4 vs = a_[rb] + b_[rb];
            const f32x4 v = AB_L2E ? vs * 0.6931471805599453f : vs;            // p
            const f32x4 s = AB_L2E ? co_sigmoid4n(vs) : co_sigmoid4(vs * (-1.4426950408889634f));
            const f32x4 t = v * s;
            sp[rb] = t * (s * vmone + vone) + s;                                // s (1 + v (1 - s)): two v_pk_fma (a literal 1 - s is taken apart)
#else
            f32x4 v, s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = (a_[rb][i] + b_[rb][i]) * (AB_L2E ? 0.6931471805599453f : 1.0f);
                s[i] = co_sigmoid(v[i]);
                sp[rb][i] = s[i] * (1.0f + v[i] * (1.0f - s[i]));
            }
#endif
            unsigned h0, l0, h1, l1;
            split_f16_prod(v[0], s[0], v[1], s[1], h0, l0);
            split_f16_prod(v[2], s[2], v[3], s[3], h1, l1);
            const u32x2 h = {h0, h1}, l = {l0, l1};
            *reinterpret_cast<u32x2*>(&S[0][16 * rb + n16][ppos]) = h;
            *reinterpret_cast<u32x2*>(&S[1][16 * rb + n16][ppos]) = l;
        }
    };
    // ---- prologue: tile 0's s1 staged, its dagg rows and tile 1's A / Bm rows requested
    f32x4 spp[2], sppn[2];
    CoIdx i0 = load_idx(tile0), i1 = load_idx(tile0 + 1), i2 = load_idx(tile0 + 2);
    int ctA = i0.tA, ctB = i0.tB;                                             // targets of the current tile's two edges
    if (nsteps > 0) {
        load_ab(i0, pa, pb);
        load_g(i0);
        produce(s1b[0], pa, pb, spp);
        load_ab(i1, pa, pb);
#if CO_ONE_BARRIER
        co_barrier();                                                         // (tile 0's s1: every later tile's is published by the loop's barrier)
#endif
    }

    // dW2 in an 2 x 4 arrangement of 16 x 16 blocks per wave - rows (dq features) 32 wr .., columns (s1 features) 64 wc .. - so that a wave
    // reads 2 + 4 transposed operand blocks per tile instead of 1 + 8
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 accw[2][4];                                                         // G dW2[32 wr + 16 (a ^ wc) + co_sw(4 q4 + i)][64 wc + 16 b + co_sw(n16)]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) accw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                                         // G db2[32 wr + 16 wc + co_sw(n16)], this lane's edges
    const float usc = invG * (1.0f / 2048.0f);

#if CO_FLIP_PRIO
    // The older wave of a SIMD wins every arbitration (priority, then age), reaches the tile's barrier ~2 000 cycles early and idles there while
    // its partner finishes ALONE (a lone in-order wave fills about half of the issue slots; trace: profiles/r6/edge_bwd_co_trace_*.txt).
    // Waves 4 .. 7 sit at priority 1; waves 0 .. 3 take priority 2 behind the barrier and drop to 0 at CO_FLIP_PRIO's point of the
    // iteration (1: behind the matrix part, 2: behind dp / scan, 3: behind the next tile's s1), so that the partner catches up while both
    // still have work.
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#elif CO_YOUNG_PRIO
    // the second-dispatched half of the workgroup loses every arbitration against its older SIMD partner (priority, then age): it reaches
    // the barriers ~2 500 cycles late and the older half waits there (trace: profiles/r6/edge_bwd_co_trace_v1.txt) - static priority 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(CO_YOUNG_PRIO);
#endif
    // the sum of the row that is open at a tile's end travels to the next tile in registers (lane 15 of every 16-lane row holds it behind
    // the scan); only the rows cut by the WORKGROUP's range go through `partials` ([workgroup range][2][128]: slot 0 = the range's part of
    // the row that entered it, 1 = of the row that leaves it) and k_segsum_fixup_wg
    const int e00 = tile0 * 32;
    f32x4 carry = {0.f, 0.f, 0.f, 0.f};
    int k3 = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int tile = tile0 + step;
#if CO_ONE_BARRIER
        CoImg& S = s1b[k3];
        CoImg& Sn = s1b[k3 == 2 ? 0 : k3 + 1];
        k3 = k3 == 2 ? 0 : k3 + 1;
#else
        CoImg& S = s1b[step & 1];
        CoImg& Sn = s1b[(step + 1) & 1];
#endif
        CoImg& D = dqb[step & 1];
#ifdef CO_TRACE
        const bool tr_on = blockIdx.x == 0 && step >= 100 && step < 132;
        long long tr[10];
#endif
        CO_STAMP(0);
        const int e0 = tile * 32, eA = e0 + 2 * n16;
        const bool okA = eA < E, okB = eA + 1 < E;
        // what the dq / dA phases need of this tile's bookkeeping, before the registers are handed to the next tile's loads
        const float idA = okA ? __builtin_amdgcn_rcpf((float)(re[0] - rs[0])) : 0.0f;     // (1 / deg to 1 ulp; invalid edges: dq = 0)
        const float idB = okB ? __builtin_amdgcn_rcpf((float)(re[1] - rs[1])) : 0.0f;
        const int tA = okA ? ctA : (-16 - 2 * n16), tB = okB ? ctB : (-17 - 2 * n16);
        const int rsA = rs[0], reA = re[0], rsB = rs[1], reB = re[1];
        // ---- requests: A / Bm rows of tile + 2, indices of tile + 3 (past the range: the clamped last tile again, unused)
        asm volatile("" ::: "memory");
#if !CO_ONE_BARRIER
        load_ab(i2, qa, qb);
#endif
        const CoIdx inext = i1, iab = i2;
        ctA = i1.tA, ctB = i1.tB;
        i1 = i2;
        i2 = load_idx(tile + 3);
        asm volatile("" ::: "memory");
        CO_STAMP(1);
#if CO_ONE_BARRIER
        // ONE barrier per tile: the next tile's s1 is staged HERE, in front of this tile's only barrier (which publishes it together with
        // this tile's dq), into the third s1 image - its last readers (the weight gradient of tile - 2) passed the previous barrier
        produce(Sn, pa, pb, sppn);
        asm volatile("" ::: "memory");
        load_ab(iab, pa, pb);                                                 // tile + 2's rows: into the registers the production has just released
        asm volatile("" ::: "memory");                                        // (a whole iteration ahead of their use; no second register set, no copies)
#else
        co_barrier();                                                         // s1 of the tile staged (by the previous iteration)
#endif
        CO_STAMP(2);
#if CO_FLIP_PRIO == 3
        if (wave < 4) __builtin_amdgcn_s_setprio(0);
#endif
        // ---- GEMM 1: 2^11 q = 2^11 (W2 s1 + b2) for the wave's 16 q features x 32 edges; dq = dagg[i] / deg * Swish'(q)
        f32x4 acc[2];
        co_gemm(S, w1h, w1l, acc, n16, q4);
        CO_STAMP(3);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float Gd = G * (rb ? idB : idA);                            // 1 / deg rides in the split's multiplier
            // Swish'(q) with q = acc / 2^11: the scale rides in the constants
#if CO_PACKED
            const f32x4 s = co_sigmoid4(acc[rb] * (-1.4426950408889634f / 2048.0f) + bql);
            const f32x4 t = (acc[rb] * (1.0f / 2048.0f) + bq) * s;
            const f32x4 d = gv[rb] * (t * (s * vmone + vone) + s);
#else
            f32x4 d;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float q = __builtin_fmaf(acc[rb][i], 1.0f / 2048.0f, bq[i]);
                const float s = (CO_ABL & 16) ? q * 0.25f : __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(q * -1.4426950408889634f));
                d[i] = (gv[rb][i] * s) * __builtin_fmaf(q, 1.0f - s, 1.0f);
            }
#endif
            unsigned h0, l0, h1, l1;
            split_f16_prod(d[0], Gd, d[1], Gd, h0, l0);
            split_f16_prod(d[2], Gd, d[3], Gd, h1, l1);
            const u32x2 h = {h0, h1}, l = {l0, l1};
            *reinterpret_cast<u32x2*>(&D[0][16 * rb + n16][ppos]) = h;
            *reinterpret_cast<u32x2*>(&D[1][16 * rb + n16][ppos]) = l;
        }
        asm volatile("" ::: "memory");
        load_g(inext);                                                        // dagg rows / row bounds of tile + 1: into the registers dq has just released
        asm volatile("" ::: "memory");
        CO_STAMP(4);
        co_barrier();                                                         // G dq of the tile staged
        CO_STAMP(5);
#if CO_FLIP_PRIO
        if (wave < 4) __builtin_amdgcn_s_setprio(2);
#endif
        // ---- behind the barrier: the matrix part (GEMM 2 + the weight gradient: 48 MFMAs)
        f32x4 x2[2];
        auto matrix_part = [&]() {
            // GEMM 2: 2^11 G ds1 = 2^11 W2^T (G dq) for the wave's 16 s1 features
            co_gemm(D, w2h, w2l, x2, n16, q4);
            // dW2 blocks += (G dq)^T s1 over the tile's 32 edges (k order: lane group q4 takes LDS rows 4 q4 .. + 3 and 16 + 4 q4 .. + 3), db2
            const int r0 = 4 * q4 + (n16 >> 2), c0 = 4 * (n16 & 3);
            f16x8 ah[2], al[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {                                     // accw[a] = row block a ^ wc: the pair's two waves start on different blocks
                const int f = 32 * wr + 16 * (a ^ wc) + c0;
                const u32x2 a0 = co_tr(&D[0][r0][f]), a1 = co_tr(&D[0][16 + r0][f]);
                const u32x2 b0 = co_tr(&D[1][r0][f]), b1 = co_tr(&D[1][16 + r0][f]);
                ah[a] = __builtin_bit_cast(f16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
                al[a] = __builtin_bit_cast(f16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
                if (a == 0) {                                                 // db2 = colsum(G dq) of block wc: each wave of the pair sums one block
                    float cs = co_dot_ones(a0[0], 0.f);
                    cs = co_dot_ones(a0[1], cs), cs = co_dot_ones(a1[0], cs), cs = co_dot_ones(a1[1], cs);
                    cs = co_dot_ones(b0[0], cs), cs = co_dot_ones(b0[1], cs), cs = co_dot_ones(b1[0], cs), cs = co_dot_ones(b1[1], cs);
                    bsum += cs;
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int f = 64 * wc + 16 * b + c0;
                const u32x2 p0 = co_tr(&S[0][r0][f]), p1 = co_tr(&S[0][16 + r0][f]);
                const u32x2 q0 = co_tr(&S[1][r0][f]), q1 = co_tr(&S[1][16 + r0][f]);
                const f16x8 xh = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p1[0], p1[1]});
                const f16x8 xl = __builtin_bit_cast(f16x8, u32x4{q0[0], q0[1], q1[0], q1[1]});
                if (CO_ABL & 2) { accw[0][b][0] += (float)xh[0] + (float)xl[1]; continue; }
#pragma unroll
                for (int a = 0; a < 2; ++a) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], xh, accw[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], xl, accw[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], xh, accw[a][b], 0, 0, 0);
            }
        };
        matrix_part();
        CO_STAMP(6);
#if CO_FLIP_PRIO == 1
        if (wave < 4) __builtin_amdgcn_s_setprio(0);
#endif
        // ---- dp = ds1 * Swish'(p)
        {
            float* dpt = dp_out + (size_t)tile * 4096;
            asm volatile("" : "+s"(dpt));
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                x2[rb] = x2[rb] * usc * spp[rb];
                unsigned off = (unsigned)(2 * n16 + rb) * 512u + 4u * fcol;
                asm volatile("" : "+v"(off));                                 // (a loop-invariant offset becomes a hoisted - and spilled - 64-bit address)
                if (!(CO_ABL & 8) && (rb ? okB : okA)) co_stg4(dpt, off, x2[rb]);
            }
        }
        // ---- dA[i] = sum of dp over the row's edges: segmented inclusive sums over the tile's 32 edges (edge 2 n16 + rb: in the lane first,
        // then along the 16-lane DPP row), the last edge of every run stores - to dA, or, rows cut by the workgroup's range, to `partials`
        {
            const int tp1 = dpp_i<DPP_SHR1, 0xF>(tB), tp2 = dpp_i<DPP_SHR2, 0xF>(tB), tp4 = dpp_i<DPP_SHR4, 0xF>(tB);
            const int tp8 = dpp_i<DPP_SHR8, 0xF>(tB);
            const int tn1 = dpp_i<DPP_SHL1, 0xF>(tA);                         // target of the edge after edge B (lane n16 + 1's edge A)
            const float m01 = tA == tB ? 1.f : 0.f;
            const float m1 = ((n16 >= 1) & (tp1 == tB)) ? 1.f : 0.f, m2 = ((n16 >= 2) & (tp2 == tB)) ? 1.f : 0.f;
            const float m4 = ((n16 >= 4) & (tp4 == tB)) ? 1.f : 0.f, m8 = ((n16 >= 8) & (tp8 == tB)) ? 1.f : 0.f;
            const float mC = ((n16 >= 1) & (tp1 == tA)) ? 1.f : 0.f;
            // x1 += m01 x0, four levels of x1 += m_d * x1[lane - d], x0 += mC * x1[lane - 1]: v_fmac_f32 with the DPP row shift on its own
            // source (hipcc emits v_mov_b32_dpp + v_fmac_f32: 44 instead of 24 instructions).  The four values are interleaved, so every
            // DPP read is three VALU instructions behind the write of its register (the hazard needs two wait states); lanes shifted in
            // from outside the 16-lane row read 0 (bound_ctrl) and their mask is 0 as well.
            float x00 = x2[0][0], x01 = x2[0][1], x02 = x2[0][2], x03 = x2[0][3];
            float x10 = x2[1][0], x11 = x2[1][1], x12 = x2[1][2], x13 = x2[1][3];
#define CO_L4(op) op(x10) "\n\t" op(x11) "\n\t" op(x12) "\n\t" op(x13) "\n\t"
            // (first: the tile's edge 0 takes what the previous tile left of its row - lane 15's sum, one lane round the 16-lane row)
            const float mIn = (n16 == 0 && step > 0 && rsA < e0) ? 1.f : 0.f;
            asm volatile("v_fmac_f32_dpp %0, %15, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %1, %16, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %2, %17, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32_dpp %3, %18, %14 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f32 %4, %8, %0\n\tv_fmac_f32 %5, %8, %1\n\tv_fmac_f32 %6, %8, %2\n\tv_fmac_f32 %7, %8, %3\n\t"
                         "v_fmac_f32_dpp %4, %4, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %5, %5, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %6, %6, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %7, %7, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %4, %4, %10 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %5, %5, %10 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %6, %6, %10 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %7, %7, %10 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %4, %4, %11 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %5, %5, %11 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %6, %6, %11 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %7, %7, %11 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %4, %4, %12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %5, %5, %12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %6, %6, %12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %7, %7, %12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %0, %4, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %1, %5, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %2, %6, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %3, %7, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1"
                         : "+v"(x00), "+v"(x01), "+v"(x02), "+v"(x03), "+v"(x10), "+v"(x11), "+v"(x12), "+v"(x13)
                         : "v"(m01), "v"(m1), "v"(m2), "v"(m4), "v"(m8), "v"(mC), "v"(mIn), "v"(carry[0]), "v"(carry[1]), "v"(carry[2]), "v"(carry[3]));
#undef CO_L4
            x2[0] = f32x4{x00, x01, x02, x03};
            x2[1] = f32x4{x10, x11, x12, x13};
            carry = x2[1];
            const bool endA = okA && tB != tA;
            const bool openB = n16 == 15 && reB > e0 + 32;                    // the tile's last edge, and its row goes on in the next tile
            const bool endB = okB && (n16 == 15 ? (!openB || step + 1 == nsteps) : tn1 != tB);
            float *pd = dA, *pp = partials + (size_t)vb * 256;
            asm volatile("" : "+s"(pd), "+s"(pp));
            if (endA) {                                                       // (edge B follows in the tile or does not exist: never open)
                if (rsA >= e00) co_stg4(pd, (unsigned)tA * 512u + 4u * fcol, x2[0]);
                else co_stg4(pp, 4u * fcol, x2[0]);
            }
            if (endB) {
                if (!openB && rsB >= e00) co_stg4(pd, (unsigned)tB * 512u + 4u * fcol, x2[1]);
                else co_stg4(pp, (rsB < e00 ? 0u : 512u) + 4u * fcol, x2[1]);
            }
        }
        CO_STAMP(7);
#if CO_FLIP_PRIO == 2
        if (wave < 4) __builtin_amdgcn_s_setprio(0);
#endif
#if !CO_ONE_BARRIER
        produce(Sn, pa, pb, sppn);                                            // (the next tile's s1: the last thing before its first barrier)
#endif
        CO_STAMP(8);
#ifdef CO_TRACE
        if (tr_on && lane == 0)
            for (int k = 0; k < 9; ++k) g_co_trace[wave][step - 100][k] = tr[k];
#endif
        // ---- hand the pipeline registers on (the rows of tile + 2 were requested at the top of this iteration)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#if !CO_ONE_BARRIER
            pa[rb] = qa[rb], pb[rb] = qb[rb];
#endif
            spp[rb] = sppn[rb];
        }
    }
    // ---- this workgroup's partial sums of dW2 / db2
    float* wout = wpart + (size_t)blockIdx.x * 128 * 128;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wout[(size_t)(32 * wr + 16 * (a ^ wc) + co_sw(4 * q4 + i)) * 128 + 64 * wc + 16 * b + co_sw(n16)] = accw[a][b][i] * invG;
    {
        float bs = bsum;
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);
        if (q4 == 0) bpart[(size_t)blockIdx.x * 128 + 32 * wr + 16 * wc + co_sw(n16)] = bs * invG;
    }
}

int magnet_launch_edge_bwd_co(const float* A, const float* Bm, const float* dagg, const int32_t* src, const int32_t* tgt,
                              const int32_t* rowptr, int n_edges, const void* w2_f16, const void* w2t_f16, const float* b2, float* dp,
                              float* dA, float* partials, const float* amax, int n_amax, float* wpart, float* bpart, int ab_l2e, int grid,
                              void* stream) {
    if (ab_l2e)
        hipLaunchKernelGGL(k_mpnn_edge_bwd_co<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream, A, Bm, dagg, src, tgt, rowptr, n_edges,
                           reinterpret_cast<const u32x4*>(w2_f16), reinterpret_cast<const u32x4*>(w2t_f16), b2, dp, dA, partials, amax,
                           n_amax, wpart, bpart);
    else
    hipLaunchKernelGGL(k_mpnn_edge_bwd_co<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream, A, Bm, dagg, src, tgt, rowptr, n_edges,
                       reinterpret_cast<const u32x4*>(w2_f16), reinterpret_cast<const u32x4*>(w2t_f16), b2, dp, dA, partials, amax, n_amax,
                       wpart, bpart);
    return (int)hipGetLastError();
}
