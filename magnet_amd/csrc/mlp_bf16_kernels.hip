// Generic register-chained MLP kernel on the bf16 matrix cores (exact hi/mid/lo split, see common.cuh "bf16x6").
//
// Same contract as k_mlp (mlp_kernels.hip): 32 rows per wave, layer 0 streamed from gathered / concatenated /
// differenced sources, hidden layers chained in registers, epilogues store | +residual | CSR mean.  Differences:
//   * every Linear is 6 v_mfma_f32_32x32x16_bf16 per 16-k slot instead of 8 v_mfma_f32_32x32x2_f32
//     (2.7x fewer matrix-pipe cycles, fp32-class accuracy);
//   * the split weight planes do not come from L2 per tile: the 4 waves of a workgroup walk the layers in
//     lockstep and the planes of one k tile (NT * 6 KiB) are DMA'd (global_load_lds, no VGPR round trip) into
//     an LDS ring one or two k tiles ahead (counted s_waitcnt) - one workgroup barrier per k tile (ring_barrier: a raw s_barrier,
//     because __syncthreads() drains every outstanding memory operation and with it the counted waits);
//   * activations are split per 8-k slot right before use, so the split's VALU work sits in MFMA shadows.
// layer[l].w points at that layer's planes in fragment order [kt][nt][m][plane][64 lanes][8 bf16]
// (packing.pack_linear_bf16x3), layer[l].b at the zero-padded fp32 bias.
#include "common.cuh"
#include <type_traits>
#include "edge_tile.cuh"
#ifndef MAGNET_BIAS_LDS
#define MAGNET_BIAS_LDS 1
#endif
// Tape of a training forward (args.hid[]): 1 = the activations a layer hands on are stored tile by tile inside the NEXT layer's k loop,
// four 16-byte stores per k tile behind the weight request, and the loop's waits are counted past them; 0 = all 16 stores of a layer
// at its end, drained by the next k tile's vmcnt(0) (rounds 3-5).
#ifndef MLP_TAPE_SPREAD
#define MLP_TAPE_SPREAD 1
#endif
// ... 1 = and as whole 128-byte lines through a wave-private LDS tile (eight rows x one line per store instruction instead of 32 rows x
// a quarter line)
// chunks of lead of the 5-Linear chains' weight ring (experiment builds; see DIST below)
#ifndef MLP_DIST5
#define MLP_DIST5 1
#endif
#ifndef MLP_TAPE_LINES
#define MLP_TAPE_LINES 1
#endif

template <int A, int B>
struct MaxI { static constexpr int v = A > B ? A : B; };

// DMA one chunk (npieces x 1 KiB, contiguous in global memory) into LDS.  Every wave issues exactly
// ceil(npieces / 4) loads (indices past the end repeat the last piece: same bytes to the same place), so that a
// counted s_waitcnt can leave exactly one younger chunk in flight.
template <int NPIECES>
__device__ __forceinline__ void dma_chunk(const f32x4* __restrict__ gw, u32x4* l, int wave, int lane) {
    const u32x4* g = reinterpret_cast<const u32x4*>(gw);
    constexpr int PW = (NPIECES + 3) / 4;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        int p = wave + 4 * i;
        p = p < NPIECES ? p : NPIECES - 1;
        __builtin_amdgcn_global_load_lds(g + p * 64 + lane, l + p * 64, 16, 0, 0);
    }
}
// The workgroup barrier of the weight ring.  __syncthreads() is a fence + s_barrier and hipcc drains EVERY outstanding vector memory
// operation in front of it (s_waitcnt vmcnt(0): it cannot know which LDS-DMA loads the barrier is to publish) - which turns every counted
// wait of this file into vmcnt(0) and exposes the full latency of the tape's stores.  What the barrier has to order is (a) this wave's
// DMA pieces of the chunk about to be read - the counted wait right before it - and (b) this wave's LDS traffic (bias staging, the
// fragment reads of the slot about to be overwritten): lgkmcnt(0).
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most the PW loads of the youngest chunk are outstanding
template <int NPIECES>
__device__ __forceinline__ void wait_all_but_one_chunk() {
    constexpr int PW = (NPIECES + 3) / 4;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
}

// acc[nt] += W(chunk)[nt] * x   for one k tile whose planes sit in LDS at `wl` ([nt][m][plane][64])
// acc[nt] += W(chunk)[nt] * x   for one k tile whose planes sit in LDS at `wl` ([nt][m][plane][64]).
// PIPE (the 128-wide 5-Linear chains, one wave per SIMD at ~400 registers): left alone hipcc emits
// "ds_read, s_waitcnt lgkmcnt(0), v_mfma" - a full LDS latency per MFMA group with nobody else on the SIMD to hide it.
// So the weight fragments of group g+1 are requested before the MFMAs of group g (double-buffered registers), the
// split of the second 16-k slot is spread over the groups of the first, and a sched_barrier pins the order
// (C5 edge MLP 2.42 -> 2.25 ms).  For the narrower / shorter chains, which run two waves per SIMD, the plain form is
// faster (the pinned order costs registers: +48 spills in the 2-Linear node update).
template <int NT, int MODE, bool PIPE>
__device__ __forceinline__ void mma_chunk(const f32x16& x, f32x16 (&accM)[NT], f32x16 (&accC)[NT], const u32x4* wl, int lane) {
    constexpr int NP = Split<MODE>::NP;
    if constexpr (MODE == GEMM_FP16X3S && NT % 2 == 0) {
        // one accumulator per output block: the blocks are taken in pairs so that consecutive MFMAs write different accumulators
        using S = Split<GEMM_FP16X3S>;
        constexpr int NPR = NT / 2, NG = 2 * NPR;          // groups gq = m * NPR + pair
        if (!PIPE) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                u32x4 xs[2];
                split_slot<MODE>(x, m, xs);
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    u32x4 w0[2], w1[2];
                    const u32x4* wp = wl + (((2 * pr) * 2 + m) * 2) * 64 + lane;
                    w0[0] = wp[0]; w0[1] = wp[64]; w1[0] = wp[4 * 64]; w1[1] = wp[5 * 64];
                    S::mma2(w0, w1, xs, accM[2 * pr], accM[2 * pr + 1]);
                }
            }
            return;
        }
        u32x4 xs[2][2], wr[2][2][2];
        auto wload = [&](int gq) {
            const int m = gq / NPR, pr = gq % NPR;
            const u32x4* wp = wl + (((2 * pr) * 2 + m) * 2) * 64 + lane;
            wr[gq & 1][0][0] = wp[0]; wr[gq & 1][0][1] = wp[64]; wr[gq & 1][1][0] = wp[4 * 64]; wr[gq & 1][1][1] = wp[5 * 64];
        };
        split_slot<MODE>(x, 0, xs[0]);
        wload(0);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            const int m = gq / NPR, pr = gq % NPR;
            if (gq + 1 < NG) wload(gq + 1);
            if (m == 0) {
#pragma unroll
                for (int p = pr * 4 / NPR; p < (pr + 1) * 4 / NPR; ++p) S::quarter(x, 1, p, xs[1]);
            }
            S::mma2(wr[gq & 1][0], wr[gq & 1][1], xs[m], accM[2 * pr], accM[2 * pr + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    if (!PIPE) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u32x4 xs[NP];
            split_slot<MODE>(x, m, xs);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4* wp = wl + ((nt * 2 + m) * NP) * 64 + lane;
                u32x4 w[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) w[q] = wp[64 * q];
                Split<MODE>::mma(w, xs, accM[nt], accC[nt]);
            }
        }
        return;
    }
    constexpr int NG = 2 * NT;                          // groups g = m * NT + nt
    constexpr int AHEAD = 2;                            // fragments requested this many groups ahead
    u32x4 xs[2][NP], wr[AHEAD + 1][NP];
    split_slot<MODE>(x, 0, xs[0]);
#pragma unroll
    for (int g0 = 0; g0 < AHEAD && g0 < NG; ++g0) {
        const u32x4* wp = wl + (((g0 % NT) * 2 + g0 / NT) * NP) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q) wr[g0][q] = wp[64 * q];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int m = g / NT, nt = g % NT;
        if (g + AHEAD < NG) {
            const int m1 = (g + AHEAD) / NT, n1 = (g + AHEAD) % NT;
            const u32x4* wp = wl + ((n1 * 2 + m1) * NP) * 64 + lane;
#pragma unroll
            for (int q = 0; q < NP; ++q) wr[(g + AHEAD) % (AHEAD + 1)][q] = wp[64 * q];
        }
        if (m == 0) {                                   // quarters of slot 1: 4 / NT per group
#pragma unroll
            for (int p = nt * 4 / NT; p < (nt + 1) * 4 / NT; ++p) Split<MODE>::quarter(x, 1, p, xs[1]);
        }
        Split<MODE>::mma(wr[g % (AHEAD + 1)], xs[m], accM[nt], accC[nt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NT, int MODE>
__device__ __forceinline__ void fold_tiles(f32x16 (&accM)[NT], f32x16 (&accC)[NT]) {
    if (MODE == GEMM_FP16X3 || MODE == GEMM_FP16X3S) {
#pragma unroll
        for (int n = 0; n < NT; ++n) Split<MODE>::fold(accM[n], accC[n]);
    }
}

// development ablation flags (results invalid): 1 = no MFMA/LDS-read work, 2 = no weight DMA after the prologue, 4 = no epilogue,
// 8 = no workgroup barriers, 16 = no gathered x loads after the first k tile.  Compiled in only with -DMAGNET_DEV_VARIANTS:
// even as never-taken run-time branches they cost the product kernel 20 % (the conditional loads break hipcc's software
// pipelining of the gathers).
#ifdef MAGNET_DEV_VARIANTS
__device__ int g_mlp_abl = 0;
#define MLP_ABL_INIT const int abl = __builtin_amdgcn_readfirstlane(g_mlp_abl);
// cycle trace (development builds): wave 0 of workgroup g_mlp_trace_wg writes s_memtime stamps with a phase tag
__device__ unsigned long long* g_mlp_trace = nullptr;
__device__ int g_mlp_trace_wg = 0;
#define TRACE_INIT int tr_n = 0; const bool tr_on = g_mlp_trace && (int)blockIdx.x == g_mlp_trace_wg && threadIdx.x == 0;
#define TRACE(tag) if (tr_on && tr_n < 500) { g_mlp_trace[2 * tr_n] = __builtin_readcyclecounter(); g_mlp_trace[2 * tr_n + 1] = (tag); ++tr_n; }
#else
#ifndef MLP_ABL_STATIC
#define MLP_ABL_STATIC 0                // (experiment builds: the same flags at compile time - the product kernel's own code with a piece left out)
#endif
#define MLP_ABL_INIT constexpr int abl = MLP_ABL_STATIC;
#define TRACE_INIT
#define TRACE(tag)
#endif

// DIST: chunks requested ahead.  1 for full grids: 2 (three-slot ring, counted vmcnt) paid +7 % while the wide 5-Linear chains ran
// one workgroup per CU and still costs 3 % with two per CU (C5 edge MLP 1.63 vs 1.59 ms).  2 for launches of at most one
// workgroup per CU (the launch-bound 1-D configs: 100 rows = one workgroup): nothing else on the CU hides the L2 -> LDS latency of
// a chunk (~1.8 us per k tile exposed), so the extra chunk of lead pays.
template <int MODE, int NL, int HT, int OT, int ACT, bool LAST_ACT, bool LN, int EPI, int DIST = 1>
__global__ void __launch_bounds__(256, 2) k_mlp_bf16(const MagnetMlpArgs a) {
    MLP_ABL_INIT
    TRACE_INIT
    TRACE(1)
    constexpr int NTMAX = MaxI<HT, OT>::v;
    constexpr int PPN = 2 * Split<MODE>::NP;                              // 1 KiB pieces per output tile and k tile
    constexpr int NBUF = DIST + 1;
    // fp16 split modes: the largest |activation| handed from one Linear of the chain to the next (args.status, magnet_hip.h) - one
    // v_max3_f32 per two values
    constexpr bool TRACK_SAT = NL > 1 && (MODE == GEMM_FP16X3 || MODE == GEMM_FP16X3S);
    // ... and of what the launch hands to whoever reads its output (another fp16 GEMM launch, as between the convolutions of EDSR),
    // unless a LayerNorm bounds it
    constexpr bool TRACK_OUT = !LN && (MODE == GEMM_FP16X3 || MODE == GEMM_FP16X3S);
    float sat = 0.f;
    constexpr bool PIPE = (NL >= 5 && HT >= 4) || DIST == 2;             // see mma_chunk; the one-workgroup-per-CU variants too:
                                                                         // nobody else on the SIMD hides the LDS latency of a group
    constexpr int CH = NTMAX * PPN * 64;                                   // u32x4 per chunk buffer
    // weight chunk ring; a 128-wide CSR-mean epilogue reuses it as four 32 x 68 float transposition tiles (34 KiB)
    constexpr bool LDS_MEAN = (EPI == EPI_SEGMEAN && OT == 4);
    constexpr int SMEM_U4 = (LDS_MEAN && NBUF * CH < 4 * 32 * 68 / 4) ? 4 * 32 * 68 / 4 : NBUF * CH;
#if MAGNET_BIAS_LDS
    // the biases of layers 1 .. NL-1 staged BEHIND the weight ring in the same __shared__ array and read with ds_read at the top of their
    // layer: from global memory they are 16 loads that are the wave's youngest VMEM operations right before the layer's first
    // `s_waitcnt vmcnt(0)` - an exposed L2 round trip per layer and tile (C5 edge chain 1.18 -> 1.14 ms, C4b 0.390 -> 0.376 ms).  Round 2
    // built this and found it intermittently wrong; round 3 found why (a missing hazard wait behind the inline-asm split, DESIGN.md
    // 10.10 - nothing to do with the biases) and ships it.  MAGNET_BIAS_LDS=0: biases from global memory (the variants of the
    // root-cause hunt, profiles/r3/bias_lds_root_cause.txt, lived here under values 2 .. 10)
    constexpr int BIAS_F = (NL > 1 ? NL - 1 : 1) * NTMAX * 32;
    // (tape, MLP_TAPE_LINES: one 32 x 32 transposition tile per wave behind the biases - store_tile_lines)
    constexpr bool TAPE_LINES = MLP_TAPE_LINES && MLP_TAPE_SPREAD && NL > 1;
    constexpr int TL_F = TAPE_LINES ? 32 * NH_TS : 0;
    __shared__ __attribute__((aligned(16))) u32x4 smem[SMEM_U4 + BIAS_F / 4 + 4 * TL_F / 4];
    float* sbias = reinterpret_cast<float*>(smem + SMEM_U4);
    if (NL > 1) {
        for (int t = threadIdx.x; t < BIAS_F; t += 256) {
            const int l = 1 + t / (NTMAX * 32), c = t % (NTMAX * 32);
            const int w = (l == NL - 1 ? OT : HT) * 32;
            sbias[t] = c < w ? a.layer[l].b[c] : 0.f;
        }
    }
#else
    __shared__ __attribute__((aligned(16))) u32x4 smem[SMEM_U4];
#endif
    u32x4* wbuf = smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
#if MAGNET_BIAS_LDS
    float* tl = sbias + BIAS_F + wave * TL_F;
#endif
    // XCD-aware row order: workgroups are dispatched round-robin over the 8 XCDs (each with its own L2); consecutive row
    // tiles - target-sorted edges gather neighbouring rows - go to the same XCD
    const int g8 = gridDim.x / 8;
    const int vbx = (int)blockIdx.x < g8 * 8 ? (int)(blockIdx.x % 8) * g8 + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int tile = vbx * 4 + wave;
    const int row_raw = tile * 32 + (lane & 31);
    const bool row_ok = row_raw < a.M;
    const int row = row_ok ? row_raw : (a.M - 1);
    const bool any_ok = tile * 32 + 32 <= a.M;          // (uniform) a whole tile: every row-masked store of the wave is an instruction that issues
    TileMeta md;                                        // CSR-mean epilogue: requested now, hidden behind the GEMMs
    if (EPI == EPI_SEGMEAN) md = tile_meta(tile, a.M, a.tgt, a.rowptr, lane & 31);

    constexpr int NT0 = (NL == 1) ? OT : HT;
    constexpr int NT1 = (NL == 2) ? OT : HT;            // output tiles of layer 1 (when it exists)
    // Column groups of a single-Linear launch: over gridDim.y, or - groups_in_block - one after the other inside the
    // workgroup, so that the second group finds its input rows in L2 instead of HBM (MPNN: the A and Bm halves of
    // message_net_1 share [h | node_in]).
    const int gib = (NL == 1 && a.groups_in_block > 1) ? a.groups_in_block : 1;
    for (int gi = 0; gi < gib; ++gi) {
        const int cg = (NL == 1) ? (gib > 1 ? gi : (int)blockIdx.y) : 0;
        if (gi > 0) __syncthreads();                    // every wave is done with the previous group's ring slots
        const MagnetLayer l0 = a.layer[cg];
        // Chunk c (one k tile of one layer) lives in ring slot c % NBUF and is requested DIST chunks ahead; iteration c:
        //   wait (DIST 2: counted, chunk c+1 stays in flight) -> barrier (chunk c visible, slot of c-1 free) ->
        //   request chunk c+DIST -> compute chunk c.
        int c = 0;                                         // running chunk index (uniform)
        dma_chunk<NT0 * PPN>(l0.w, wbuf, wave, lane);
        if (DIST == 2) {
            if (a.nkt > 1) dma_chunk<NT0 * PPN>(l0.w + (size_t)NT0 * PPN * 64, wbuf + CH, wave, lane);
            else if (NL > 1) dma_chunk<NT1 * PPN>(a.layer[1].w, wbuf + CH, wave, lane);
        }

        f32x16 out[OT];
        f32x16 h0[HT], h1[HT];
        // ---------------- layer 0: K streamed from the sources
        {
            f32x16 acc[NT0], accc[NT0];
            // per-node products of a factored first layer (magnet_hip.h: pre): pre[0] (which already contains the bias) IS the
            // accumulator's initial value and pre[1] the cross accumulator's (scaled by 2^11 after it has landed: the fold divides it
            // out again) - pure loads issued here, first touched after the first barrier, instead of 2 x 16 gathered loads that the
            // wave waited for between the first layer and its activation
            const bool pre_init = NL > 1 && (MODE == GEMM_FP16X3 || MODE == GEMM_FP16X3S) && a.pre[0].p && a.pre[1].p;
            if (pre_init) {
                load_rows<NT0>(acc, a.pre[0], row, half);
                load_rows<NT0>(accc, a.pre[1], row, half);
            } else {
                init_bias<NT0>(acc, l0.b, half);                            // (fp16x3s: the pack holds 2^11 b)
                if (MODE != GEMM_FP16X3S) zero_tiles<NT0>(accc);
            }
            KRaw xr = ktile_raw(a.kt[0], ktile_rows(a.kt[0], row), half);
            KRows rn = ktile_rows(a.kt[a.nkt > 1 ? 1 : 0], row);          // indices of k tile 1, requested one iteration early
            for (int kt = 0; kt < a.nkt; ++kt, ++c) {
                // outstanding, oldest first: chunk c, x tile kt, indices of kt+1, chunk c+1  ->  leave only chunk c+1 in flight
                TRACE(10)
                if (DIST == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (kt + 1 < a.nkt) wait_all_but_one_chunk<NT0 * PPN>();
                else if (NL > 1) wait_all_but_one_chunk<NT1 * PPN>();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TRACE(11)
                if (!(abl & 8)) ring_barrier();
                TRACE(12)
                const f32x16 x = ktile_finish(a.kt[kt], xr);               // x tile kt has landed: scales / sum of the sources now
                if (pre_init && kt == 0) {
#pragma unroll
                    for (int t = 0; t < NT0; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            if (MODE == GEMM_FP16X3S) acc[t][q] = (acc[t][q] + accc[t][q]) * 2048.0f;   // the one accumulator starts from both rows
                            else accc[t][q] *= 2048.0f;
                        }
                }
                // raw loads of x tile kt+1 (its indices arrived during the previous iteration), indices of x tile kt+2, then
                // chunk c+2: the next iteration's counted wait then covers both
                if (!(abl & 16)) xr = ktile_raw(a.kt[kt + 1 < a.nkt ? kt + 1 : kt], rn, half);
                rn = ktile_rows(a.kt[kt + 2 < a.nkt ? kt + 2 : a.nkt - 1], row);
                if (abl & 2) {
                } else if (kt + DIST < a.nkt) dma_chunk<NT0 * PPN>(l0.w + (size_t)(kt + DIST) * NT0 * PPN * 64, wbuf + ((c + DIST) % NBUF) * CH, wave, lane);
                else if (NL > 1 && kt + DIST - a.nkt < HT) dma_chunk<NT1 * PPN>(a.layer[1].w + (size_t)(kt + DIST - a.nkt) * NT1 * PPN * 64, wbuf + ((c + DIST) % NBUF) * CH, wave, lane);
                TRACE(13)
                if (!(abl & 1)) mma_chunk<NT0, MODE, PIPE>(x, acc, accc, wbuf + (c % NBUF) * CH, lane);
                TRACE(14)
                if (a.x0_out && gi == 0 && !((a.x0_skip >> kt) & 1)) store_ktile(x, a.x0_out, a.nkt * 32, kt * 32, row, row_ok, half);
            }
            fold_tiles<NT0, MODE>(acc, accc);
            if (NL > 1 && !pre_init) {                     // (other GEMM modes: added after the first layer)
                if (a.pre[0].p) add_rows<NT0>(acc, a.pre[0], row, half);
                if (a.pre[1].p) add_rows<NT0>(acc, a.pre[1], row, half);
            }
            if (NL == 1) {
    #pragma unroll
                for (int t = 0; t < NT0; ++t) out[t < OT ? t : 0] = acc[t];
            } else {
    #pragma unroll
                for (int t = 0; t < NT0; ++t) {
                    if (NL == 2 && a.hid[2]) {             // (two-Linear launches: the hidden layer's PRE-activation too - magnet_hip.h)
#if MAGNET_BIAS_LDS
                        if constexpr (TAPE_LINES) store_tile_lines<HT * 32>(acc[t], tl, a.hid[2], t * 32, (long)tile * 32, a.M, lane);
                        else
#endif
                        store_ktile(acc[t], a.hid[2], HT * 32, t * 32, row, row_ok, half);
                    }
                    apply_act<ACT>(acc[t]);
                    h0[t < HT ? t : 0] = acc[t];
                    if (TRACK_SAT) track_abs_max(acc[t], sat);
                }
                if (!MLP_TAPE_SPREAD && a.hid[0]) store_rows<HT, 0>(h0, a.hid[0], HT * 32, HT * 32, nullptr, 0, row, row_ok, half);
            }
        }
        // ---------------- layers 1 .. NL-1, activations in registers (fully unrolled: static register indices)
        // here the only VMEM traffic is the weight DMA, so the waits are counted: chunks c+1 (already requested) stays
        // in flight while chunk c is consumed.
    #pragma unroll
        for (int l = 1; l < NL; ++l) {
            const bool last = (l == NL - 1);
            const bool src0 = (l & 1);                     // odd layers read h0, even layers read h1
            f32x16 cc[NTMAX];
            if (MODE != GEMM_FP16X3S) zero_tiles<NTMAX>(cc);
            f32x16(&cco)[OT] = reinterpret_cast<f32x16(&)[OT]>(cc);
            f32x16(&cch)[HT] = reinterpret_cast<f32x16(&)[HT]>(cc);
#if MAGNET_BIAS_LDS
            {
                const float* sb = sbias + (l - 1) * NTMAX * 32;
                if (last) init_bias<OT>(out, sb, half);
                else if (src0) init_bias<HT>(h1, sb, half);
                else init_bias<HT>(h0, sb, half);
            }
#else
            if (last) init_bias<OT>(out, a.layer[l].b, half);
            else if (src0) init_bias<HT>(h1, a.layer[l].b, half);
            else init_bias<HT>(h0, a.layer[l].b, half);
#endif
    #pragma unroll
            for (int kt = 0; kt < HT; ++kt, ++c) {
                // chunk c+1: next k tile of this layer, or the first k tile of the next layer, or nothing (very last chunk)
                const bool has_next = (kt + 1 < HT) || (l + 1 < NL);
                const bool next_is_last_layer = (kt + 1 < HT) ? last : (l + 1 == NL - 1);
                TRACE(20)
                // (tape, MLP_TAPE_SPREAD: the iteration before this one issued the four stores of one 32-column tile of its layer's input
                // BEHIND the request of this chunk - memory operations complete in order, so vmcnt(4) says the chunk has landed and
                // leaves those stores in flight for a whole k tile instead of exposing the write latency to every wave)
                const bool prev_stored = MLP_TAPE_SPREAD && any_ok &&
                                         (kt > 0 ? a.hid[l - 1] != nullptr : (l >= 2 && a.hid[l >= 2 ? l - 2 : 0] != nullptr));
                constexpr int PWH = (HT * PPN + 3) / 4, PWO = (OT * PPN + 3) / 4;      // loads per wave of a chunk (dma_chunk)
                if (!has_next || DIST == 1) {
                    if (prev_stored) wait_vm<4>();
                    else wait_vm<0>();
                } else if (next_is_last_layer) {
                    if (prev_stored) wait_vm<PWO + 4>();
                    else wait_vm<PWO>();
                } else {
                    if (prev_stored) wait_vm<PWH + 4>();
                    else wait_vm<PWH>();
                }
                TRACE(21)
                if (!(abl & 8)) ring_barrier();
                TRACE(22)
                // chunk c+2
                if (!(abl & 2)) {
                    const int k2 = kt + DIST;
                    if (k2 < HT) {
                        if (last) dma_chunk<OT * PPN>(a.layer[l].w + (size_t)k2 * OT * PPN * 64, wbuf + ((c + DIST) % NBUF) * CH, wave, lane);
                        else dma_chunk<HT * PPN>(a.layer[l].w + (size_t)k2 * HT * PPN * 64, wbuf + ((c + DIST) % NBUF) * CH, wave, lane);
                    } else if (l + 1 < NL) {
                        const int k3 = k2 - HT;                               // 0 or 1: k tile of the next layer
                        if (l + 1 == NL - 1) dma_chunk<OT * PPN>(a.layer[l + 1 < NL ? l + 1 : l].w + (size_t)k3 * OT * PPN * 64, wbuf + ((c + DIST) % NBUF) * CH, wave, lane);
                        else dma_chunk<HT * PPN>(a.layer[l + 1 < NL ? l + 1 : l].w + (size_t)k3 * HT * PPN * 64, wbuf + ((c + DIST) % NBUF) * CH, wave, lane);
                    }
                }
                if (MLP_TAPE_SPREAD && a.hid[l - 1]) {
#if MAGNET_BIAS_LDS
                    if constexpr (TAPE_LINES) store_tile_lines<HT * 32>(src0 ? h0[kt] : h1[kt], tl, a.hid[l - 1], kt * 32, (long)tile * 32, a.M, lane);
                    else
#endif
                    store_ktile(src0 ? h0[kt] : h1[kt], a.hid[l - 1], HT * 32, kt * 32, row, row_ok, half);
                }
                TRACE(23)
                if (abl & 1) {
                } else if (last) mma_chunk<OT, MODE, PIPE>(src0 ? h0[kt] : h1[kt], out, cco, wbuf + (c % NBUF) * CH, lane);
                else if (src0) mma_chunk<HT, MODE, PIPE>(h0[kt], h1, cch, wbuf + (c % NBUF) * CH, lane);
                else mma_chunk<HT, MODE, PIPE>(h1[kt], h0, cch, wbuf + (c % NBUF) * CH, lane);
                TRACE(24)
            }
            if (last) fold_tiles<OT, MODE>(out, cco);
            else if (src0) fold_tiles<HT, MODE>(h1, cch);
            else fold_tiles<HT, MODE>(h0, cch);
            if (!last) {
    #pragma unroll
                for (int t = 0; t < HT; ++t) {
                    if (src0) apply_act<ACT>(h1[t]);
                    else apply_act<ACT>(h0[t]);
                    if (TRACK_SAT) track_abs_max(src0 ? h1[t] : h0[t], sat);
                }
                if (!MLP_TAPE_SPREAD && a.hid[l]) {
                    if (src0) store_rows<HT, 0>(h1, a.hid[l], HT * 32, HT * 32, nullptr, 0, row, row_ok, half);
                    else store_rows<HT, 0>(h0, a.hid[l], HT * 32, HT * 32, nullptr, 0, row, row_ok, half);
                }
            }
        }
        if (NL > 1 && a.hid[NL - 1]) {
#if MAGNET_BIAS_LDS
            if constexpr (TAPE_LINES) {
#pragma unroll
                for (int t = 0; t < OT; ++t) store_tile_lines<OT * 32>(out[t], tl, a.hid[NL - 1], t * 32, (long)tile * 32, a.M, lane);
            } else
#endif
            store_rows<OT, 0>(out, a.hid[NL - 1], OT * 32, OT * 32, nullptr, 0, row, row_ok, half);
        }
        if (LAST_ACT) {
    #pragma unroll
            for (int t = 0; t < OT; ++t) apply_act<ACT>(out[t]);
        }
        TRACE(30)
        if (TRACK_OUT && row_ok) {
#pragma unroll
            for (int t = 0; t < OT; ++t) track_abs_max(out[t], sat);
        }
        if (LN) layer_norm_rows<OT>(out, a.ln_g, a.ln_b, a.ln_eps, half);
        TRACE(31)

        if (abl & 4) {
            if (out[0][0] == 12345.678f) a.y[0] = 1.f;      // keep the computation alive
        } else if (EPI == EPI_SEGMEAN) {
            if constexpr (LDS_MEAN) {
                __syncthreads();                        // every wave is done with the weight ring
                segmean_epilogue_lds128(reinterpret_cast<const f32x16(&)[4]>(out), md, a.M, a.y, a.partials,
                                        reinterpret_cast<float*>(smem) + wave * (32 * 68), lane);
            } else {
                segmean_epilogue_dpp<OT>(out, md, a.M, a.y, a.partials, lane);
            }
        } else if (EPI == EPI_QUADBLEND) {
            // the LIIF area blend of the grid INR decoder (models/magnet_cnn_2d.py:285-290) in the epilogue: the four shifted predictions
            // of a (query, step) are four CONSECUTIVE rows (magnet_inr_grid_prepare_blend_f32 orders them so), i.e. four adjacent lanes
            // of the accumulator layout: weight each row, add across the quad with two DPP quad permutes, lane 0 of the quad stores -
            // the [4 M'][C] predictions are never written (1.34 GB per call at 256^2 queries) nor read back by a blend kernel
            const float wr = a.res[row];
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = out[t][i] * wr;
                    v += dpp_f<0xB1, 0xF>(v);                   // quad_perm [1, 0, 3, 2]
                    v += dpp_f<0x4E, 0xF>(v);                   // quad_perm [2, 3, 0, 1]
                    out[t][i] = v;
                }
            store_rows<OT, 0>(out, a.y, a.ldy, a.ncols, nullptr, 0, row >> 2, row_ok && (lane & 3) == 0, half);
        } else {
            float* yg = a.y + (a.y_group_stride ? (size_t)cg * a.y_group_stride : (size_t)cg * OT * 32);
            // y == nullptr: a tape-only launch (the training backward's recompute wants x0_out / hid[], not the chain's own output)
            if (EPI == EPI_STORE_MASK && a.partials) {
                // partials != NULL on a masked store: 256 floats (zeroed by the caller) that receive the block maxima of |y| - the
                // amax_in of magnet_linear_bwd128_f32 / 64 for the layer below (a maximum: the order of the atomics does not matter)
                float mx = 0.f;
                store_rows<OT, 2>(out, yg, a.ldy, a.ncols, a.res + cg * OT * 32, a.ldres, row, row_ok && a.y != nullptr, half, &mx);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(a.partials) + ((blockIdx.x * 4 + wave) & 255), __float_as_uint(mx));
            } else
            store_rows<OT, EPI == EPI_STORE_MASK ? 2 : (EPI == EPI_STORE_RES ? 1 : 0)>(out, yg, a.ldy, a.ncols, a.res + cg * OT * 32, a.ldres, row, row_ok && a.y != nullptr, half);
        }
        TRACE(32)
    }
    if ((TRACK_SAT || TRACK_OUT) && a.status) {
        // (NaN compares false: a non-finite activation that came from non-finite INPUTS is the input / output checks' business)
        if (__builtin_amdgcn_ballot_w64(sat >= 65504.0f) && (threadIdx.x & 63) == 0) atomicOr(a.status, a.status_bit);
    }
}

// ------------------------------------------------------------------------------------ dispatch
typedef void (*mlp_fn)(const MagnetMlpArgs);

#define MLP_CASE(NL_, HT_, OT_, ACT_, LA_, LN_, EPI_)                                              \
    if (nl == NL_ && ht == HT_ && ot == OT_ && act == ACT_ && last_act == LA_ && ln == LN_ && epi == EPI_) \
        fn = k_mlp_bf16<MODE, NL_, HT_, OT_, ACT_, LA_, LN_, EPI_, (NL_ >= 5 ? MLP_DIST5 : 1)>;
// the same with two chunks of lead, for launches of at most one workgroup per CU (MPNN node-level kernels of the small configs)
#define MLP_CASE_SMALL(NL_, HT_, OT_, ACT_, LA_, LN_, EPI_)                                        \
    if (small && nl == NL_ && ht == HT_ && ot == OT_ && act == ACT_ && last_act == LA_ && ln == LN_ && epi == EPI_) \
        fn = k_mlp_bf16<MODE, NL_, HT_, OT_, ACT_, LA_, LN_, EPI_, 2>;

template <int MODE>
static mlp_fn pick(int nl, int ht, int ot, int act, int last_act, int ln, int epi, bool small = false) {
    mlp_fn fn = nullptr;
    if constexpr (MODE != GEMM_FP16X3S) {                  // (the one-accumulator mode is instantiated for 5-Linear chains only ...)
    MLP_CASE(2, 4, 4, ACT_SWISH, true, false, EPI_STORE)
    MLP_CASE(2, 4, 4, ACT_SWISH, true, false, EPI_STORE_RES)
    MLP_CASE(1, 4, 4, ACT_NONE, false, false, EPI_STORE)
    // EDSR convolutions as gathered-tap GEMMs (conv + ReLU, conv + skip)
    MLP_CASE(1, 4, 4, ACT_RELU, true, false, EPI_STORE)
    MLP_CASE(1, 4, 4, ACT_NONE, false, false, EPI_STORE_RES)
    MLP_CASE(1, 2, 2, ACT_RELU, true, false, EPI_STORE)
    MLP_CASE(1, 2, 2, ACT_NONE, false, false, EPI_STORE_RES)
    MLP_CASE(1, 2, 2, ACT_NONE, false, false, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_NONE, false, false, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_RELU, true, false, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_NONE, false, false, EPI_STORE_RES)
    // backward-data GEMMs with the ReLU mask of the layer below as the epilogue (magnet_amd/autograd.py)
    MLP_CASE(1, 4, 4, ACT_NONE, false, false, EPI_STORE_MASK)
    MLP_CASE(1, 2, 2, ACT_NONE, false, false, EPI_STORE_MASK)
    MLP_CASE(1, 1, 1, ACT_NONE, false, false, EPI_STORE_MASK)
    // (... and there for the 64-wide ones: the 128-wide chains stay at two waves per SIMD either way and take the same time)
    MLP_CASE(5, 4, 4, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 4, 4, ACT_RELU, false, true, EPI_STORE_RES)
    MLP_CASE(5, 4, 4, ACT_RELU, false, true, EPI_SEGMEAN)
    MLP_CASE(5, 4, 1, ACT_RELU, false, false, EPI_STORE)
    }
    MLP_CASE(5, 2, 1, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 2, 1, ACT_RELU, false, true, EPI_STORE_RES)
    MLP_CASE(5, 2, 1, ACT_RELU, false, true, EPI_SEGMEAN)
    MLP_CASE(5, 2, 1, ACT_RELU, false, false, EPI_STORE)
    MLP_CASE(5, 2, 2, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 2, 4, ACT_RELU, false, true, EPI_STORE)
    MLP_CASE(5, 2, 4, ACT_RELU, false, true, EPI_QUADBLEND)
    if constexpr (MODE != GEMM_FP16X3S) {
    // single Linear + LayerNorm (+ residual / CSR mean / quad blend): the LAST layer of a chain with mlp_layers != 4, whose hidden layers run as
    // one-Linear launches (magnet_amd/ops.py:_mlp_by_layers; models/backbones/mlp.py:9-28 is generic in its depth)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_STORE)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_STORE_RES)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_SEGMEAN)
    MLP_CASE(1, 1, 1, ACT_NONE, false, true, EPI_STORE)
    MLP_CASE(1, 1, 1, ACT_NONE, false, true, EPI_STORE_RES)
    MLP_CASE(1, 1, 1, ACT_NONE, false, true, EPI_SEGMEAN)
    MLP_CASE(1, 2, 2, ACT_NONE, false, true, EPI_STORE)
    MLP_CASE(1, 4, 4, ACT_NONE, false, true, EPI_QUADBLEND)
    MLP_CASE_SMALL(2, 4, 4, ACT_SWISH, true, false, EPI_STORE)
    MLP_CASE_SMALL(2, 4, 4, ACT_SWISH, true, false, EPI_STORE_RES)
    MLP_CASE_SMALL(1, 4, 4, ACT_NONE, false, false, EPI_STORE)
    }
    return fn;
}

constexpr int SMALL_M = 128 * 256;      // rows of at most one workgroup per CU
static int launch_mlp(mlp_fn fn, const MagnetMlpArgs* args, void* stream) {
    if (!fn) return -3;
    const int tiles = (args->M + 31) / 32;
    hipLaunchKernelGGL(fn, dim3((tiles + 3) / 4, (args->ncolgroups > 1 && args->groups_in_block <= 1) ? args->ncolgroups : 1), dim3(256), 0, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------- MPNN per-node halves, weights resident
// A = W1a [h | node_in] + b1 and Bm = W1b [h | node_in] (the per-node factorisation of message_net_1, models/mpnn.py:73-79 /
// models/mpnn_2d.py:73-79) for one hop, fused with the InstanceNorm APPLY of the previous hop (GNN_Layer.forward, mpnn_2d.py:69:
// `x = self.norm(x, batch)`): the kernel reads the un-normalised update output, normalises it on load with the per-(graph, channel)
// mean / variance, and - besides A and Bm - writes the normalised h that the next update reads.  One launch instead of
// k_norm_apply + the generic one-Linear kernel, and no pass over h in between.
//
// The generic kernel streams weight chunks through an LDS ring with a barrier per k tile and two workgroups per CU; at 1 Linear
// that structure is all overhead (measured at 4 x 256^2 rows, scripts/bench_node_abl.py: 166 us of which 42 us is the empty
// skeleton, 44 us exposed source loads, 46 us the epilogue; 2.6 TB/s).  Here one workgroup of 8 waves per CU keeps ONE column
// group's planes (NKT x 16 KiB) in LDS for its lifetime, waves are independent (no barrier after the prologue), every wave walks
// 32-row tiles and the loads of its next tile are requested k tile by k tile as the current tile's registers are consumed.
// Workgroups b and b + 8 (same XCD, same L2) hold the two column groups and walk the same rows at the same time, so the second
// read of [h | node_in] is an L2 hit.  Same split arithmetic and accumulation order as k_mlp_bf16<GEMM_FP16X3, 1, 4, 4>.
struct NodeHalvesArgs {
    const float* h;            // [n][128] update output (un-normalised when mean != nullptr)
    const float* node_in;      // [n][kin]
    const float* mean;         // [n_graphs][128] or nullptr
    const float* var;          // [n_graphs][128] biased variance
    const f32x4* w[2];         // fp16 planes [NKT][4][2][2][64] x 16 B of the two column groups
    const float* b[2];         // padded biases
    float* y[2];               // A, Bm [n][128]
    float* h_norm;             // [n][128] or nullptr
    long n;
    int rows_per_graph, n_graphs, kin;
    float eps;
};
#define NH_MAX_GRAPHS 8
#ifndef NH_PIPE
#define NH_PIPE true
#endif
#ifndef NH_WAVES
#define NH_WAVES 4
#endif
template <int NKT>
__global__ void __launch_bounds__(NH_WAVES * 64, 1) k_node_halves(NodeHalvesArgs a) {
    constexpr int MODE = GEMM_FP16X3;
    MLP_ABL_INIT
    TRACE_INIT
    TRACE(40)
    __shared__ __attribute__((aligned(16))) u32x4 wl[NKT * 1024];
    __shared__ __attribute__((aligned(16))) float st[2][NH_MAX_GRAPHS][128];          // mean, 1 / sqrt(var + eps)
    __shared__ __attribute__((aligned(16))) float tls[NH_WAVES][32 * NH_TS];                 // per-wave transposition tile
    __shared__ __attribute__((aligned(16))) float bl[128];      // bias: re-read per tile from LDS - a global load here would be the
                                                                // youngest VMEM operation of the wave and waiting for it (in-order
                                                                // vmcnt) drains the prefetched rows AND the previous tile's stores
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    float* tl = tls[wave];
    // blockIdx -> (XCD, slot); slot parity = column group, so that the two groups of a row range share an L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot & 1;
    const int wi = (slot >> 1) * 8 + xcd, nwg = gridDim.x >> 1;
    {
        // all NKT * 2 loads of a thread in flight at once (a load -> LDS-write loop pays one memory round trip per iteration)
        const u32x4* g = reinterpret_cast<const u32x4*>(a.w[grp]);
        constexpr int NQ = NKT * 1024 / (NH_WAVES * 64);
        u32x4 wv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) wv[q] = g[q * (NH_WAVES * 64) + threadIdx.x];
#pragma unroll
        for (int q = 0; q < NQ; ++q) wl[q * (NH_WAVES * 64) + threadIdx.x] = wv[q];
        if (threadIdx.x < 128) bl[threadIdx.x] = a.b[grp][threadIdx.x];
        if (a.mean)
            for (int i = threadIdx.x; i < a.n_graphs * 128; i += NH_WAVES * 64) {
                st[0][i >> 7][i & 127] = a.mean[i];
                st[1][i >> 7][i & 127] = 1.0f / sqrtf(a.var[i] + a.eps);
            }
    }
    __syncthreads();
    TRACE(41)
    const bool norm = a.mean != nullptr;
    const bool write_h = norm && grp == 0 && a.h_norm;
    const long ntiles = (a.n + 31) / 32;
    const long stride = (long)nwg * NH_WAVES;
    long t = (long)wi * NH_WAVES + wave;
    if (t >= ntiles) return;
    // two tiles of raw rows in flight per wave (the wave is alone on its SIMD: nothing else covers the ~2 us of a load)
    f32x4 xa[NKT][4], xb[NKT][4];
    auto request = [&](f32x4 (&xr)[NKT][4], long tile, int kt) {
        const long r = min(min(tile, ntiles - 1) * 32 + j, a.n - 1);      // clamped: the last iterations prefetch rows they drop
        const float* p = kt < 4 ? a.h + r * 128 + kt * 32 + 4 * half : a.node_in + r * a.kin + (kt - 4) * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[kt][g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
    };
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) request(xa, t, kt);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) request(xb, t + stride, kt);
    const float* bias = bl;
    float* y = a.y[grp];
    auto do_tile = [&](f32x4 (&xr)[NKT][4], long tt) {
        const long row = tt * 32 + j;
        const int gid = norm ? (int)(min(row, a.n - 1) / a.rows_per_graph) : 0;
        f32x16 acc[4], accc[4];
        TRACE(42)
        init_bias<4>(acc, bias, half);
        zero_tiles<4>(accc);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x16 x;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] = xr[kt][g][c];
            if (kt < 4 && norm) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(&st[0][gid][kt * 32 + 8 * g + 4 * half]);
                    const f32x4 r = *reinterpret_cast<const f32x4*>(&st[1][gid][kt * 32 + 8 * g + 4 * half]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[4 * g + c] = (x[4 * g + c] - m[c]) * r[c];
                }
            }
#ifdef MAGNET_DEV_VARIANTS
            asm volatile("" ::"v"(x[0]), "v"(x[15]));
#endif
            TRACE(43)
            if (write_h && kt < 4 && !(abl & 4)) store_tile_lines(x, tl, a.h_norm, kt * 32, tt * 32, a.n, lane);
            if (!(abl & 1)) mma_chunk<4, MODE, NH_PIPE>(x, acc, accc, wl + kt * 1024, lane);
            TRACE(44)
            if (!(abl & 16)) request(xr, tt + 2 * stride, kt);            // into the registers this k tile has just released
        }
        fold_tiles<4, MODE>(acc, accc);
        TRACE(45)
        if (!(abl & 4)) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) store_tile_lines(acc[nt], tl, y, nt * 32, tt * 32, a.n, lane);
        } else if (acc[0][0] == 12345.678f) y[row] = acc[1][1] + acc[2][2] + acc[3][3];      // keep the values live
        TRACE(46)
    };
    for (; t < ntiles; t += 2 * stride) {
        do_tile(xa, t);
        if (t + stride < ntiles) do_tile(xb, t + stride);
    }
}

// A, Bm: [n_rows][128].  mean == NULL: h is used as it is and h_norm is not written.  Returns -2 for shapes this kernel does
// not cover (the caller then runs magnet_instance_norm_f32 + magnet_mlp_fp16x3_f32): kin not 32 or 64, more than 8 graphs,
// graph boundaries inside a 32-row tile.
extern "C" int magnet_mpnn_node_halves_f32(const float* h, const float* node_in, int kin, long n_rows, const float* mean,
                                           const float* var, float eps, int n_graphs, int rows_per_graph, const void* wa_f16,
                                           const float* ba, const void* wb_f16, const float* bb, float* A, float* Bm,
                                           float* h_norm, void* stream) {
    (void)hipGetLastError();
    if (n_rows <= 0) return 0;
    if (kin != 0 && kin != 32 && kin != 64) return -2;              // kin = 0: x = h alone (MAgNet InteractionNetwork's per-node halves)
    if (mean && (n_graphs > NH_MAX_GRAPHS || n_graphs < 1 || rows_per_graph % 32 || !var)) return -2;
    NodeHalvesArgs a;
    a.h = h, a.node_in = node_in, a.mean = mean, a.var = var;
    a.w[0] = reinterpret_cast<const f32x4*>(wa_f16), a.w[1] = reinterpret_cast<const f32x4*>(wb_f16);
    a.b[0] = ba, a.b[1] = bb, a.y[0] = A, a.y[1] = Bm, a.h_norm = h_norm;
    a.n = n_rows, a.rows_per_graph = rows_per_graph, a.n_graphs = n_graphs, a.kin = kin, a.eps = eps;
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const long ntiles = (n_rows + 31) / 32;
    long pairs = (ntiles + NH_WAVES - 1) / NH_WAVES;                 // workgroup pairs that have at least one tile per wave
    int grid = (int)(pairs * 2 < ncu ? pairs * 2 : ncu);
    grid = (grid + 15) / 16 * 16;                                    // whole (XCD, slot pair) blocks
    hipStream_t s = (hipStream_t)stream;
    if (kin == 0) hipLaunchKernelGGL(k_node_halves<4>, dim3(grid), dim3(NH_WAVES * 64), 0, s, a);
    else if (kin == 32) hipLaunchKernelGGL(k_node_halves<5>, dim3(grid), dim3(NH_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(k_node_halves<6>, dim3(grid), dim3(NH_WAVES * 64), 0, s, a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------- MPNN per-node halves, two waves per SIMD (round 5)
// k_node_halves above runs ONE wave per SIMD (two accumulator sets + two tiles of prefetched rows: ~400 registers), so every wait of
// that wave - the LDS round trip of a weight fragment, the tail of a row load, the transposition tile of the stores - is paid in full
// (DESIGN.md 10 rule 9): 103 us for 0.44 GB, 4.2 TB/s.  This variant is k_node_update2's recipe applied to it: EIGHT waves per
// workgroup (two per SIMD, <= 256 registers) share the same resident planes - ONE accumulator per output block (GEMM_FP16X3S:
// planes pre-scaled by 2^11, |w| < 31.98 - the caller keeps k_node_halves otherwise) and one tile of rows per wave, refilled k tile by
// k tile for the wave's next tile as the registers are consumed.  Same pack layout, same arithmetic as the one-accumulator chains.
template <int NKT>
__global__ void __launch_bounds__(512, 1) k_node_halves2(NodeHalvesArgs a) {
    constexpr int MODE = GEMM_FP16X3S;
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) u32x4 wl[NKT * 1024];
    __shared__ __attribute__((aligned(16))) float st[2][NH_MAX_GRAPHS][128];          // mean, 1 / sqrt(var + eps)
    __shared__ __attribute__((aligned(16))) float tls[NW][32 * NH_TS];                // per-wave transposition tile
    __shared__ __attribute__((aligned(16))) float bl[128];                            // 2^11 bias (as the pack holds it)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    float* tl = tls[wave];
    // blockIdx -> (XCD, slot); slot parity = column group, so that the two groups of a row range share an L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot & 1;
    const int wi = (slot >> 1) * 8 + xcd, nwg = gridDim.x >> 1;
    {
        const u32x4* g = reinterpret_cast<const u32x4*>(a.w[grp]);
        constexpr int NQ = NKT * 1024 / (NW * 64);
        u32x4 wv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) wv[q] = g[q * (NW * 64) + threadIdx.x];
#pragma unroll
        for (int q = 0; q < NQ; ++q) wl[q * (NW * 64) + threadIdx.x] = wv[q];
        if (threadIdx.x < 128) bl[threadIdx.x] = a.b[grp][threadIdx.x];
        if (a.mean)
            for (int i = threadIdx.x; i < a.n_graphs * 128; i += NW * 64) {
                st[0][i >> 7][i & 127] = a.mean[i];
                st[1][i >> 7][i & 127] = 1.0f / sqrtf(a.var[i] + a.eps);
            }
    }
    __syncthreads();                                   // (weights, bias and statistics in place before any wave's first tile)
    const bool norm = a.mean != nullptr;
    const bool write_h = norm && grp == 0 && a.h_norm;
    const long ntiles = (a.n + 31) / 32;
    const long stride = (long)nwg * NW;
    long t = (long)wi * NW + wave;
    if (t >= ntiles) return;
    f32x4 xa[NKT][4];
    // two base pointers per tile (this lane's row of h / node_in), every load at a compile-time offset from one of them; the row index is
    // made opaque per tile (left alone hipcc turns each of the 20 load addresses into an induction variable of its own and spills)
    struct Rows { const float* h; const float* nin; };
    auto rows_of = [&](long tile) {
        long r = min(min(tile, ntiles - 1) * 32 + j, a.n - 1);            // clamped: the last iterations prefetch rows they drop
        asm volatile("" : "+v"(r));
        Rows p;
        p.h = a.h + r * 128 + 4 * half;
        p.nin = a.node_in + r * a.kin + 4 * half;
        return p;
    };
    auto request = [&](const Rows& rw, int kt) {
        const float* p = kt < 4 ? rw.h + kt * 32 : rw.nin + (kt - 4) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) xa[kt][g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
    };
    {
        const Rows first = rows_of(t);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) request(first, kt);
    }
    float* y = a.y[grp];
    for (; t < ntiles; t += stride) {
        const Rows nxt = rows_of(t + stride);
        const long row = t * 32 + j;
        const int gid = norm ? (int)(min(row, a.n - 1) / a.rows_per_graph) : 0;
        f32x16 acc[4], accc[4];
        init_bias<4>(acc, bl, half);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x16 x;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] = xa[kt][g][c];
            if (kt < 4 && norm) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(&st[0][gid][kt * 32 + 8 * g + 4 * half]);
                    const f32x4 r = *reinterpret_cast<const f32x4*>(&st[1][gid][kt * 32 + 8 * g + 4 * half]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[4 * g + c] = (x[4 * g + c] - m[c]) * r[c];
                }
            }
            request(nxt, kt);                          // into the registers this k tile has just released
            if (write_h && kt < 4) store_tile_lines(x, tl, a.h_norm, kt * 32, t * 32, a.n, lane);
            mma_chunk<4, MODE, false>(x, acc, accc, wl + kt * 1024, lane);
        }
        fold_tiles<4, MODE>(acc, accc);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) store_tile_lines(acc[nt], tl, y, nt * 32, t * 32, a.n, lane);
    }
}

// As magnet_mpnn_node_halves_f32 with the packs of the one-accumulator mode (packing mode "fp16x3s": planes and biases times 2^11,
// |w| < 31.98).
extern "C" int magnet_mpnn_node_halves2_f32(const float* h, const float* node_in, int kin, long n_rows, const float* mean,
                                            const float* var, float eps, int n_graphs, int rows_per_graph, const void* wa_f16s,
                                            const float* ba_s, const void* wb_f16s, const float* bb_s, float* A, float* Bm,
                                            float* h_norm, void* stream) {
    (void)hipGetLastError();
    if (n_rows <= 0) return 0;
    if (kin != 0 && kin != 32 && kin != 64) return -2;
    if (mean && (n_graphs > NH_MAX_GRAPHS || n_graphs < 1 || rows_per_graph % 32 || !var)) return -2;
    NodeHalvesArgs a;
    a.h = h, a.node_in = node_in, a.mean = mean, a.var = var;
    a.w[0] = reinterpret_cast<const f32x4*>(wa_f16s), a.w[1] = reinterpret_cast<const f32x4*>(wb_f16s);
    a.b[0] = ba_s, a.b[1] = bb_s, a.y[0] = A, a.y[1] = Bm, a.h_norm = h_norm;
    a.n = n_rows, a.rows_per_graph = rows_per_graph, a.n_graphs = n_graphs, a.kin = kin, a.eps = eps;
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const long ntiles = (n_rows + 31) / 32;
    long pairs = (ntiles + 7) / 8;                                   // workgroup pairs that have at least one tile per wave
    int grid = (int)(pairs * 2 < ncu ? pairs * 2 : ncu);
    grid = (grid + 15) / 16 * 16;                                    // whole (XCD, slot pair) blocks
    hipStream_t s = (hipStream_t)stream;
    if (kin == 0) hipLaunchKernelGGL(k_node_halves2<4>, dim3(grid), dim3(512), 0, s, a);
    else if (kin == 32) hipLaunchKernelGGL(k_node_halves2<5>, dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(k_node_halves2<6>, dim3(grid), dim3(512), 0, s, a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------- MPNN node update, persistent
// g = h + Swish(W2 Swish(W1 [h | agg | node_in] + b1) + b2)   (GNN_Layer.update, models/mpnn.py:81-90 / mpnn_2d.py:81-91, residual
// branch in_features == out_features) for the large-graph inference path.  Same lessons as k_node_halves, for a chain whose
// planes (147 + 64 KiB) do not fit in LDS together:
//   * one workgroup of 4 waves per CU, persistent over 128-row tiles; W2 (64 KiB) resident; the W1 chunks (16 KiB per k tile) pass
//     through a two-slot LDS buffer, fetched by plain loads into registers D chunks ahead and written to LDS one step before use -
//     compiler-tracked waits, one barrier per first-layer step, and the chunk stream runs on across tile boundaries;
//   * every wave keeps one whole tile of source rows in flight ahead of itself; the first layer takes the agg / node_in k tiles first
//     and the h k tiles last, so the h registers are still intact at the epilogue (they ARE the residual) and are re-requested for
//     the next tile only then, five steps before their next use;
//   * no global load is consumed soon after it is issued (vmcnt retires in order: such a load would wait for every store of the
//     previous tile): biases come from LDS;
//   * results leave through a wave-private LDS tile as whole 128-byte lines.
struct NodeUpdateArgs {
    const float* h;            // [n][128] (normalised) node features: input and residual
    const float* agg;          // [n][128] mean-aggregated messages
    const float* node_in;      // [n][kin]
    const f32x4* w1;           // fp16 planes of update_net_1 [NKT][4][2][2][64] x 16 B, columns [h | agg | node_in]
    const f32x4* w2;           // fp16 planes of update_net_2 [4][4][2][2][64] x 16 B
    const float* b1;
    const float* b2;
    float* y;                  // [n][128]
    const float* mean;         // [n_graphs][128] or nullptr: h is then normalised on load, (h - mean) / sqrt(var + eps)
    const float* var;
    float* part_sum;           // [n / 128][128] or nullptr: per 128-row tile the column sums of y ...
    float* part_m2;            // ... and the sums of squares centred on the tile's own column means (for k_colstat_final)
    long n;
    int kin, rows_per_graph, n_graphs;
    float eps;
};
template <int NKT, bool NORM>
__global__ void __launch_bounds__(256, 1) k_node_update(NodeUpdateArgs a) {
    constexpr int MODE = GEMM_FP16X3;
    constexpr int D = (NKT % 3 == 0) ? 3 : 2;                        // W1 chunks in flight in registers (NKT % D == 0: static ring index)
    TRACE_INIT
    TRACE(50)
    static_assert(NKT % D == 0, "ring depth");
    __shared__ __attribute__((aligned(16))) u32x4 w2l[4 * 1024];
    __shared__ __attribute__((aligned(16))) u32x4 w1s[2][1024];
    __shared__ __attribute__((aligned(16))) float tls[4][32 * NH_TS];
    __shared__ __attribute__((aligned(16))) float bl[2][128];
    __shared__ __attribute__((aligned(16))) float st[2][NH_MAX_GRAPHS][128];          // mean, 1 / sqrt(var + eps)
    __shared__ __attribute__((aligned(16))) float ws[4][2][128];                      // per wave: column sums / centred squares of a tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    float* tl = tls[wave];
    constexpr bool norm = NORM;
    if (norm)
        for (int i = threadIdx.x; i < a.n_graphs * 128; i += 256) {
            st[0][i >> 7][i & 127] = a.mean[i];
            st[1][i >> 7][i & 127] = 1.0f / sqrtf(a.var[i] + a.eps);
        }
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.w1);
    // first-layer step i takes k tile KSEQ(i): agg and node_in first, h last
    auto kseq = [](int i) { return i < NKT - 4 ? i + 4 : i - (NKT - 4); };
    {
        const u32x4* g2 = reinterpret_cast<const u32x4*>(a.w2);
        u32x4 wv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) wv[q] = g2[q * 256 + threadIdx.x];
#pragma unroll
        for (int q = 0; q < 16; ++q) w2l[q * 256 + threadIdx.x] = wv[q];
        if (threadIdx.x < 128) bl[0][threadIdx.x] = a.b1[threadIdx.x];
        else bl[1][threadIdx.x - 128] = a.b2[threadIdx.x - 128];
    }
    const long ntiles = (a.n + 127) / 128;
    long tile = blockIdx.x;
    if (tile >= ntiles) return;                                       // (whole workgroup: uniform)
    // W1 chunk stream: wr[r] holds this thread's 4 pieces of a chunk on its way to LDS
    u32x4 wr[D][4];
    auto wload = [&](int r, int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[r][q] = g1[kt * 1024 + q * 256 + threadIdx.x];
    };
    auto wstore = [&](int r, int slot) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w1s[slot][q * 256 + threadIdx.x] = wr[r][q];
    };
    // prologue: chunk of step 0 straight to slot 0, chunks of steps 1 .. D in flight
    wload(0, kseq(0));
    wstore(0, 0);
#pragma unroll
    for (int i = 1; i <= D; ++i) wload(i % D, kseq(i % NKT));
    f32x4 xr[NKT][4];
    auto request = [&](long tl_, int kt) {
        const long r = min(min(tl_, ntiles - 1) * 128 + wave * 32 + j, a.n - 1);   // clamped: rows past the end are dropped later
        const float* p = kt < 4 ? a.h + r * 128 + kt * 32 + 4 * half
                       : kt < 8 ? a.agg + r * 128 + (kt - 4) * 32 + 4 * half
                                : a.node_in + r * a.kin + (kt - 8) * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[kt][g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
    };
#pragma unroll
    for (int i = 0; i < NKT; ++i) request(tile, kseq(i));
    int par = 0;                                                      // LDS slot of the chunk of first-layer step 0 of this tile
    // bl[0] (written by waves 0 and 1) is read by EVERY wave's init_bias below, before the first barrier of the tile loop: without this
    // one a wave that got ahead read what the previous kernel had left at that LDS address - a 32-row block of a launch came out with a
    // neighbouring hop's bias now and then (scripts/stress_precision_leg.py).  It drains the same loads the first
    // barrier of the loop would drain: no extra wait.
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * 128 + wave * 32;
        const int gid = norm ? (int)(min(row0 + j, a.n - 1) / a.rows_per_graph) : 0;
        f32x16 acc[4], accc[4];
        init_bias<4>(acc, bl[0], half);
        zero_tiles<4>(accc);
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int kt = kseq(i);
            const int slot = (par + i) & 1;
            TRACE(51)
            __syncthreads();                                          // chunk of step i complete in `slot`; the other slot is free
            TRACE(52)
            // chunk of step i+1 (this tile's or, at the end, the next tile's first) -> the free slot; then its ring place is
            // refilled with the chunk of step i+1+D
            wstore((i + 1) % D, slot ^ 1);
            wload((i + 1) % D, kseq((i + 1 + D) % NKT));
            f32x16 x;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] = xr[kt][g][c];
            if (kt < 4 && norm) {                                      // normalise on load; the residual is the normalised h too
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(&st[0][gid][kt * 32 + 8 * g + 4 * half]);
                    const f32x4 r = *reinterpret_cast<const f32x4*>(&st[1][gid][kt * 32 + 8 * g + 4 * half]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        x[4 * g + c] = (x[4 * g + c] - m[c]) * r[c];
                        xr[kt][g][c] = x[4 * g + c];
                    }
                }
            }
#ifdef MAGNET_DEV_VARIANTS
            asm volatile("" ::"v"(x[0]), "v"(x[15]));
#endif
            TRACE(53)
            mma_chunk<4, MODE, false>(x, acc, accc, w1s[slot], lane);
            TRACE(54)
            if (kt >= 4) request(tile + gridDim.x, kt);                // h tiles stay: they are the residual
        }
        par = (par + NKT) & 1;
        fold_tiles<4, MODE>(acc, accc);
        TRACE(55)
        f32x16 h0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            apply_act<ACT_SWISH>(acc[t]);
            h0[t] = acc[t];
        }
        init_bias<4>(acc, bl[1], half);
        zero_tiles<4>(accc);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mma_chunk<4, MODE, false>(h0[kt], acc, accc, w2l + kt * 1024, lane);
        fold_tiles<4, MODE>(acc, accc);
        TRACE(56)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            apply_act<ACT_SWISH>(acc[t]);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][4 * g + c] += xr[t][g][c];
            store_tile_lines(acc[t], tl, a.y, t * 32, row0, a.n, lane);
            if (a.part_sum) {
                // InstanceNorm statistics of the output without another pass over it: the tile is still in the wave's LDS tile; lane
                // (column j, half) walks 16 of its 32 rows, the halves are combined, the squares are centred on the wave's own column
                // mean (the pairwise scheme of k_colstat_partial, 32-row chunks)
                float vcol[16], sm = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    vcol[r] = tl[(16 * half + r) * NH_TS + j];
                    sm += vcol[r];
                }
                sm += __shfl_xor(sm, 32);
                const float mu = sm * (1.0f / 32.0f);
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) q = fmaf(vcol[r] - mu, vcol[r] - mu, q);
                q += __shfl_xor(q, 32);
                if (half == 0) {
                    ws[wave][0][t * 32 + j] = sm;
                    ws[wave][1][t * 32 + j] = q;
                }
            }
        }
        if (a.part_sum) {
            __syncthreads();
            if (wave == 0) {                             // merge the four 32-row chunks of this 128-row tile (fixed order)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int col = lane + 64 * cc;
                    float s4[4], tot = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        s4[w] = ws[w][0][col];
                        tot += s4[w];
                    }
                    const float mu = tot * (1.0f / 128.0f);
                    float m2 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float d = s4[w] * (1.0f / 32.0f) - mu;
                        m2 += ws[w][1][col] + 32.0f * d * d;
                    }
                    a.part_sum[tile * 128 + col] = tot;
                    a.part_m2[tile * 128 + col] = m2;
                }
            }
        }
        TRACE(57)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) request(tile + gridDim.x, kt);
    }
}

// ---------------------------------------------------------------------------- MPNN node update, two workgroups per CU (round 4)
// k_node_update above is bound by neither roof (160 us at 4 x 256^2 rows: 0.19 of the split's MFMA ceiling, 0.43 of the copy ceiling):
// ONE wave per SIMD issues ~10 k cycles of MFMAs and ~7 k cycles of VALU work per 128-row tile and spends the other 60 % of a tile's
// 42 k cycles behind its 13 barriers and its own loads, with nobody else on the SIMD to fill the gaps.  This variant makes room for a
// SECOND workgroup per CU (the two drift against each other: matrix time of one beside the memory time of the other):
//   * LDS per workgroup 64 KiB -> the planes of update_net_2 are streamed through the chunk slots like those of update_net_1 (13 or 14
//     chunks of 16 KiB per tile from L2) instead of living in LDS (64 KiB);
//   * <= 256 registers per wave -> ONE accumulator per output block (GEMM_FP16X3S: planes pre-scaled by 2^11, |w| < 31.98 - the
//     caller falls back to k_node_update otherwise), source rows three k tiles ahead in a ring of three register sets instead of a
//     whole tile ahead, and the residual h is read again at the end of the tile (an L2 hit: the same rows were loaded ~9 k tile
//     steps earlier) instead of being held in 64 registers.
// Same arithmetic as the one-accumulator chains (Split<GEMM_FP16X3S>), same output layout, same optional statistics partials.
template <int NKT, bool NORM>
__global__ void __launch_bounds__(256, 2) k_node_update2(NodeUpdateArgs a) {
    constexpr int MODE = GEMM_FP16X3S;
    constexpr int NS = NKT + 4;                                       // chunk steps per tile: update_net_1's k tiles, then update_net_2's
    __shared__ __attribute__((aligned(16))) u32x4 wsl[2][1024];
    __shared__ __attribute__((aligned(16))) float tls[4][32 * NH_TS];
    __shared__ __attribute__((aligned(16))) float bl[2][128];
    __shared__ __attribute__((aligned(16))) float st[2][NH_MAX_GRAPHS][128];          // mean, 1 / sqrt(var + eps)
    __shared__ __attribute__((aligned(16))) float ws[4][2][128];                      // per wave: column sums / centred squares of a tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    float* tl = tls[wave];
    if (NORM)
        for (int i = threadIdx.x; i < a.n_graphs * 128; i += 256) {
            st[0][i >> 7][i & 127] = a.mean[i];
            st[1][i >> 7][i & 127] = 1.0f / sqrtf(a.var[i] + a.eps);
        }
    if (threadIdx.x < 128) bl[0][threadIdx.x] = a.b1[threadIdx.x];
    else bl[1][threadIdx.x - 128] = a.b2[threadIdx.x - 128];
    const u32x4* g1 = reinterpret_cast<const u32x4*>(a.w1);
    const u32x4* g2 = reinterpret_cast<const u32x4*>(a.w2);
    // first-layer step i takes k tile kseq(i): agg and node_in first, h last (as k_node_update: same accumulation order)
    auto kseq = [](int i) { return i < NKT - 4 ? i + 4 : i - (NKT - 4); };
    auto chunk = [&](int step) { return step < NKT ? g1 + kseq(step) * 1024 : g2 + (step - NKT) * 1024; };
    const long ntiles = (a.n + 127) / 128;
    long tile = blockIdx.x;
    if (tile >= ntiles) return;                                       // (whole workgroup: uniform)
    // chunk stream: this thread's four pieces of the NEXT step's chunk wait in registers, are written to the free slot right after the
    // barrier of the current step, and the step after that is requested.  (Tried instead: LDS-DMA two steps ahead into a ring of three
    // slots with counted vmcnt waits - 135 us against 128 us on the 4 x 256^2 graph: the chunk stream is not what the kernel waits for.)
    u32x4 wr[4];
    // (uniform chunk base kept opaque in SGPRs + one 32-bit per-thread offset: written with 64-bit per-thread pointers hipcc keeps all 52
    // chunk addresses of a tile alive as loop invariants - 104 registers - and spills them)
    const unsigned voff = threadIdx.x * 16u;
    auto wload = [&](int step) {
        const char* c = reinterpret_cast<const char*>(chunk(step));
        asm volatile("" : "+s"(c));
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[q] = *reinterpret_cast<const u32x4*>(c + (voff + q * 4096u));
    };
    auto wstore = [&](int slot) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wsl[slot][q * 256 + threadIdx.x] = wr[q];
    };
    wload(0);
    wstore(0);
    wload(1);
    int par = 0;                                                      // LDS slot of the chunk of step 0 of this tile
    // source rows: a ring of three sets, step i of a tile reads set i % 3 (NKT is 9 or 10: the ring position is kept across tiles by
    // taking the next tile's first steps in the order the ring leaves them)
    f32x4 xr[3][4];
    // three base pointers per tile (this lane's row of h / agg / node_in), every load at a compile-time offset from one of them.  The row
    // index is made opaque per tile: left alone hipcc turns each of the ~50 distinct load addresses into an induction variable of its
    // own and spills them (106 registers).
    struct Rows { const float* h; const float* agg; const float* nin; };
    auto rows_of = [&](long tl_) {
        long r = min(min(tl_, ntiles - 1) * 128 + wave * 32 + j, a.n - 1);         // clamped: rows past the end are dropped later
        asm volatile("" : "+v"(r));
        Rows p;
        p.h = a.h + r * 128 + 4 * half;
        p.agg = a.agg + r * 128 + 4 * half;
        p.nin = a.node_in + r * a.kin + 4 * half;
        return p;
    };
    auto request = [&](int set, const Rows& rw, int kt) {
        const float* p = kt < 4 ? rw.h + kt * 32 : kt < 8 ? rw.agg + (kt - 4) * 32 : rw.nin + (kt - 8) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[set][g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
    };
    static_assert(NKT % 3 == 0, "row sets: step i <-> set i % 3 in every tile");
    auto body = [&]() __attribute__((always_inline)) {
        constexpr int ROT = 0;
        const long row0 = tile * 128 + wave * 32;
        const int gid = NORM ? (int)(min(row0 + j, a.n - 1) / a.rows_per_graph) : 0;
        const Rows cur = rows_of(tile), nxt = rows_of(tile + gridDim.x);
        f32x16 acc[4], accc[4];
        init_bias<4>(acc, bl[0], half);
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int kt = kseq(i);
            const int slot = (par + i) & 1;
            __syncthreads();                                          // chunk of step i complete in `slot`; the other slot is free
            wstore(slot ^ 1);                                         // chunk of step i + 1
            wload((i + 2) % NS);                                      // chunk of step i + 2 (the next tile's first ones at the end)
            f32x16 x;
            const int set = (i + ROT) % 3;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] = xr[set][g][c];
            if (kt < 4 && NORM) {                                      // normalise on load
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(&st[0][gid][kt * 32 + 8 * g + 4 * half]);
                    const f32x4 r = *reinterpret_cast<const f32x4*>(&st[1][gid][kt * 32 + 8 * g + 4 * half]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[4 * g + c] = (x[4 * g + c] - m[c]) * r[c];
                }
            }
            // the set is free: rows of step i + 3 (this tile's, or the next tile's step i + 3 - NKT)
            if (i + 3 < NKT) request(set, cur, kseq(i + 3));
            else request(set, nxt, kseq(i + 3 - NKT));
            mma_chunk<4, MODE, false>(x, acc, accc, wsl[slot], lane);
        }
        fold_tiles<4, MODE>(acc, accc);
        f32x16 h0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            apply_act<ACT_SWISH>(acc[t]);
            h0[t] = acc[t];
        }
        init_bias<4>(acc, bl[1], half);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int i = NKT + kt;
            const int slot = (par + i) & 1;
            __syncthreads();
            wstore(slot ^ 1);
            wload((i + 2) % NS);
            mma_chunk<4, MODE, false>(h0[kt], acc, accc, wsl[slot], lane);
        }
        fold_tiles<4, MODE>(acc, accc);
        // residual: the (normalised) h rows again - requested before the second Swish, they come out of L2.  (The barrier keeps hipcc from
        // hoisting these 16 loads above the second layer, where h0 is still alive: 64 registers too many.)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 hres[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) hres[t][g] = *reinterpret_cast<const f32x4*>(cur.h + t * 32 + 8 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            apply_act<ACT_SWISH>(acc[t]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 hv = hres[t][g];
                if (NORM) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(&st[0][gid][t * 32 + 8 * g + 4 * half]);
                    const f32x4 r = *reinterpret_cast<const f32x4*>(&st[1][gid][t * 32 + 8 * g + 4 * half]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) hv[c] = (hv[c] - m[c]) * r[c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][4 * g + c] += hv[c];
            }
            store_tile_lines(acc[t], tl, a.y, t * 32, row0, a.n, lane);
            if (a.part_sum) {                                    // as k_node_update: the tile is still in the wave's LDS tile
                float vcol[16], sm = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    vcol[r] = tl[(16 * half + r) * NH_TS + j];
                    sm += vcol[r];
                }
                sm += __shfl_xor(sm, 32);
                const float mu = sm * (1.0f / 32.0f);
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) q = fmaf(vcol[r] - mu, vcol[r] - mu, q);
                q += __shfl_xor(q, 32);
                if (half == 0) {
                    ws[wave][0][t * 32 + j] = sm;
                    ws[wave][1][t * 32 + j] = q;
                }
            }
        }
        if (a.part_sum) {
            __syncthreads();
            if (wave == 0) {                                     // merge the four 32-row chunks of this 128-row tile (fixed order)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int col = lane + 64 * cc;
                    float s4[4], tot = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        s4[w] = ws[w][0][col];
                        tot += s4[w];
                    }
                    const float mu = tot * (1.0f / 128.0f);
                    float m2 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float d = s4[w] * (1.0f / 32.0f) - mu;
                        m2 += ws[w][1][col] + 32.0f * d * d;
                    }
                    a.part_sum[tile * 128 + col] = tot;
                    a.part_m2[tile * 128 + col] = m2;
                }
            }
        }
        par = (par + NS) & 1;
    };
    // prologue: rows of steps 0 .. 2 of the first tile
#pragma unroll
    for (int i = 0; i < 3; ++i) request(i, rows_of(tile), kseq(i));
    // bl[0] (written by waves 0 and 1) is read by every wave's init_bias at the top of body(), BEFORE the tile's first barrier: this
    // barrier was missing (round 4) and a wave that got ahead of wave 0 / 1 started its first tile from whatever the previous kernel
    // had left at that LDS address - one 32-row block of ~1 in 100 launches off by ~1e-2 (found by scripts/stress_precision_leg.py +
    // scripts/dev/find_flaky_kernel.py).  It drains the loads the loop's first barrier would drain anyway.
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) body();
}

// The same update for launches of at most one 128-row tile per CU (the launch-bound 1-D configs: 100 rows = one workgroup): the
// persistent kernel's W1 chunk stream is nine dependent L2 -> register -> LDS hand-overs with a barrier each - 16.8 us for 100 rows, a
// third of a C1 hop.  Here nothing is streamed: ALL of W1 (9 x 16 KiB, kin = 32) is requested in one burst of LDS-DMA loads, the tile's
// source rows and W2 (64 registers per lane) in the same breath; one wait, one barrier, the first layer runs without another; W2 then
// takes W1's place in LDS (one more barrier) and the store tiles the space behind it.  Same accumulation order as k_node_update (agg /
// node_in k tiles first, h last): bitwise the same results.  No InstanceNorm on load, no statistics (the small-graph path has neither).
// NORM_OUT (a launch of ONE workgroup, n <= 128 rows: BASELINE.json configs[0], 2 x 50 nodes): the InstanceNorm of the hop
// (GNN_Layer.forward, models/mpnn.py:69-70: x = self.norm(update, batch)) applied to the update's output before it leaves - the tile
// goes to LDS, every (graph, column) gets its mean and centred second moment in two passes over at most 128 LDS rows, and what is
// stored is the normalised h of the next hop: one launch per hop less in a rollout that is bound by launches (C1: 5 -> 4 per hop).
template <bool NORM_OUT>
__global__ void __launch_bounds__(256, 1) k_node_update_burst(NodeUpdateArgs a) {
    constexpr int MODE = GEMM_FP16X3, NKT = 9;
    __shared__ __attribute__((aligned(16))) u32x4 wb[NKT * 1024];                     // W1 whole; later W2 in [0, 4096) + the store tiles
    __shared__ __attribute__((aligned(16))) float bl[2][128];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const long row0 = (long)blockIdx.x * 128 + wave * 32;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) dma_chunk<16>(a.w1 + kt * 1024, wb + kt * 1024, wave, lane);
    f32x4 xr[NKT][4];
    {
        const long r = min(row0 + j, a.n - 1);                                        // clamped: rows past the end are dropped at the store
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const float* p = kt < 4 ? a.h + r * 128 + kt * 32 + 4 * half
                           : kt < 8 ? a.agg + r * 128 + (kt - 4) * 32 + 4 * half
                                    : a.node_in + r * a.kin + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) xr[kt][g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
        }
    }
    u32x4 w2r[16];
    {
        const u32x4* g2 = reinterpret_cast<const u32x4*>(a.w2);
#pragma unroll
        for (int q = 0; q < 16; ++q) w2r[q] = g2[q * 256 + threadIdx.x];
    }
    if (threadIdx.x < 128) bl[0][threadIdx.x] = a.b1[threadIdx.x];
    else bl[1][threadIdx.x - 128] = a.b2[threadIdx.x - 128];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  // the DMA'd chunks (the compiler does not track them)
    __syncthreads();
    f32x16 acc[4], accc[4];
    init_bias<4>(acc, bl[0], half);
    zero_tiles<4>(accc);
#pragma unroll
    for (int i = 0; i < NKT; ++i) {
        const int kt = i < NKT - 4 ? i + 4 : i - (NKT - 4);                           // agg and node_in first, h last (as k_node_update)
        f32x16 x;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) x[4 * g + c] = xr[kt][g][c];
        mma_chunk<4, MODE, false>(x, acc, accc, wb + kt * 1024, lane);
    }
    fold_tiles<4, MODE>(acc, accc);
    f32x16 h0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        apply_act<ACT_SWISH>(acc[t]);
        h0[t] = acc[t];
    }
    __syncthreads();                                                                  // everybody is done with W1
#pragma unroll
    for (int q = 0; q < 16; ++q) wb[q * 256 + threadIdx.x] = w2r[q];
    __syncthreads();
    init_bias<4>(acc, bl[1], half);
    zero_tiles<4>(accc);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) mma_chunk<4, MODE, false>(h0[kt], acc, accc, wb + kt * 1024, lane);
    fold_tiles<4, MODE>(acc, accc);
    float* tl = reinterpret_cast<float*>(wb + 4096) + wave * 32 * NH_TS;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        apply_act<ACT_SWISH>(acc[t]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][4 * g + c] += xr[t][g][c];
        if constexpr (!NORM_OUT) store_tile_lines(acc[t], tl, a.y, t * 32, row0, a.n, lane);
    }
    if constexpr (NORM_OUT) {
        constexpr int TS = 132;                                                       // 128 x 132 floats = 66 KiB of the 80 KiB behind W2
        float* yt = reinterpret_cast<float*>(wb + 4096);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) yt[(wave * 32 + j) * TS + t * 32 + 8 * (i >> 2) + 4 * half + (i & 3)] = acc[t][i];
        __syncthreads();
        const int c = threadIdx.x & 127, n = (int)a.n, rpg = a.rows_per_graph;
        for (int r0 = (int)(threadIdx.x >> 7) * rpg; r0 < n; r0 += 2 * rpg) {         // graphs alternate between the two thread halves
            const int r1 = min(r0 + rpg, n);
            const float cnt = (float)(r1 - r0);
            // eight independent partial sums per pass: a single running sum would wait one LDS latency per row (measured: the fused
            // launch was then SLOWER than the two it replaces)
            const float* col = yt + c;
            float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int r = r0;
            for (; r + 8 <= r1; r += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) p[q] += col[(r + q) * TS];
            }
            for (; r < r1; ++r) p[0] += col[r * TS];
            const float m = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) / cnt;
#pragma unroll
            for (int q = 0; q < 8; ++q) p[q] = 0.f;
            for (r = r0; r + 8 <= r1; r += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float d = col[(r + q) * TS] - m;
                    p[q] += d * d;
                }
            }
            for (; r < r1; ++r) {
                const float d = col[r * TS] - m;
                p[0] += d * d;
            }
            const float m2 = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            const float inv = 1.0f / sqrtf(m2 / cnt + a.eps);
            float* out = a.y + c;
            for (r = r0; r + 8 <= r1; r += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) out[(size_t)(r + q) * 128] = (col[(r + q) * TS] - m) * inv;
            }
            for (; r < r1; ++r) out[(size_t)r * 128] = (col[r * TS] - m) * inv;
        }
    }
}

// y [n_rows][128] = hn + Swish(W2 Swish(W1 [hn | agg | node_in] + b1) + b2), hn = mean ? (h - mean[g]) / sqrt(var[g] + eps) : h;
// w1 / w2: fp16x3 planes as magnet_mlp_fp16x3_f32 takes them.  Returns -2 for kin not in {32, 64} or more than 8 graphs (the
// caller then uses the generic chain).  y must not alias h.  part_sum / part_m2 ([n_rows / 128][128] each, n_rows % 128 == 0): per
// 128-row tile the column sums of y and the squares centred on the tile's column means - input of magnet_colstat_final_f32.
// planes_scaled != 0: w1 / w2 / b1 / b2 are the ONE-accumulator packs (packing mode "fp16x3s": planes and biases carry 2^11) and the
// launch is k_node_update2 (two workgroups per CU); 0: the two-accumulator packs and k_node_update, as before.
extern "C" int magnet_mpnn_node_update2_f32(const float* h, const float* agg, const float* node_in, int kin, long n_rows,
                                            const float* mean, const float* var, float eps, int n_graphs, int rows_per_graph,
                                            const void* w1_f16s, const float* b1s, const void* w2_f16s, const float* b2s, float* y,
                                            float* part_sum, float* part_m2, void* stream) {
    (void)hipGetLastError();
    if (n_rows <= 0) return 0;
    if (kin != 32) return -2;                 // (kin = 64 - the 1-D tw = 50 config - spills at 256 registers: k_node_update serves it)
    if (mean && (n_graphs > NH_MAX_GRAPHS || n_graphs < 1 || rows_per_graph < 1 || !var)) return -2;
    if (part_sum && (n_rows % 128 || !part_m2)) return -2;
    NodeUpdateArgs a;
    a.part_sum = part_sum, a.part_m2 = part_m2;
    a.mean = mean, a.var = var, a.eps = eps, a.n_graphs = n_graphs, a.rows_per_graph = rows_per_graph;
    a.h = h, a.agg = agg, a.node_in = node_in, a.kin = kin, a.n = n_rows, a.y = y, a.b1 = b1s, a.b2 = b2s;
    a.w1 = reinterpret_cast<const f32x4*>(w1_f16s), a.w2 = reinterpret_cast<const f32x4*>(w2_f16s);
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const long ntiles = (n_rows + 127) / 128;
    const int grid = (int)(ntiles < 2 * ncu ? ntiles : 2 * ncu);
    hipStream_t s = (hipStream_t)stream;
    if (mean) hipLaunchKernelGGL((k_node_update2<9, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_node_update2<9, false>), dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" int magnet_mpnn_node_update_f32(const float* h, const float* agg, const float* node_in, int kin, long n_rows,
                                           const float* mean, const float* var, float eps, int n_graphs, int rows_per_graph,
                                           const void* w1_f16, const float* b1, const void* w2_f16, const float* b2, float* y,
                                           float* part_sum, float* part_m2, void* stream) {
    (void)hipGetLastError();
    if (n_rows <= 0) return 0;
    if (kin != 32 && kin != 64) return -2;
    if (mean && (n_graphs > NH_MAX_GRAPHS || n_graphs < 1 || rows_per_graph < 1 || !var)) return -2;
    if (part_sum && (n_rows % 128 || !part_m2)) return -2;        // (tiles must be whole: every row of a tile enters its statistics)
    NodeUpdateArgs a;
    a.part_sum = part_sum, a.part_m2 = part_m2;
    a.mean = mean, a.var = var, a.eps = eps, a.n_graphs = n_graphs, a.rows_per_graph = rows_per_graph;
    a.h = h, a.agg = agg, a.node_in = node_in, a.kin = kin, a.n = n_rows, a.y = y, a.b1 = b1, a.b2 = b2;
    a.w1 = reinterpret_cast<const f32x4*>(w1_f16), a.w2 = reinterpret_cast<const f32x4*>(w2_f16);
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    const long ntiles = (n_rows + 127) / 128;
    const int grid = (int)(ntiles < ncu ? ntiles : ncu);
    hipStream_t s = (hipStream_t)stream;
    static int burst = -1;             // MAGNET_NODE_BURST=0: the persistent kernel for small launches too (A/B knob)
    if (burst < 0) { const char* e = getenv("MAGNET_NODE_BURST"); burst = (e && e[0] == '0') ? 0 : 1; }
    if (burst && kin == 32 && !mean && !part_sum && ntiles <= ncu) {
        hipLaunchKernelGGL(k_node_update_burst<false>, dim3((int)ntiles), dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
    if (kin == 32 && mean) hipLaunchKernelGGL((k_node_update<9, true>), dim3(grid), dim3(256), 0, s, a);
    else if (kin == 32) hipLaunchKernelGGL((k_node_update<9, false>), dim3(grid), dim3(256), 0, s, a);
    else if (mean) hipLaunchKernelGGL((k_node_update<10, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_node_update<10, false>), dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// The same update followed by the hop's InstanceNorm (models/mpnn.py:69-70) in ONE launch, for a batch that fits one workgroup:
// y_norm [n_rows][128] = InstanceNorm(h + Swish(W2 Swish(W1 [h | agg | node_in] + b1) + b2)) over blocks of rows_per_graph rows (the
// last block may be short), biased variance, eps inside the square root - what magnet_mpnn_node_update_f32 followed by
// magnet_instance_norm_f32 computes (fixed summation order; equal to them to rounding, not bitwise).  -2 unless kin == 32,
// 1 <= n_rows <= 128 and rows_per_graph >= 1: the caller then issues the two launches.  y_norm must not alias h.
extern "C" int magnet_mpnn_node_update_norm_small_f32(const float* h, const float* agg, const float* node_in, int kin, long n_rows,
                                                      int rows_per_graph, float eps, const void* w1_f16, const float* b1,
                                                      const void* w2_f16, const float* b2, float* y_norm, void* stream) {
    (void)hipGetLastError();
    if (n_rows <= 0) return 0;
    if (kin != 32 || n_rows > 128 || rows_per_graph < 1) return -2;
    NodeUpdateArgs a;
    a.part_sum = nullptr, a.part_m2 = nullptr, a.mean = nullptr, a.var = nullptr;
    a.eps = eps, a.n_graphs = (int)((n_rows + rows_per_graph - 1) / rows_per_graph), a.rows_per_graph = rows_per_graph;
    a.h = h, a.agg = agg, a.node_in = node_in, a.kin = kin, a.n = n_rows, a.y = y_norm, a.b1 = b1, a.b2 = b2;
    a.w1 = reinterpret_cast<const f32x4*>(w1_f16), a.w2 = reinterpret_cast<const f32x4*>(w2_f16);
    hipLaunchKernelGGL(k_node_update_burst<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------- EDSR convolution on a small grid, weights resident
// y[p][cg*32 ..] = act(sum_taps W_tap . fmap[tap_idx[tap][p]] + b) (+ res[p]) for one 32-column group of a k x k "same" convolution
// over a channels-last map (models/backbones/edsr.py:32-60; the gathered-tap GEMM of magnet_conv_tap_index).  On the LR grid of the
// MAgNet[CNN] configs (64 x 64 pixels) the generic launch is 128 workgroups that each stream 36 weight chunks behind a barrier with one
// k tile of gathered-source lead: 37 us per convolution, all latency, 150 convolutions per C4b rollout.  Here a workgroup of two waves
// (64 pixels) keeps its column group's planes (K x 32 columns: 144 KiB at K = 1152) in LDS after ONE burst, the waves are independent,
// and each keeps RING k tiles of gathered rows in flight.
struct ConvResidentArgs {
    const float* fmap;         // [P + 1][width] channels-last rows, row P all zero (padding)
    const int32_t* tap;        // [ntaps][P] neighbour row of every pixel per tap
    const f32x4* w[4];         // per column group: fp16 planes [nkt][1][2][2][64] x 16 B
    const float* b[4];
    const float* res;          // [P + 1][ldy] or nullptr
    float* y;                  // [P + 1][ldy]
    int P, width, ntaps, ldy, relu;
    int32_t* status;           // nullptr = off: status_bit is OR-ed in when an output reaches 65504 (see magnet_mlp_args_t.status)
    int status_bit;
};
template <int NKT, int TPT /* k tiles per tap = width / 32 */, int RING /* k tiles of gathered rows in flight per wave */>
__global__ void __launch_bounds__(128, 1) k_conv_resident(ConvResidentArgs a) {
    constexpr int MODE = GEMM_FP16X3;
    static_assert(NKT % RING == 0, "ring");
    __shared__ __attribute__((aligned(16))) u32x4 wl[NKT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, j = lane & 31;
    const int cg = blockIdx.y;
    const int row = (blockIdx.x * 2 + wave) * 32 + j;
    const int rowc = row < a.P ? row : a.P - 1;
    // this lane's gathered row per tap (requested first: the addresses of everything below depend on them)
    int trow[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) trow[t] = t < a.ntaps ? a.tap[(size_t)t * a.P + rowc] : a.P;
    {
        const u32x4* g = reinterpret_cast<const u32x4*>(a.w[cg]);
        constexpr int NQ = NKT * 256 / 128;                            // 16-byte pieces per thread
        constexpr int BATCH = NQ < 72 ? NQ : 72;
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += BATCH) {
            u32x4 wv[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) wv[q] = g[(q0 + q) * 128 + threadIdx.x];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) wl[(q0 + q) * 128 + threadIdx.x] = wv[q];
        }
    }
    f32x4 xr[RING][4];
    auto request = [&](int kt, int slot) {                            // (kt is a compile-time value at every call: static tap index)
        const int t = kt / TPT, cb = kt - t * TPT;
        const float* p = a.fmap + (size_t)trow[t] * (TPT * 32) + cb * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[slot][g] = *reinterpret_cast<const f32x4*>(p + 8 * g);
    };
#pragma unroll
    for (int s = 0; s < RING; ++s) request(s, s);
    f32x16 acc[1], accc[1];
    init_bias<1>(acc, a.b[cg], half);
    zero_tiles<1>(accc);
    __syncthreads();                                                  // the planes are in LDS
#pragma unroll
    for (int k0 = 0; k0 < NKT; k0 += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            f32x16 x;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) x[4 * g + c] = xr[s][g][c];
            mma_chunk<1, MODE, false>(x, acc, accc, wl + (k0 + s) * 256, lane);
            if (k0 + s + RING < NKT) request(k0 + s + RING, s);
        }
    }
    fold_tiles<1, MODE>(acc, accc);
    if (a.relu) apply_act<ACT_RELU>(acc[0]);
    if (a.status) {                                                   // (rows past the end are copies of the last row: nothing new)
        float sat = 0.f;
        track_abs_max(acc[0], sat);
        if (__builtin_amdgcn_ballot_w64(sat >= 65504.0f) && lane == 0) atomicOr(a.status, a.status_bit);
    }
    if (row < a.P) {
        float* yo = a.y + (size_t)row * a.ldy + cg * 32 + 4 * half;
        const float* ro = a.res ? a.res + (size_t)row * a.ldy + cg * 32 + 4 * half : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o = {acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]};
            if (ro) o += *reinterpret_cast<const f32x4*>(ro + 8 * g);
            *reinterpret_cast<f32x4*>(yo + 8 * g) = o;
        }
    }
}

// One k x k convolution (ntaps <= 9 gathered taps of `width` in {32, 128} channels each) into ncg <= 4 column groups of 32 output
// channels: y[:, 32 g .. 32 g + 32) for g < ncg.  w[g] / b[g]: the per-column-group fp16x3 packs the generic launch takes
// (magnet_amd/models/backbones/edsr.py).  Returns -2 for shapes outside this kernel (the caller then uses magnet_mlp_fp16x3_f32).
extern "C" int magnet_conv_resident_f32(const float* fmap, int width, const int32_t* tap, int ntaps, int P, const void* const* w,
                                        const float* const* b, int ncg, int relu, const float* res, float* y, int ldy, int32_t* status,
                                        int status_bit, void* stream) {
    (void)hipGetLastError();
    if (P <= 0) return 0;
    if ((width != 32 && width != 128) || (ntaps != 9 && ntaps != 3) || ncg < 1 || ncg > 4) return -2;
    ConvResidentArgs a;
    a.fmap = fmap, a.tap = tap, a.res = res, a.y = y, a.P = P, a.width = width, a.ntaps = ntaps, a.ldy = ldy, a.relu = relu;
    a.status = status, a.status_bit = status_bit;
    for (int g = 0; g < ncg; ++g) a.w[g] = reinterpret_cast<const f32x4*>(w[g]), a.b[g] = b[g];
    const dim3 grid((P + 63) / 64, ncg), block(128);
    // (the tap index of a k tile is compile-time: 3 x 3 in 2-D, k = 3 in 1-D)
    if (ntaps == 9 && width == 32) hipLaunchKernelGGL((k_conv_resident<9, 1, 9>), grid, block, 0, (hipStream_t)stream, a);
    else if (ntaps == 9) hipLaunchKernelGGL((k_conv_resident<36, 4, 9>), grid, block, 0, (hipStream_t)stream, a);
    else if (ntaps == 3 && width == 32) hipLaunchKernelGGL((k_conv_resident<3, 1, 3>), grid, block, 0, (hipStream_t)stream, a);
    else if (ntaps == 3) hipLaunchKernelGGL((k_conv_resident<12, 4, 6>), grid, block, 0, (hipStream_t)stream, a);
    else return -2;
    return (int)hipGetLastError();
}

extern "C" int magnet_mlp_bf16x6_f32(const MagnetMlpArgs* args, int nl, int ht, int ot, int act, int last_act, int ln,
                                     int epi, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (!args || args->M <= 0) return args && args->M == 0 ? 0 : -1;
    if (args->nkt < 1 || args->nkt > MAGNET_MAX_KT) return -2;
    if (args->ncolgroups > 1 && (nl != 1 || args->ncolgroups > MAGNET_MAX_LAYERS)) return -2;
    return launch_mlp(pick<GEMM_BF16X6>(nl, ht, ot, act, last_act, ln, epi, args->M <= SMALL_M), args, stream);
}

extern "C" int magnet_mlp_bf16_f32(const MagnetMlpArgs* args, int nl, int ht, int ot, int act, int last_act, int ln,
                                  int epi, void* stream) {
    (void)hipGetLastError();
    if (!args || args->M <= 0) return args && args->M == 0 ? 0 : -1;
    if (args->nkt < 1 || args->nkt > MAGNET_MAX_KT) return -2;
    if (args->ncolgroups > 1 && (nl != 1 || args->ncolgroups > MAGNET_MAX_LAYERS)) return -2;
    return launch_mlp(pick<GEMM_BF16>(nl, ht, ot, act, last_act, ln, epi, args->M <= SMALL_M), args, stream);
}

extern "C" int magnet_mlp_fp16x3_f32(const MagnetMlpArgs* args, int nl, int ht, int ot, int act, int last_act, int ln,
                                     int epi, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (!args || args->M <= 0) return args && args->M == 0 ? 0 : -1;
    if (args->nkt < 1 || args->nkt > MAGNET_MAX_KT) return -2;
    if (args->ncolgroups > 1 && (nl != 1 || args->ncolgroups > MAGNET_MAX_LAYERS)) return -2;
    return launch_mlp(pick<GEMM_FP16X3>(nl, ht, ot, act, last_act, ln, epi, args->M <= SMALL_M), args, stream);
}

// the 5-Linear chains with ONE accumulator per output block (Split<GEMM_FP16X3S>, common.cuh): layer[l].w / .b from packing mode
// "fp16x3s" (high plane and bias times 2^11).  Returns -3 for any other chain shape.
extern "C" int magnet_mlp_fp16x3s_f32(const MagnetMlpArgs* args, int nl, int ht, int ot, int act, int last_act, int ln,
                                      int epi, void* stream) {
    (void)hipGetLastError();
    if (!args || args->M <= 0) return args && args->M == 0 ? 0 : -1;
    if (args->nkt < 1 || args->nkt > MAGNET_MAX_KT) return -2;
    if (args->ncolgroups > 1) return -2;
    return launch_mlp(pick<GEMM_FP16X3S>(nl, ht, ot, act, last_act, ln, epi, false), args, stream);
}
extern "C" int magnet_mlp_fp16x3s_supported(int nl, int ht, int ot, int act, int last_act, int ln, int epi) {
    return pick<GEMM_FP16X3S>(nl, ht, ot, act, last_act, ln, epi, false) != nullptr;
}

#ifdef MAGNET_DEV_VARIANTS
extern "C" int magnet_debug_set_mlp_ablation(int value) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_abl), &value, sizeof(int));
}
// cycle trace of one wave: buf = device buffer of 1000 uint64 (stamp, tag pairs), wg = the workgroup whose wave 0 writes it
extern "C" int magnet_debug_set_mlp_trace(void* buf, int wg) {
    int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_trace), &buf, sizeof(void*));
    return rc ? rc : (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_trace_wg), &wg, sizeof(int));
}
#endif
