// Fused MPNN edge kernel, generation 13: row-aligned tiles, message_net_2 in the accumulation registers, ONE software pipeline per wave in
// which the matrix pipe never waits for the epilogue.
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// Generation 12 (mpnn_edge_rowsr.hip) moved W2's planes from LDS into a[0:255] and lost: with 512 registers per lane there is ONE wave per
// SIMD, and its tile was still "GEMM phase (96 MFMAs + the next tile's input production), then epilogue phase (64 Swish per lane, no MFMA)":
// the second wave that used to fill the epilogue phase was gone (profiles/r5/edge_rows_gen12_ab_v1.txt: 1.04 ms for the tiles against
// 0.98 with the weights in LDS and two waves).  Here a tile's 96 MFMAs run as four sub-phases of 24 - output blocks (0, 1) x k steps 0-3,
// blocks (2, 3) x k steps 0-3, blocks (0, 1) x k steps 4-7, blocks (2, 3) x k steps 4-7 - and every sub-phase carries one quarter of the
// VALU / transcendental work between its MFMAs:
//     SP1  (blocks 0-1, k 0-3)   Swish + row sum of the PREVIOUS tile's blocks 2-3 (complete since its SP4)
//     SP2  (blocks 2-3, k 0-3)   Swish + fp16 pieces of this tile's own k steps 4-7 (first read in SP3)
//     SP3  (blocks 0-1, k 4-7)   Swish + fp16 pieces of the NEXT tile's k steps 0-3 (their registers were last read in SP2)
//     SP4  (blocks 2-3, k 4-7)   Swish + row sum of this tile's blocks 0-1 (complete since SP3)
// No second accumulator set.  The activation pieces do not live in registers either (64 of them on top of the accumulators' 64, six sets of
// gathered columns, the Swish temporaries and a dozen lane invariants: hipcc spilled W2 fragments from the AGPRs to scratch): a lane
// parks the pieces it produced in a wave-private 16 KiB LDS block, 16 bytes per lane and (k step, plane) - LDS as a register file: no
// other lane ever reads them - and fetches the operands of a double group one double group ahead (two ds_read_b128).
//
// STRAGGLERS (the 33rd edge of an interior MPNN row) never cost a gather of their own: while a row tile's inputs are produced, the wave
// also loads the Bm row of the row's 33rd edge (512 coalesced bytes), forms its 128 activations with the A row it has parked in LDS anyway
// (two values per lane) and parks the fp16 pieces in LDS; the row itself stores sum / 33.  After 32 rows the parked pieces ARE a 32-edge
// activation tile: one S tile (operands straight from that park, no production) runs the GEMM and adds message / 33 to its rows in agg
// (the rows were written by this very wave at least three sub-phases earlier; the loads bypass the vector L1).  Rows without a 33rd edge are stored by their own tile; rows with more than one edge beyond the
// tile (degree > 33: no radius graph of the reference) keep generation 11's contract - the mean over those edges in agg[row] on entry.
//
// The MFMAs are inline asm (the B operand is an AGPR: hipcc would copy it to a VGPR first); hipcc therefore pads none of their hazards -
// the sub-phase order puts >= 6 MFMAs between the last write of an accumulator and its first VALU read, and magnet_amd/hazard_check.py
// verifies every path of the linked code object at build time.
#include "common.cuh"

typedef unsigned int u32;

namespace {

constexpr float RP_SCALE = 2048.0f;        // every MFMA product carries 2^11 (host-scaled weight planes, packing.pack_edge32t_fp16x2)
constexpr int XP_RS = 132;                 // words per parked activation row: 64 hi + 64 lo + 4 (row stride = 4 banks: 16 rows tile the 64)

__device__ __forceinline__ void rp_mfma(f32x16& c, const u32x4& x, const u32x4& w) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "a"(w));
}
__device__ __forceinline__ void rp_mfma0(f32x16& c, const u32x4& x, const u32x4& w) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(x), "a"(w));
}
// ... with two wait states in front: for the rarely taken variants of a sub-phase (short rows, S tiles, pipeline fill), where hipcc moves
// whole accumulators between register ranges around the long epilogue code and puts the v_mov right in front of the asm it cannot read
__device__ __forceinline__ void rp_mfma_s(f32x16& c, const u32x4& x, const u32x4& w) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "a"(w));
}
__device__ __forceinline__ void rp_mfma0_s(f32x16& c, const u32x4& x, const u32x4& w) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(x), "a"(w));
}
// orders the MFMAs that wrote the pair in front of the VALU reads behind it (no wait states of its own: see the header)
// (`behind`: the accumulator the MFMA in front of the statement has just written - it ties the statement, and with it every read of the
//  pair, behind that MFMA: eight passes between the pair's last MFMA and its first reader)
__device__ __forceinline__ void rp_pair_ready(f32x16& a, f32x16& b, f32x16& behind) { asm volatile("" : "+v"(a), "+v"(b), "+v"(behind)); }

template <bool L2E>
__device__ __forceinline__ void rp_pieces(float v0, float v1, u32& hw, u32& lw) {
    const float e0 = __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E), e1 = __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E);
    const float r0 = __builtin_amdgcn_rcpf(1.0f + e0), r1 = __builtin_amdgcn_rcpf(1.0f + e1);
    u32 h, l;
    asm("s_nop 0\n\t"
        "v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %5, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l)
        : "v"(v0), "v"(v1), "v"(r0), "v"(r1));
    hw = h;
    lw = l;
}
// 16 bytes at a wave-uniform base + a 32-bit per-lane byte offset (global_load with an SGPR base: one address register per gather stream)
typedef __attribute__((address_space(1))) const char* rp_gconst;
__device__ __forceinline__ f32x4 rp_ldg4(const float* base, unsigned off_bytes) {
    return *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>((rp_gconst)(base) + off_bytes);
}
__device__ __forceinline__ void rp_half_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// what an epilogue needs to know about its tile (wave-uniform)
struct RpTile {
    int kind;              // 0 = none (pipeline fill), 1 = row tile, 2 = S tile
    int row, cnt, rem, deg, pos;
};

}  // namespace

// ABL (development builds of this file alone, -DRP_ABL_BUILD; results INVALID): 1 no MFMAs, 2 no epilogue fillers, 4 no production fillers,
// 8 no gather requests, 16 no operand fetches from LDS, 32 no slot barriers (hipcc places the fillers), 64 no straggler parking,
// 128 no A row hand-over, 256 no A-column reads, 512 no epilogue finish / rare epilogues
template <bool L2E, bool STRAG, int ABL = 0>
__global__ void __launch_bounds__(256) k_mpnn_edge_rowsp(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const int32_t* __restrict__ csr_src, const int4* __restrict__ tiles, int ntiles,
                                                         const u32x4* __restrict__ w2p /* [8 ks][4 nt][2 planes][64] */,
                                                         const float* __restrict__ b2, float* __restrict__ agg) {
    constexpr float l2e = L2E ? MAGNET_LOG2E : 1.0f;                       // the Swish outputs carry log2 e when the inputs do
    __shared__ __attribute__((aligned(16))) float arow[4][128];            // the A row of the tile whose inputs are in production
    __shared__ __attribute__((aligned(16))) u32 xparks[STRAG ? 4 : 1][STRAG ? 32 * XP_RS : 4];      // parked straggler activations
    __shared__ __attribute__((aligned(16))) u32x4 pbufs[4][8 * 2 * 64];    // the current tile's pieces: [k step][plane][lane] x 16 B
    __shared__ int rparks[4][32];                                          // row of group position p (-1: no straggler)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nb8 = gridDim.x / 8;         // XCD-aware contiguous ranges (k_mpnn_edge6): consecutive tiles = neighbouring targets
    const int vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + gridDim.x * 4 - 1) / (gridDim.x * 4);
    const int t_lo = (vb * 4 + wave) * tpw, t_hi = t_lo + tpw < ntiles ? t_lo + tpw : ntiles;
    if (t_lo >= t_hi) return;              // (no workgroup barrier anywhere in this kernel)
    u32x4 wq[8][4][2];                     // both planes of W2, resident in a[0:255]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) wq[ks][nt][p] = w2p[((ks * 4 + nt) * 2 + p) * 64 + lane];
    const int n_r = t_hi - t_lo;
    const int G = STRAG ? (n_r + 31) / 32 : 0;         // groups of (up to) 32 row tiles, each FOLLOWED by its S tile
    const int K = n_r + G;
    float* const ar = arow[wave];
    const float* const arl = ar + 8 * half;            // this lane's columns of the parked row: 16 ks + 8 half + 0..7
    u32* const xpark = xparks[STRAG ? wave : 0];
    u32x4* const pbl = pbufs[wave] + lane;             // this lane's slot of (k step ks, plane p): pbl[(2 ks + p) * 64]
    int* const rpark = rparks[wave];
    float bl[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bl[nt] = b2[32 * nt + j] * l2e;
    if (STRAG && lane < 32) rpark[lane] = -1;

    // The wave's sequence: its row tiles in order, and - STRAG - behind every 32nd of them (and behind the last one) the S tile of that
    // group.  Everything the loop needs to know about the tiles ahead arrives through VECTOR loads issued a whole tile earlier (a scalar
    // load in the loop is waited for with lgkmcnt(0), i.e. together with every LDS operation in flight, and a load used right behind its
    // issue stalls the wave for a memory round trip: the first version of this kernel spent half its time that way).
    // ---- pipeline state
    u32x4 xa[2][2];                        // MFMA operands of the double group in flight / the next one: [buffer][plane (0 = high)]
    f32x16 out[4];
    f32x4 gc[4][2];                        // gathered Bm columns of k steps 4-7 of the tile that is (about to be) current
    RpTile cur, prv;                       // the current tile; the previous one (its blocks 2-3 are finished in SP1)
    float accB[2] = {0.f, 0.f};            // running sums of the previous tile's blocks 2, 3 (SP1); of this tile's blocks 0, 1: accA (SP4)
    int rc = 0;                            // row tile (inside the wave) of the current element when it is one
    int rn = 1;                            // the next row tile of the sequence (an S tile in between does not count)
    int4 dnxt;                             // ... its descriptor
    unsigned pb_cur;                       // byte offset of Bm[src][8 half] for this lane's edge of the current / the next row tile
    unsigned pb_nxt;
    int s33n = 0;                          // source of the next row tile's 33rd edge
    f32x2 arv;                             // the next tile's A row on its way to LDS (this lane's two columns)
    f32x2 sgv = {0.f, 0.f};                // the current row's 33rd edge: this lane's two columns of its Bm row

    auto describe_row = [=](int r, const int4& d) {
        RpTile t;
        t.kind = 1, t.row = d.y, t.cnt = d.z & 0xFF, t.rem = d.z >> 8, t.deg = d.w, t.pos = r & 31;
        return t;
    };
    // descriptor of row tile r (clamped) through a per-lane load of the same address: counted by vmcnt, not waited for with the LDS traffic
    auto vload_desc = [=](int r) {
        int rr = r < n_r - 1 ? r : n_r - 1;
        asm volatile("" : "+v"(rr));               // (a VGPR index: a scalar load otherwise)
        return tiles[t_lo + rr];
    };
    auto uniform4 = [](const int4& v) {
        return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                         __builtin_amdgcn_readfirstlane(v.w));
    };
    // the lane's source row of a row tile (d: its descriptor) as a byte offset into Bm; the source of the tile's 33rd edge
    auto vload_src = [=](const int4& d) {
        const int c1 = (d.z & 0xFF) - 1, jj = j < c1 ? j : c1;
        return csr_src[d.x + (jj > 0 ? jj : 0)];
    };
    auto vload_s33 = [=](const int4& d) {
        int e = d.x + ((d.z >> 8) == 1 ? 32 : 0);
        asm volatile("" : "+v"(e));
        return csr_src[e];
    };
    auto pb_of = [=](int src) { return ((unsigned)src * 128u + 8u * half) * 4u; };
    auto request = [=](f32x4 (&set)[2], unsigned pb, int ks) {
        if (ABL & 8) return;
        set[0] = rp_ldg4(Bm, pb + 64u * ks);
        set[1] = rp_ldg4(Bm, pb + 64u * ks + 16u);
    };
    // ---- the MFMAs of a sub-phase, one at a time: m = 0 .. 5 of the double group (k step ks, output blocks n0 / n1, alternating)
    // (the operands sit in xa[buf]: buf = parity of the double group's number inside the tile - 16 per tile, so it never needs a swap)
    auto mfma_of = [&](int m, int ks, int n0, int n1, int buf, bool safe) {
        const int nt = (m & 1) ? n1 : n0;
        const u32x4& x = xa[buf][(m >> 1) == 1 ? 1 : 0];                      // MFMAs 2, 3 take the low plane of the activations
        const u32x4& w = wq[ks][nt][m < 2 ? 1 : 0];                           // MFMAs 0, 1 the low plane of the weights
        if (ABL & 1) {                             // development: the operands stay alive, the matrix pipe stays idle
            asm volatile("" : "+v"(out[nt]) : "v"(x), "a"(w));
            return;
        }
        if (m < 2 && ks == 0) {
            if (safe) rp_mfma0_s(out[nt], x, w);
            else rp_mfma0(out[nt], x, w);
        } else {
            if (safe) rp_mfma_s(out[nt], x, w);
            else rp_mfma(out[nt], x, w);
        }
    };
    // operands of k step ks -> xa[buf] from an operand source: a row tile's pieces sit in the lane's own slots of the piece buffer
    // (k steps 2 KiB apart, planes 1 KiB), an S tile's in the park (lane = edge: row j, 32 bytes per k step, low plane 256 bytes on)
    struct OpSrc { const char* base; int kstride, plane; };
    auto op_src = [=](bool s_tile) {
        OpSrc o;
        if (STRAG && s_tile) o.base = reinterpret_cast<const char*>(xpark + j * XP_RS + 4 * half), o.kstride = 32, o.plane = 256;
        else o.base = reinterpret_cast<const char*>(pbl), o.kstride = 2048, o.plane = 1024;
        return o;
    };
    auto fetch = [&](int buf, int ks, const OpSrc& o) {
        if (ABL & 16) return;
        xa[buf][0] = *reinterpret_cast<const u32x4*>(o.base + ks * o.kstride);
        xa[buf][1] = *reinterpret_cast<const u32x4*>(o.base + ks * o.kstride + o.plane);
    };
    // ---- fillers.  A wave issues in order: what is to run beside an MFMA has to FOLLOW it in the instruction stream, a few instructions
    // per MFMA (a 32x32x16 MFMA occupies the pipe for 32 cycles = 5 .. 8 issue slots); every slot below ends in a scheduling barrier.
    // (1) input production of one k step (8 values per lane = 4 pairs A B C D) over the six slots of a double group:
    //     slot 0: A exp / rcp   1: A pieces, B exp   2: B rcp, pieces   3: C exp / rcp   4: C pieces, D exp   5: D rcp, pieces
    // One wave per SIMD: what hides a VALU / transcendental latency is an INDEPENDENT instruction of the same wave, so two pairs (four
    // values) go through every stage together:  slot 0: A, B exp | 1: A, B rcp + A pieces | 2: B pieces | 3-5: the same for C, D.
    float pv[4] = {0.f, 0.f, 0.f, 0.f}, pr[4] = {0.f, 0.f, 0.f, 0.f};     // the two pairs in flight: pre-activations, 2^-v then 1 / (1 + 2^-v)
    u32x4 wh, wl;                                              // the k step in production: its four words of each plane
    f32x4 la[2];                                               // the A columns of the k step in production (from LDS)
    auto quad_head = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = a[i] + b[i];
        asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]));     // (pinned to its slot, as the epilogue values below)
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = __builtin_amdgcn_exp2f(L2E ? -pv[i] : -pv[i] * MAGNET_LOG2E);
    };
    auto quad_rcp = [&]() {
        asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = __builtin_amdgcn_rcpf(1.0f + pr[i]);
        asm volatile("" : "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]));     // (all four before the first piece: no result is read in the
    };                                                                             //  slot right behind its transcendental)
    auto pair_tail = [&](int i0, u32& hw, u32& lw) {
        u32 h, l;
        asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
            "v_fma_mixhi_f16 %0, %3, %5, 0\n\t"
            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %5, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(l)
            : "v"(pv[i0]), "v"(pv[i0 + 1]), "v"(pr[i0]), "v"(pr[i0 + 1]));
        hw = h, lw = l;
    };
    auto produce_slot = [&](int m, int ks, const f32x4 (&set)[2]) {
        if (ABL & 4) return;
        const int part = m / 3;                                // slots 0-2: columns +0..3 (pairs A, B), slots 3-5: columns +4..7 (C, D)
        u32 hw, lw;
        switch (m % 3) {
            case 0: quad_head(la[part], set[part]); break;
            case 1:
                quad_rcp();
                pair_tail(0, hw, lw);
                wh[2 * part] = hw, wl[2 * part] = lw;
                break;
            default:
                pair_tail(2, hw, lw);
                wh[2 * part + 1] = hw, wl[2 * part + 1] = lw;
                if (part == 1) pbl[(2 * ks) * 64] = wh, pbl[(2 * ks + 1) * 64] = wl;       // the k step is complete
                break;
        }
    };
    auto la_part = [&](int ks, int part) { if (ABL & 256) return; la[part] = *reinterpret_cast<const f32x4*>(arl + 16 * ks + 4 * part); };
    auto la_load = [&](int ks) { la_part(ks, 0), la_part(ks, 1); };
    // (inside a sub-phase the next k step's A columns are read as soon as their registers are free: part 0 in slot 1, part 1 in slot 4)
    auto la_next = [&](int m, int ks_next) {
        if (m == 1) la_part(ks_next, 0);
        if (m == 4) la_part(ks_next, 1);
    };
    // (2) Swish + row sum of a full row tile's output blocks nb, nb + 1: quarter q = accumulator registers 4 q .. 4 q + 3 of both
    //     blocks = 8 values over six slots (2 1 1 2 1 1)
    //     as four PAIRS (register c of both blocks) in two halves each - the pair's two chains cover each other's transcendental
    //     wait states: slot 0: pair 0 exp | 1: pair 0 rcp + sum | 2: pair 1 | 3: pair 2 exp | 4: pair 2 rcp + sum | 5: pair 3
    float ey[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, ee[2][2] = {{0.f, 0.f}, {0.f, 0.f}};       // two pairs in flight (set = pair & 1)
    auto epi_h1 = [&](int nb, int c, int q) {
        const int st = c & 1;
        float o0 = out[nb][4 * q + c], o1 = out[nb + 1][4 * q + c];
        asm volatile("" : "+v"(o0), "+v"(o1));     // (starts in its slot: IR-level code motion ignores the scheduling barriers)
        ey[st][0] = fmaf(o0, 1.0f / RP_SCALE, bl[nb]), ey[st][1] = fmaf(o1, 1.0f / RP_SCALE, bl[nb + 1]);
        ee[st][0] = __builtin_amdgcn_exp2f(L2E ? -ey[st][0] : -ey[st][0] * MAGNET_LOG2E);
        ee[st][1] = __builtin_amdgcn_exp2f(L2E ? -ey[st][1] : -ey[st][1] * MAGNET_LOG2E);
    };
    auto epi_h2 = [&](int c, float (&acc)[2]) {
        const int st = c & 1;
        asm volatile("" : "+v"(ee[st][0]), "+v"(ee[st][1]));
        const float r0 = __builtin_amdgcn_rcpf(1.0f + ee[st][0]), r1 = __builtin_amdgcn_rcpf(1.0f + ee[st][1]);
        acc[0] = fmaf(ey[st][0], r0, acc[0]), acc[1] = fmaf(ey[st][1], r1, acc[1]);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));   // (... and ends there: left alone the sums are sunk behind the last MFMA)
    };
    //     slot 0: pair 0 exp | 1: pair 0 rcp + sum beside pair 1 exp | 2: pair 1 beside pair 2 | 3: pair 2 beside pair 3 | 4: pair 3 rcp + sum
    auto epi_slot = [&](int m, int nb, int q, float (&acc)[2]) {
        if (ABL & 2) return;
        switch (m) {
            case 0: epi_h1(nb, 0, q); break;
            case 1: epi_h1(nb, 1, q); epi_h2(0, acc); break;
            case 2: epi_h1(nb, 2, q); epi_h2(1, acc); break;
            case 3: epi_h1(nb, 3, q); epi_h2(2, acc); break;
            case 4: epi_h2(3, acc); break;
            default: break;
        }
    };
    // ... the same for everything that is not a full row tile, as one block behind the double group's first MFMA (rare: rows shorter than
    // 32 edges, the S tile of a group, the pipeline's first and last element)
    auto epi_chunk = [&](const RpTile& t, int nb, int q, float (&acc)[2]) {
        if (t.kind == 1) {                         // a row tile, full or short: register i of lane half h is edge 8 (i / 4) + 4 h + i % 4
            const int lim = t.cnt - 4 * half - 8 * q;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float y = fmaf(out[nb + b][4 * q + c], 1.0f / RP_SCALE, bl[nb + b]);
                    float m = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
                    asm("" : "+v"(m));             // (computed for every lane, then selected: no branch per value)
                    acc[b] += c < lim ? m : 0.f;
                }
        }
    };
    // S tile: every edge is a row of its own - agg[row] (it holds sum / 33, written by the row's own tile) += message / 33 for output blocks
    // nb, nb + 1.  All 32 loads of the lane are issued before the first value is used: one memory round trip per call, not one per value.
    // (One S tile per 32 row tiles: its addresses are formed HERE from an opaque copy of the lane id - hoisted out of the tile loop as
    //  invariants they cost the hot path a dozen registers it does not have.)
    auto s_epilogue = [&](int nb) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int jj = ln & 31, hh = ln >> 5;
        float* dst[16];
        float sv[16][2];
        bool ok[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {             // register i <-> edge 8 (i / 4) + 4 half + i % 4
            const int row = rpark[8 * (i >> 2) + 4 * hh + (i & 3)];
            ok[i] = row >= 0;
            dst[i] = agg + (size_t)(ok[i] ? row : 0) * 128 + 32 * nb + jj;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)            // (past the vector L1: the row was written by other lanes of this wave)
                sv[i][b] = ok[i] ? __hip_atomic_load(dst[i] + 32 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float y = fmaf(out[nb + b][i], 1.0f / RP_SCALE, bl[nb + b]);
                const float m = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -y : -y * MAGNET_LOG2E));
                if (ok[i]) dst[i][32 * b] = fmaf(m, 1.0f / (33.0f * l2e), sv[i][b]);
            }
    };
    // the row sums of blocks nb, nb + 1 are complete: mean over the row -> agg (or -> the park, when the row's 33rd edge is still to come)
    auto epi_finish = [&](const RpTile& t, int nb, float (&acc)[2], bool with_old) {
        if (t.kind != 1) return;
        rp_half_swap(acc[0], acc[1]);              // -> lanes 0-31: both halves of block nb, lanes 32-63: both halves of block nb + 1
        const float sum = acc[0] + acc[1];
        acc[0] = acc[1] = 0.f;
        const float invd = __builtin_amdgcn_rcpf((float)t.deg);
        float old = 0.f;
        // the mean over the row's edges outside the tile (remainder pass) - unless that is the ONE edge the group's S tile will add
        if (with_old && t.rem > (STRAG ? 1 : 0)) old = agg[(size_t)t.row * 128 + 32 * nb + lane] * ((float)t.rem * invd);
        agg[(size_t)t.row * 128 + 32 * nb + lane] = fmaf(sum, invd * (1.0f / l2e), old);
    };
#define RP_SLOT_END do { if (!(ABL & 32)) __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: element 0 is row tile 0; its k steps 0-3 are produced here, its k steps 4-7 requested (SP2 of the first iteration
    // turns them into pieces)
    const int4 dfirst = uniform4(vload_desc(0));
    cur = describe_row(0, dfirst);
    prv.kind = 0, prv.row = 0, prv.cnt = 32, prv.rem = 0, prv.deg = 1, prv.pos = 0;
    dnxt = uniform4(vload_desc(1));
    pb_cur = pb_of(vload_src(dfirst));
    pb_nxt = pb_of(vload_src(dnxt));
    s33n = __builtin_amdgcn_readfirstlane(vload_s33(dnxt));
    *reinterpret_cast<f32x2*>(ar + 2 * lane) = *reinterpret_cast<const f32x2*>(A + (size_t)dfirst.y * 128 + 2 * lane);
    {
        f32x4 g0[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) request(g0[ks], pb_cur, ks);
        request(gc[0], pb_cur, 4);         // (k steps 6, 7: SP1 of the first iteration, as in every other)
        request(gc[1], pb_cur, 5);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            la_load(ks);
#pragma unroll
            for (int m = 0; m < 6; ++m) produce_slot(m, ks, g0[ks]);
        }
    }
    if (STRAG && cur.rem == 1) {
        const int s33 = __builtin_amdgcn_readfirstlane(vload_s33(dfirst));
        sgv = *reinterpret_cast<const f32x2*>(Bm + (size_t)s33 * 128 + 2 * lane);
    }
    float accA[2] = {0.f, 0.f};
    fetch(0, 0, op_src(false));            // operands of the first double group

    // One double group: six MFMAs on k step ks of blocks n0 / n1 with the operands in xa[dgn & 1]; the operands of the NEXT double group
    // (k step ks_next of this tile - or of the next element, whose kind decides where they are) are fetched behind the second MFMA:
    // the other buffer was last read by the previous double group's MFMAs.  F(m) is the filler slot behind MFMA m.
#define RP_DGROUP_(dgn, ks, n0, n1, ks_next, next_src, SAFE, FILL)                                 \
    _Pragma("unroll") for (int m = 0; m < 6; ++m) {                                                \
        mfma_of(m, ks, n0, n1, (dgn) & 1, SAFE);                                                   \
        if (m == 0 && (dgn) == 0) rp_pair_ready(out[2], out[3], out[0]);  /* (the epilogue pairs are   */ \
        if (m == 0 && (dgn) == 12) rp_pair_ready(out[0], out[1], out[2]); /*  first touched BEHIND an MFMA) */ \
        if (m == 1) fetch(((dgn) + 1) & 1, ks_next, next_src);                                     \
        FILL;                                                                                      \
        RP_SLOT_END;                                                                               \
    }

#define RP_DGROUP(dgn, ks, n0, n1, ks_next, next_src, FILL) RP_DGROUP_(dgn, ks, n0, n1, ks_next, next_src, false, FILL)
    // EVERY element runs the same instruction stream - 96 MFMAs with the fillers of a full row tile between them.  What is not a full
    // row tile (a row shorter than 32 edges, the S tile of a group, the pipeline's fill and drain) lets those fillers work on values
    // nobody uses and does its own epilogue OUTSIDE the stream, un-overlapped: a second copy of a sub-phase with other fillers made
    // hipcc move whole accumulators between register ranges at the joins - v_mov right next to asm MFMAs it cannot see into.
    auto rare_epilogue = [&](const RpTile& t, int nb, float (&acc)[2]) {
        asm volatile("s_nop 11" : "+v"(out[nb]), "+v"(out[nb + 1]));      // (the pair's last MFMA may have been the previous instruction)
        acc[0] = acc[1] = 0.f;
        if (STRAG && t.kind == 2) {
            s_epilogue(nb);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) epi_chunk(t, nb, q, acc);
        epi_finish(t, nb, acc, true);
        acc[0] = acc[1] = 0.f;
    };
    for (int k = 0; k < K; ++k) {
        const bool cur_s = cur.kind == 2;
        // the next element: the S tile of this group (behind its 32nd row tile, or behind the wave's last one), else the next row tile
        const bool nxt_s = STRAG && !cur_s && ((rc & 31) == 31 || rc == n_r - 1);
        const bool nxt_r = !nxt_s && rn < n_r;
        // "full": the element takes the in-stream epilogue - a row tile of 32 edges with nothing to pick up from agg (rows whose remainder
        // pass left a partial mean there take the rare path: a load in the common path would have to be waited for on the spot)
        auto is_full = [=](const RpTile& t) { return t.kind == 1 && t.cnt == 32 && t.rem <= (STRAG ? 1 : 0); };
        const bool full_prv = is_full(prv), full_cur = is_full(cur);
        const OpSrc osc = op_src(cur_s), osn = op_src(nxt_s);      // where this element's / the next element's MFMA operands are
        f32x4 ga[4][2];                    // the next row tile's k steps 0-3
        const int4 vd2 = vload_desc(rn + 1);                       // the row tile after next: on its way now, looked at in SP3
        // ---- SP1: blocks 0-1, k steps 0-3 | Swish + row sum of the previous tile's blocks 2-3
        // (EVERY load of the loop is unconditional - the addresses are valid whatever the next element is: behind a branch that issues
        //  loads on one side only hipcc can no longer count them and waits with vmcnt for the youngest, i.e. for a full round trip)
        if (!(ABL & 128)) arv = *reinterpret_cast<const f32x2*>(A + (size_t)dnxt.y * 128 + 2 * lane);
        request(gc[2], pb_cur, 6);
        request(gc[3], pb_cur, 7);
        la_load(4);                        // (the A columns SP2 starts with: this row's, parked since the previous tile's SP2)
        if (!full_prv && !(ABL & 512)) rare_epilogue(prv, 2, accB);
#pragma unroll
        for (int d = 0; d < 4; ++d) { RP_DGROUP(d, d, 0, 1, (d + 1) & 3, osc, epi_slot(m, 2, d, accB)) }
        if (full_prv && !(ABL & 512)) epi_finish(prv, 2, accB, false);
        else accB[0] = accB[1] = 0.f;
        if (STRAG && prv.kind == 2 && lane < 32) rpark[lane] = -1;        // (that group's rows are written: a fresh park for this one)
        // ---- SP2: blocks 2-3, k steps 0-3 | this tile's own k steps 4-7 (an S tile has none: its fillers chew on stale registers and
        // leave garbage in piece slots 4-7, which nobody reads before the next row tile's own SP2 rewrites them)
        // (the next row tile's k steps 0-3 are requested inside this sub-phase, two now and two when the first two sets of this tile's
        //  k steps 4-7 have been consumed: never more than six sets - 48 registers - of gathered columns in flight)
        request(ga[0], pb_nxt, 0);
        request(ga[1], pb_nxt, 1);
        const f32x2 a2 = *reinterpret_cast<const f32x2*>(ar + 2 * lane);     // (this row's A columns for its 33rd edge, before the hand-over)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // (the last double group's fourth slot reads the last A column of this row: the next row is handed over in slot 4 and the first
            //  A columns SP3 needs are read in slot 5, a whole MFMA ahead of their use)
            RP_DGROUP(4 + d, d, 2, 3, d < 3 ? d + 1 : 4, osc, produce_slot(m, 4 + d, gc[d]); if (d < 3) la_next(m, 5 + d);
                      if (d == 3 && m == 4 && !(ABL & 128)) *reinterpret_cast<f32x2*>(ar + 2 * lane) = arv;
                      if (d == 3 && m == 5) la_load(0))
            if (d == 1) {
                request(ga[2], pb_nxt, 2);
                request(ga[3], pb_nxt, 3);
            }
        }
        if (STRAG && !cur_s && !(ABL & 64)) {   // the row's 33rd edge: its activations -> the park
            if (cur.rem == 1) {
                u32 hw, lw;
                rp_pieces<L2E>(a2[0] + sgv[0], a2[1] + sgv[1], hw, lw);
                xpark[cur.pos * XP_RS + lane] = hw;
                xpark[cur.pos * XP_RS + 64 + lane] = lw;
            }
            if (lane == 0) rpark[cur.pos] = cur.rem == 1 ? cur.row : -1;
        }
        // ---- SP3: blocks 0-1, k steps 4-7 | the next tile's k steps 0-3 (when the next element is not a row tile: garbage into piece
        // slots 0-3 - an S tile takes its operands from the park, and the row tile behind it rewrites the slots in the S tile's SP3)
        if (STRAG && !(ABL & 64)) sgv = *reinterpret_cast<const f32x2*>(Bm + (size_t)s33n * 128 + 2 * lane);
        // the row tile after next: its descriptor has arrived (requested at the top of the iteration) - now its source indices
        const int4 d2 = uniform4(vd2);
        const int vsrc2 = vload_src(d2), v33_2 = STRAG ? vload_s33(d2) : 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            RP_DGROUP(8 + d, 4 + d, 0, 1, d < 3 ? 5 + d : 4, osc, produce_slot(m, d, ga[d]); if (d < 3) la_next(m, 1 + d))
        }
        // ---- SP4: blocks 2-3, k steps 4-7 | Swish + row sum of this tile's blocks 0-1.  Its last double group fetches the NEXT element's
        // k step 0 (produced in SP3 - or parked, when that element is the group's S tile)
        request(gc[0], pb_nxt, 4);
        request(gc[1], pb_nxt, 5);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (d < 3) { RP_DGROUP(12 + d, 4 + d, 2, 3, 5 + d, osc, epi_slot(m, 0, d, accA)) }
            else { RP_DGROUP(15, 7, 2, 3, 0, osn, epi_slot(m, 0, 3, accA)) }
        }
        if (ABL & 512) {
        } else if (full_cur) epi_finish(cur, 0, accA, false);
        else rare_epilogue(cur, 0, accA);
        prv = cur;
        if (nxt_s) {
            cur.kind = 2, cur.row = 0, cur.cnt = 32, cur.rem = 0, cur.deg = 33, cur.pos = 0;
        } else if (nxt_r) {                // the sequence moves on to row tile rn; the one after it is the one prefetched above
            cur = describe_row(rn, dnxt);
            rc = rn, rn = rn + 1;
            dnxt = d2;
            pb_cur = pb_nxt, pb_nxt = pb_of(vsrc2);
            s33n = __builtin_amdgcn_readfirstlane(v33_2);
        } else {
            cur.kind = 0;
        }
    }
    // ---- drain: the last tile's blocks 2-3
    rare_epilogue(prv, 2, accB);
#undef RP_DGROUP
#undef RP_DGROUP_
#undef RP_SLOT_END
}

// Generation 13 (k_mpnn_edge_rowsp).  Arguments and contract as magnet_mpnn_edge_rowsg_fp16x3_f32 (include/magnet_hip.h).
extern "C" int magnet_mpnn_edge_rowsp_fp16x3_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                                 int has_stragglers, const void* w2_t32, const float* b2, float* agg,
                                                 int a_bm_log2e_scaled, void* stream) {
    (void)hipGetLastError();
    if (n_tiles <= 0) return 0;
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
        ncu = prop.multiProcessorCount;
    }
    const dim3 g(min((n_tiles + 3) / 4, ncu)), b(256);
    const int4* td = reinterpret_cast<const int4*>(tiles);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
    hipStream_t s = (hipStream_t)stream;
    const bool l2e = (a_bm_log2e_scaled & 1) != 0;
    if (has_stragglers) {
        if (l2e) hipLaunchKernelGGL((k_mpnn_edge_rowsp<true, true>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
        else hipLaunchKernelGGL((k_mpnn_edge_rowsp<false, true>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
    } else {
        if (l2e) hipLaunchKernelGGL((k_mpnn_edge_rowsp<true, false>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
        else hipLaunchKernelGGL((k_mpnn_edge_rowsp<false, false>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg);
    }
    return (int)hipGetLastError();
}

#ifdef RP_ABL_BUILD
// development: the same launch with an ablation of the pipeline (scripts/dev/bench_rowsp_abl.py builds this file alone with -DRP_ABL_BUILD)
#define RP_ABL_CASE(n) case n: hipLaunchKernelGGL((k_mpnn_edge_rowsp<true, true, n>), g, b, 0, s, A, Bm, src, td, n_tiles, ws, b2, agg); break;
extern "C" int magnet_mpnn_edge_rowsp_abl_f32(const float* A, const float* Bm, const int32_t* src, const void* tiles, int n_tiles,
                                              const void* w2_t32, const float* b2, float* agg, int abl, int ncu, void* stream) {
    (void)hipGetLastError();
    const dim3 g(min((n_tiles + 3) / 4, ncu)), b(256);
    const int4* td = reinterpret_cast<const int4*>(tiles);
    const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
    hipStream_t s = (hipStream_t)stream;
    switch (abl) {
        RP_ABL_CASE(0) RP_ABL_CASE(1) RP_ABL_CASE(2) RP_ABL_CASE(4) RP_ABL_CASE(6) RP_ABL_CASE(8) RP_ABL_CASE(16) RP_ABL_CASE(32)
        RP_ABL_CASE(14) RP_ABL_CASE(30) RP_ABL_CASE(31) RP_ABL_CASE(7) RP_ABL_CASE(95) RP_ABL_CASE(159) RP_ABL_CASE(287) RP_ABL_CASE(543)
        RP_ABL_CASE(1023) RP_ABL_CASE(1022) RP_ABL_CASE(990) RP_ABL_CASE(1006) RP_ABL_CASE(64) RP_ABL_CASE(128) RP_ABL_CASE(512) RP_ABL_CASE(704)
        default: return -2;
    }
    return (int)hipGetLastError();
}
#endif
