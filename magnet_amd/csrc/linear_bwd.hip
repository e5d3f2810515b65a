// One hidden Linear of an MLP chain walked back in ONE kernel: the data gradient, the ReLU mask of the layer below, the weight gradient
// and the bias gradient (torch autograd through models/backbones/mlp.py:9-28 - Linear -> ReLU - at hidden width 128):
//     dx [M][128] = (x > 0) * (dz W)            dW [128 out][128 in] += dz^T x            db [128] += colsum(dz)
// Before: k_mlp_bf16<bf16x6, 1 Linear, EPI_STORE_MASK> (reads dz and x, writes dx) + k_wgrad_split (reads dz and x AGAIN) + two
// reductions - five [M, 128] array passes per layer, three here.  Same construction as k_mpnn_edge_bwd_w (mpnn_edge_bwd.hip, whose header
// explains the arithmetic; a 64-wide twin for the hidden width of MAgNet[CNN]'s MLPs): one workgroup of four waves per CU, persistent over 128-row steps; every wave owns a 32-row slab of dW (rows
// = features of dz, 64 accumulator registers) and meets the other waves' rows through LDS.  Only ONE weight orientation is needed here,
// so beside its 64 KiB the LDS holds G dz of all four tiles of a step (72 KiB, feature-major) and one 32-feature block of x of all four
// tiles (18 KiB): every wave stages its own rows - no publisher, no conditional code - and each round's 24 MFMAs take the block of x just
// staged against the wave's own slab of dz, which it read once per step and keeps as fp16 planes.  dz ~ 1e-7: staged times G = 2^k from
// max|dz| (the previous layer's launch leaves the maxima of what it wrote: amax_out; the first layer of a chain takes them from
// k_absmax_partial); the data GEMM scales every row by its own power of two instead (exact for a linear map).
#include "common.cuh"
#include "split_single.cuh"

// Development ablations (scripts/dev/build_lb_ablations.sh, profiles/r4/linear_bwd_ablations.txt): -DLB_ABL=<bits> compiles pieces out -
// 1 the loop's loads, 2 the dx stores, 4 the weight gradient's MFMAs, 8 the data GEMM, 16 the x operand reads + splits, 32 the barriers.
// Timing only (the results are wrong); the shipped library is built with 0.
#ifndef LB_ABL
#define LB_ABL 0
#endif
#define LB_BARRIER() do { if (!(LB_ABL & 32)) wg_barrier(); } while (0)
#define LB_ES 36                        // staging row stride (floats): 32 rows + 4 pad
struct LinBwdArgs {
    const float* dz;                    // [M][ldz] gradient at the layer's output, 128 columns
    const float* x;                     // [M][ldx] the layer's input (the activated output of the layer below)
    const u32x4* wt;                    // fp16x3 planes of W^T as a Linear [128 in][128 out] (standard pack; the high plane is scaled here)
    float* dx;                          // [M][lddx]
    float* wpart;                       // [grid][128][128] per-workgroup partial sums of dW
    float* bpart;                       // [grid][128]
    const float* amax_in;               // n_amax block maxima of |dz|
    float* amax_out;                    // [grid] block maxima of |dx| (nullptr: not wanted)
    long M;
    int ldz, ldx, lddx, n_amax, mask;
};

// NT = width / 32.  NT = 4: every wave owns one 32-row slab of dW and meets the four 32-feature blocks of x in four rounds.  NT = 2 (hidden
// width 64): two waves share a slab of dz, each takes one of the two blocks of x - both staged in a single round.
template <int NT>
__global__ void __launch_bounds__(256, NT == 2 ? 2 : 1) k_linear_bwd(const LinBwdArgs a) {
    constexpr int ES = LB_ES, W = 32 * NT;
    constexpr int XB = 1;                                                     // blocks of x staged per round (NT = 2: 72 KiB of LDS, two
                                                                              // workgroups per CU - one's memory phase under the other's MFMAs)
    constexpr int NR = NT / XB;                                               // rounds per step
    constexpr int NACC = NT == 4 ? 4 : 1;
    __shared__ __attribute__((aligned(16))) u32x4 wl[NT * NT * 4 * 64];       // W^T planes (2^11 hw, 2^11 lw)
    __shared__ __attribute__((aligned(16))) float sdz[4 * W * ES];            // G dz of the step's four tiles, [tile][feature][row]
    __shared__ __attribute__((aligned(16))) float sx[4 * 32 * XB * ES];       // XB 32-feature blocks of x, [tile][feature][row]
    __shared__ float gsc[4];
    for (int i = threadIdx.x; i < NT * NT * 4 * 64; i += 256) wl[i] = ((i >> 6) & 1) == 0 ? scale_plane_2048(a.wt[i]) : a.wt[i];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int slab = NT == 4 ? wave : (wave & 1);                             // this wave's 32 features of dz (= rows of dW)
    const int xblk = NT == 4 ? 0 : (wave >> 1);                               // NT = 2: its block of x (= columns of dW)
    float G, invG;
    {
        float m = 0.f;
        for (int i = threadIdx.x; i < a.n_amax; i += 256) m = fmaxf(m, a.amax_in[i]);
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) gsc[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(gsc[0], gsc[1]), fmaxf(gsc[2], gsc[3]));
        const unsigned eb = (__float_as_uint(m) >> 23) & 0xFFu;               // max|dz| in [2^(eb-127), 2^(eb-126))
        const bool ok = eb >= 14u && eb < 255u;
        G = ok ? __uint_as_float((267u - eb) << 23) : 1.0f;                   // 2^(140-eb): G max|dz| < 2^14
        invG = ok ? __uint_as_float((eb - 13u) << 23) : 1.0f;
    }
    const long nsteps = (a.M + 127) / 128;
    f32x16 dzr[NT], xr[NT];
    auto load_tile = [&](f32x16 (&t)[NT], const float* base, int ld, long step) {
        const long r = min(step * 128 + wave * 32 + j, a.M - 1);              // (rows past the end: clamped here, zeroed / dropped below)
        const float* p = base + r * ld + 4 * half;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + nt * 32 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) t[nt][4 * g + c] = v[c];
            }
    };
    long step = blockIdx.x;
    if (step < nsteps) {
        load_tile(dzr, a.dz, a.ldz, step);
        load_tile(xr, a.x, a.ldx, step);
    }
    f32x16 accw[NACC];                                                        // G dW[32 slab + row][32 b + column]
    zero_tiles<NACC>(accw);
    float bsum = 0.f, dxmax = 0.f;
    float* const my_dz = sdz + wave * W * ES;
    float* const my_x = sx + wave * 32 * XB * ES;
    for (; step < nsteps; step += gridDim.x) {
        const long row = step * 128 + wave * 32 + j;
        const bool row_ok = row < a.M;
        // ---- this wave's rows of dz: zero past the end, per-row power of two for the data GEMM's fp16 split, G dz to LDS
        float mx = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = row_ok ? dzr[nt][i] : 0.f;
                dzr[nt][i] = v;
                mx = fmaxf(mx, fabsf(v));
                my_dz[(nt * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = v * G;
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const unsigned eb = (__float_as_uint(mx) >> 23) & 0xFFu;
        const float sc = __uint_as_float((254u - eb) << 23);                  // scaled row max in [1, 2)
        const float usc = __uint_as_float(eb << 23);                          // (0 for an all-zero row)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) dzr[nt][i] *= sc;
        // ---- data GEMM: 2^11 sc (dz W)
        f32x16 acc[NT];
        zero_tiles<NT>(acc);
        if (!(LB_ABL & 8)) gemm_single<NT>(wl, dzr, acc, lane);
        else for (int nt = 0; nt < NT; ++nt) acc[nt] = dzr[nt];
        // next step's dz rows: into the registers the GEMM has just released, in flight during everything below
        asm volatile("" ::: "memory");
        if (!(LB_ABL & 1) && step + gridDim.x < nsteps) load_tile(dzr, a.dz, a.ldz, step + gridDim.x);
        LB_BARRIER();                                                         // G dz of all four tiles staged
        // ---- this wave's slab of dz (32 features x 128 rows) as fp16 planes, kept for the rounds below; db
        f16x8 ah[4][2], al[4][2];
        if (NT == 4) {                                                        // (NT = 2: read in the wave's own round - 64 registers it does not have)
            float cs = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    cs += stage_operand(sdz + (t * W + slab * 32 + j) * ES + 16 * ks + 8 * half, ah[t][ks], al[t][ks]);
            bsum += cs;
        }
        // ---- dx = (x > 0) * (dz W): formed now (x is about to be staged away), STORED after the rounds - behind the next step's x
        // loads in the memory queue: vmcnt retires in order, so a load consumed at the top of the next step must not sit behind
        // 64 KiB of stores whose acknowledgement takes microseconds
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float t = acc[nt][i] * (1.0f / 2048.0f) * usc;
                if (a.mask) t = xr[nt][i] > 0.f ? t : 0.f;
                acc[nt][i] = t;
                dxmax = fmaxf(dxmax, fabsf(t));
            }
        // ---- dW += dz^T x: NR rounds, XB 32-feature blocks of x of all four tiles per round
#pragma unroll
        for (int rd = 0; rd < NR; ++rd) {
            if (rd) LB_BARRIER();                                             // the previous blocks have been read by everybody
#pragma unroll
            for (int q = 0; q < XB; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    my_x[(q * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = row_ok ? xr[rd * XB + q][i] : 0.f;
            LB_BARRIER();
            f32x16& cw = accw[NT == 4 ? rd : 0];
            if (NT == 4 || xblk == rd)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f16x8 xh[2], xl[2];
                if (NT != 4) {
                    float cs = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) cs += stage_operand(sdz + (t * W + slab * 32 + j) * ES + 16 * ks + 8 * half, ah[t][ks], al[t][ks]);
                    bsum += cs;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if ((LB_ABL & 16) && (rd | t)) { xh[ks] = ah[t][ks], xl[ks] = al[t][ks]; continue; }
                    stage_operand(sx + (t * 32 + j) * ES + 16 * ks + 8 * half, xh[ks], xl[ks]);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (LB_ABL & 4) { cw[0] += (float)xh[ks][0] + (float)xl[ks][1]; continue; }
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t][ks], xh[ks], cw, 0, 0, 0);
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t][ks], xl[ks], cw, 0, 0, 0);
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t][ks], xh[ks], cw, 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        if (!(LB_ABL & 1) && step + gridDim.x < nsteps) load_tile(xr, a.x, a.ldx, step + gridDim.x);
        asm volatile("" ::: "memory");
        if (!(LB_ABL & 2) && row_ok) {
            float* o = a.dx + row * a.lddx + 4 * half;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(o + nt * 32 + 8 * g) = v;
                }
        }
        LB_BARRIER();                                                         // staging free for the next step
    }
    // ---- this workgroup's partial sums (C/D row = dz feature = output of the Linear, column = x feature = its input)
    float* wout = a.wpart + (size_t)blockIdx.x * W * W;
#pragma unroll
    for (int b = 0; b < NACC; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = slab * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            wout[(size_t)n * W + (NT == 4 ? b : xblk) * 32 + j] = accw[b][r] * invG;
        }
    bsum += __shfl_xor(bsum, 32);
    if (half == 0 && (NT == 4 || wave < 2)) a.bpart[(size_t)blockIdx.x * W + slab * 32 + j] = bsum * invG;
    if (a.amax_out) {
#pragma unroll
        for (int o = 32; o; o >>= 1) dxmax = fmaxf(dxmax, __shfl_xor(dxmax, o));
        __syncthreads();
        if (lane == 0) gsc[wave] = dxmax;
        __syncthreads();
        if (threadIdx.x == 0) a.amax_out[blockIdx.x] = fmaxf(fmaxf(gsc[0], gsc[1]), fmaxf(gsc[2], gsc[3]));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Generation 1 (k_linear_bwd_r; profiles/r4/linear_bwd_ablations.txt is why).  Generation 0 above runs one wave per SIMD with 64 KiB of W
// planes + 72 KiB of staging in LDS; its compute (0.65 ms at 2.1 M rows) and its memory traffic (0.70 ms) take turns.  Here: two waves per
// SIMD on 32-row steps (the shapes: see the template's comment below); with NB = 1, W = 128:
//  * W^T lives in REGISTERS: wave w owns in-features 16 w .. + 15 of the data GEMM (dx^T = W^T dz^T, v_mfma_f32_16x16x32_f16; its 4 k tiles
//    x 2 planes of A fragments are 32 registers, read once from the standard pack) - no W in LDS at all;
//  * dz and x are split into fp16 planes ONCE, by the wave that loaded them (4 rows per wave, a lane = 4 features of 2 rows: every load
//    instruction reads whole 512-byte rows), and written row-major [row][position] to LDS: ONE image of G dz and one of x.  The data GEMM's
//    B operand (lane = row) is one ds_read_b128 per k slot and plane; the weight gradient's operands (lane = feature, registers = rows)
//    come out of ds_read_b64_tr_b16 (profiles/r4/tr_read_probe.txt) - no v_perm, no split on the reading side;
//  * the ReLU mask of dx travels as 4 bits per lane and row through LDS (LrBuf::pos): x is read from HBM once;
//  * wave w owns rows 16 w .. + 15 of dW (dz features) x all 128 columns: 8 accumulators of 4 registers;
//  * the step's operands are double buffered: ONE barrier per step; the next step's rows are loaded while this step's GEMMs run.
// G dz serves both GEMMs (generation 0 scaled the data GEMM's rows one by one): a row 2^16 times smaller than the launch maximum still has
// 22 bits, below that its error is bounded by 2^-38 of the launch maximum.  Measured: profiles/r4/linear_bwd_gen1.txt (0.615 ms at 2.1 M
// rows = 5.3 TB/s against 0.90-0.96), traffic_linear_bwd.txt (HBM reads = the algorithmic 2.157 GB), stress_linear_bwd.txt.
#define LR_ST 144                       // halfwords per row: 288 B = 72 banks - the 8 rows of a transposed read's 32-lane group tile the 64 banks
                                        // (3.0 LDS cycles per read, the floor: scripts/dev/tr_bank_probe.cpp), and so do the 16-lane groups of the
                                        // data GEMM's ds_read_b128
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
// Within every block of 16 features the images hold feature 8 hi + 4 kg + c at position 8 kg + 4 hi + c (bits 2 and 3 swapped): the 8 k values
// the standard pack puts in one lane of an A fragment - features 4 kg + c and 8 + 4 kg + c of a 16-feature k slot - are then ONE 16-byte
// read for the data GEMM, and the same image serves the transposed reads (whose lane i is then feature lr_sw(i) of its block).
template <int ST, int LPR>
struct LrBuf {
    unsigned short dz[2][32][ST];       // [plane][row][position] G dz
    unsigned short x[2][32][ST];
    unsigned char pos[32][LPR];         // bit c of [row][j]: x[row][4 j + c] > 0 (the ReLU mask of dx, formed from the fp32 values by the
};                                      // lane that loaded them: re-reading x at the dx positions cost 0.35 GB of HBM reads per 2.1 M rows)
__device__ __forceinline__ int lr_sw(int i) { return (i & 3) | ((i & 4) << 1) | ((i & 8) >> 1); }
typedef __attribute__((address_space(1))) const char* lr_gconst;
typedef __attribute__((address_space(1))) char* lr_gbytes;
// 16 bytes at (wave-uniform base made opaque) + 32-bit lane offset + immediate: see wgrad_private.cuh ldg4; the explicit address space
// keeps the access global_* after the base went through an asm statement
__device__ __forceinline__ f32x4 lr_ldg4(const float* base, unsigned off, int imm) {
    return *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>((lr_gconst)(base) + off + imm);
}
__device__ __forceinline__ void lr_stg4(float* base, unsigned off, int imm, const f32x4& v) {
    *reinterpret_cast<__attribute__((address_space(1))) f32x4*>((lr_gbytes)(base) + off + imm) = v;
}
__device__ __forceinline__ u32x2 lr_tr(const unsigned short* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p)));
}
__device__ __forceinline__ float lr_dot_ones(unsigned w, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), one, acc, false);
}

// NB = blocks of 16 features a wave owns (in-features of the data GEMM = rows of W^T in registers; dz features of the weight gradient).
// <128, 1>: eight waves, one workgroup per CU.  <128, 2>: four waves, two workgroups per CU (W^T 64 + dW 64 registers per wave; the
// workgroups drift apart, so one's loads and barrier waits sit under the other's MFMAs; each wave's LDS reads feed twice the MFMAs).
// <64, 1> (MAgNet[CNN]'s MLPs): four waves, two workgroups per CU; rows of 256 bytes, a load instruction reads four of them; LDS row
// stride 160 B (3.0 cycles per transposed read, conflict-free 16-byte reads).  More than one step of rows in flight was measured (no
// gain, profiles/r4/linear_bwd_gen1.txt) and is not kept.
template <int WIDTH, int NB>
__global__ void __launch_bounds__(WIDTH * 4 / NB, WIDTH == 128 && NB == 1 ? 1 : 2) k_linear_bwd_r(const LinBwdArgs a) {
    constexpr int NW = WIDTH / (16 * NB), KT = WIDTH / 32, XB = WIDTH / 16, ST = WIDTH == 128 ? LR_ST : 80;
    constexpr int LPR = WIDTH / 4, RPI = 64 / LPR;                            // lanes per row of a load instruction, rows per instruction
    constexpr int NLD = 32 / NW / RPI;                                        // load instructions per array, wave and step
    typedef LrBuf<ST, LPR> Buf;
    __shared__ __attribute__((aligned(16))) Buf sb[2];
    __shared__ float gsc[NW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, q4 = lane >> 4;
    float G, invG;
    {
        float m = 0.f;
        for (int i = threadIdx.x; i < a.n_amax; i += 64 * NW) m = fmaxf(m, a.amax_in[i]);
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) gsc[wave] = m;
        __syncthreads();
        m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m = fmaxf(m, gsc[w]);
        const unsigned eb = (__float_as_uint(m) >> 23) & 0xFFu;
        const bool ok = eb >= 14u && eb < 255u;
        G = ok ? __uint_as_float((267u - eb) << 23) : 1.0f;                   // 2^(140-eb): G max|dz| < 2^14
        invG = ok ? __uint_as_float((eb - 13u) << 23) : 1.0f;
        __syncthreads();
    }
    // ---- the wave's NB x 16 in-features of W^T as A fragments (lane = in-feature, k group q4: the pack's k slots (kt, q4 >> 1, q4 & 1))
    f16x8 wh[NB][KT], wlo[NB][KT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int fb = NB * wave + b;                                     // block of 16 features
            const int idx = (((kt * KT + (fb >> 1)) * 2 + (q4 >> 1)) * 2) * 64 + 16 * (fb & 1) + n16 + 32 * (q4 & 1);
            wh[b][kt] = __builtin_bit_cast(f16x8, scale_plane_2048(a.wt[idx]));
            wlo[b][kt] = __builtin_bit_cast(f16x8, a.wt[idx + 64]);
        }
    const long nsteps = (a.M + 31) / 32;
    // the rows of a step this lane loads and splits: prow + RPI k, features 4 l32 .. + 3 of each - every load instruction reads RPI whole
    // rows (2 x 512 bytes / 4 x 256); the four values land at position ppos of the images (see LrBuf)
    const int l32 = lane & (LPR - 1), prow = (32 / NW) * wave + lane / LPR;
    const int ppos = 16 * (l32 >> 2) + 8 * (l32 & 1) + 4 * ((l32 >> 1) & 1);
    f32x4 dv[NLD], xv[NLD];
    auto load_rows = [&](long step) {
        const long last = a.M - 1 - step * 32;
        const float* bz = a.dz + step * 32 * a.ldz;
        const float* bx = a.x + step * 32 * a.ldx;
        asm volatile("" : "+s"(bz), "+s"(bx));
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int rr = (int)min((long)(prow + RPI * k), last);            // (rows past the end: clamped here, zeroed below)
            dv[k] = lr_ldg4(bz, ((unsigned)rr * (unsigned)a.ldz + 4u * l32) * 4u, 0);
            xv[k] = lr_ldg4(bx, ((unsigned)rr * (unsigned)a.ldx + 4u * l32) * 4u, 0);
        }
    };
    f32x4 accw[NB][XB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int xb = 0; xb < XB; ++xb) accw[b][xb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NB], dxmax = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) bsum[b] = 0.f;
    long step = blockIdx.x;
    if (step < nsteps) load_rows(step);
    for (int it = 0; step < nsteps; step += gridDim.x, ++it) {
        Buf& B = sb[it & 1];
        // ---- this lane's NLD x 4 values of dz and of x as fp16 planes, once for the whole workgroup
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const bool pok = step * 32 + prow + RPI * k < a.M;
            const f32x4& d = dv[k];
            const f32x4& x = xv[k];
            u32x2 zh, zl, xh, xl;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned h, l;
                split_f16_plain(pok ? d[2 * c] * G : 0.f, pok ? d[2 * c + 1] * G : 0.f, h, l);
                zh[c] = h, zl[c] = l;
                split_f16_plain(pok ? x[2 * c] : 0.f, pok ? x[2 * c + 1] : 0.f, h, l);
                xh[c] = h, xl[c] = l;
            }
            *reinterpret_cast<u32x2*>(&B.dz[0][prow + RPI * k][ppos]) = zh;
            *reinterpret_cast<u32x2*>(&B.dz[1][prow + RPI * k][ppos]) = zl;
            *reinterpret_cast<u32x2*>(&B.x[0][prow + RPI * k][ppos]) = xh;
            *reinterpret_cast<u32x2*>(&B.x[1][prow + RPI * k][ppos]) = xl;
            if (a.mask)
                B.pos[prow + RPI * k][l32] = (unsigned char)((x[0] > 0.f ? 1 : 0) | (x[1] > 0.f ? 2 : 0) | (x[2] > 0.f ? 4 : 0) | (x[3] > 0.f ? 8 : 0));
        }
        // ---- where this lane's dx go; then the next step's rows
        float* const bdx = a.dx + step * 32 * a.lddx;
        unsigned odx[2];
        bool rok[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            rok[rb] = step * 32 + 16 * rb + n16 < a.M;
            odx[rb] = ((unsigned)(16 * rb + n16) * (unsigned)a.lddx + 16u * NB * wave + 4u * q4) * 4u;
        }
        asm volatile("" ::: "memory");
        if (step + gridDim.x < nsteps) load_rows(step + gridDim.x);
        asm volatile("" ::: "memory");
        wg_barrier();                                                         // the step's planes are in LDS (double buffered: the only barrier)
        // ---- data GEMM: acc[b][rb] = 2^11 G (dz W)^T for rows 16 rb .., in-features 16 (NB wave + b) ..
        f32x4 acc[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b][0] = acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const f16x8 zh = *reinterpret_cast<const f16x8*>(&B.dz[0][16 * rb + n16][32 * kt + 8 * q4]);
                const f16x8 zl = *reinterpret_cast<const f16x8*>(&B.dz[1][16 * rb + n16][32 * kt + 8 * q4]);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[b][kt], zh, acc[b][rb], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[b][kt], zl, acc[b][rb], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[b][kt], zh, acc[b][rb], 0, 0, 0);
            }
        // ---- dW slab += (G dz)^T x over the step's 32 rows (k order: lane group q4 takes rows 4 q4 .. + 3 and 16 + 4 q4 .. + 3), db
        {
            const int r0 = 4 * q4 + (n16 >> 2), c0 = 4 * (n16 & 3);
            f16x8 ah[NB], al[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int f = 16 * (NB * wave + b) + c0;
                const u32x2 a0 = lr_tr(&B.dz[0][r0][f]), a1 = lr_tr(&B.dz[0][16 + r0][f]);
                const u32x2 b0 = lr_tr(&B.dz[1][r0][f]), b1 = lr_tr(&B.dz[1][16 + r0][f]);
                ah[b] = __builtin_bit_cast(f16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
                al[b] = __builtin_bit_cast(f16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
                float cs = lr_dot_ones(a0[0], 0.f);
                cs = lr_dot_ones(a0[1], cs), cs = lr_dot_ones(a1[0], cs), cs = lr_dot_ones(a1[1], cs);
                cs = lr_dot_ones(b0[0], cs), cs = lr_dot_ones(b0[1], cs), cs = lr_dot_ones(b1[0], cs), cs = lr_dot_ones(b1[1], cs);
                bsum[b] += cs;
            }
#pragma unroll
            for (int xb = 0; xb < XB; ++xb) {
                const u32x2 p0 = lr_tr(&B.x[0][r0][16 * xb + c0]), p1 = lr_tr(&B.x[0][16 + r0][16 * xb + c0]);
                const u32x2 q0 = lr_tr(&B.x[1][r0][16 * xb + c0]), q1 = lr_tr(&B.x[1][16 + r0][16 * xb + c0]);
                const f16x8 xh = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p1[0], p1[1]});
                const f16x8 xl = __builtin_bit_cast(f16x8, u32x4{q0[0], q0[1], q1[0], q1[1]});
#pragma unroll
                for (int b = 0; b < NB; ++b) accw[b][xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], xh, accw[b][xb], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NB; ++b) accw[b][xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], xl, accw[b][xb], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < NB; ++b) accw[b][xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], xh, accw[b][xb], 0, 0, 0);
            }
        }
        // ---- dx = (x > 0) * (dz W): C/D row 4 q4 + i = in-feature 16 (NB wave + b) + 4 q4 + i, column n16 = row 16 rb + n16
        {
            float* bo = bdx;
            asm volatile("" : "+s"(bo));
            const float sc = (1.0f / 2048.0f) * invG;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    f32x4 t;
                    const unsigned pm = a.mask ? B.pos[16 * rb + n16][4 * (NB * wave + b) + q4] : 15u;   // in-features 16 (NB wave + b) + 4 q4 ..
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[b][rb][i] * sc;
                        v = (pm >> i) & 1u ? v : 0.f;
                        t[i] = v;
                        dxmax = fmaxf(dxmax, fabsf(v));
                    }
                    if (rok[rb]) lr_stg4(bo, odx[rb], 64 * b, t);
                }
        }
    }
    // ---- this workgroup's partial sums
    float* wout = a.wpart + (size_t)blockIdx.x * WIDTH * WIDTH;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wout[(size_t)(16 * (NB * wave + b) + lr_sw(4 * q4 + i)) * WIDTH + 16 * xb + lr_sw(n16)] = accw[b][xb][i] * invG;
        float bs = bsum[b];
        bs += __shfl_xor(bs, 16);
        bs += __shfl_xor(bs, 32);
        if (q4 == 0) a.bpart[(size_t)blockIdx.x * WIDTH + 16 * (NB * wave + b) + lr_sw(n16)] = bs * invG;
    }
    if (a.amax_out) {
#pragma unroll
        for (int o = 32; o; o >>= 1) dxmax = fmaxf(dxmax, __shfl_xor(dxmax, o));
        __syncthreads();
        if (lane == 0) gsc[wave] = dxmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) m = fmaxf(m, gsc[w]);
            a.amax_out[blockIdx.x] = m;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// TWO consecutive hidden 128 -> 128 Linears walked back in one kernel (k_linear_bwd_r<128, 1>'s construction, eight waves, one workgroup per
// CU): layer A (dz, x1 = its input, W_A) and the layer below it, B (x2 = its input, W_B):
//     dzB = (x1 > 0) * (dz W_A)      dW_A += dz^T x1      db_A += colsum(dz)
//     dx  = (x2 > 0) * (dzB W_B)     dW_B += dzB^T x2     db_B += colsum(dzB)
// dzB never leaves the chip: 4 [M, 128] array passes (dz, x1, x2 in, dx out) instead of 6 - both single-layer launches are HBM-bound (dz + x
// in, dx out at 5.2 TB/s).  dzB goes from the data GEMM's accumulators (lane = 4 consecutive in-features of one row) through the ReLU mask
// and ONE fp16 split straight into the LDS image the second data GEMM and the second weight gradient read.  Its scale G2 cannot come from
// its own maximum (nobody has seen it): it comes from the bound max|dzB| <= max|dz| * max_j sum_i |W_A[i][j]|, formed here from the wave's
// own fragments of W_A^T (the fp16 high plane: within 2^-11 of the sum) - a few bits of headroom more than the measured maximum would
// give, out of the 22 + 16 the split offers (k_linear_bwd_r's header).
struct LinBwd2Args {
    const float* dz;                    // [M][ldz] gradient at layer A's output
    const float* x1;                    // [M][ldx1] layer A's input (= layer B's activated output)
    const float* x2;                    // [M][ldx2] layer B's input
    const u32x4* wtA;                   // fp16x3 planes of W_A^T / W_B^T packed as Linears [128 in][128 out]
    const u32x4* wtB;
    float* dx;                          // [M][lddx]
    float* wpartA;                      // [grid][128][128] per-workgroup partial sums of dW_A / dW_B, [grid][128] of db_A / db_B
    float* bpartA;
    float* wpartB;
    float* bpartB;
    const float* amax_in;               // n_amax block maxima of |dz|
    float* amax_out;                    // [grid] block maxima of |dx| (nullptr: not wanted)
    long M;
    int ldz, ldx1, ldx2, lddx, n_amax;
};
template <int ST, int LPR>
struct Lr2Buf {
    unsigned short dz[2][32][ST];       // [plane][row][position] G dz
    unsigned short x1[2][32][ST];
    unsigned short x2[2][32][ST];
    unsigned char pos1[32][LPR];        // ReLU masks of x1 / x2 (4 bits per lane and row, see LrBuf::pos)
    unsigned char pos2[32][LPR];
};
// WIDTH = 128: eight waves, one workgroup per CU; 64 (MAgNet[CNN]'s chains): four waves, two workgroups per CU (k_linear_bwd_r<64, 1>'s shape)
template <int WIDTH>
__global__ void __launch_bounds__(WIDTH * 4, WIDTH == 128 ? 1 : 2) k_linear_bwd_r2(const LinBwd2Args a) {
    constexpr int NW = WIDTH / 16, KT = WIDTH / 32, XB = WIDTH / 16, ST = WIDTH == 128 ? LR_ST : 80;
    constexpr int LPR = WIDTH / 4, RPI = 64 / LPR, NLD = 32 / NW / RPI;
    typedef Lr2Buf<ST, LPR> Buf;
    typedef unsigned short Img[2][32][ST];
    __shared__ __attribute__((aligned(16))) Buf sb[2];
    __shared__ __attribute__((aligned(16))) Img zbuf;                          // G2 dzB (written behind the first data GEMM; one image:
                                                                              //  its next writer has passed the step's first barrier)
    __shared__ float gsc[NW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, q4 = lane >> 4;
    float G, invG, amax_dz;
    {
        float m = 0.f;
        for (int i = threadIdx.x; i < a.n_amax; i += 64 * NW) m = fmaxf(m, a.amax_in[i]);
#pragma unroll
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) gsc[wave] = m;
        __syncthreads();
        m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m = fmaxf(m, gsc[w]);
        amax_dz = m;
        const unsigned eb = (__float_as_uint(m) >> 23) & 0xFFu;
        const bool ok = eb >= 14u && eb < 255u;
        G = ok ? __uint_as_float((267u - eb) << 23) : 1.0f;                   // 2^(140-eb): G max|dz| < 2^14
        invG = ok ? __uint_as_float((eb - 13u) << 23) : 1.0f;
        G = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(G)));          // (uniform: scalar registers)
        invG = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(invG)));
        __syncthreads();
    }
    // ---- the wave's 16 in-features of W_A^T and W_B^T as A fragments (k_linear_bwd_r); on the way: sum_i |W_A[i][j]| of its in-features
    f16x8 wAh[KT], wAl[KT], wBh[KT], wBl[KT];
    float G2, invG2;
    {
        float colsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int idx = (((kt * KT + (wave >> 1)) * 2 + (q4 >> 1)) * 2) * 64 + 16 * (wave & 1) + n16 + 32 * (q4 & 1);
            const u32x4 hA = a.wtA[idx];
            const f16x8 hv = __builtin_bit_cast(f16x8, hA);
#pragma unroll
            for (int c = 0; c < 8; ++c) colsum += fabsf((float)hv[c]);
            wAh[kt] = __builtin_bit_cast(f16x8, scale_plane_2048(hA));
            wAl[kt] = __builtin_bit_cast(f16x8, a.wtA[idx + 64]);
            wBh[kt] = __builtin_bit_cast(f16x8, scale_plane_2048(a.wtB[idx]));
            wBl[kt] = __builtin_bit_cast(f16x8, a.wtB[idx + 64]);
        }
        colsum += __shfl_xor(colsum, 16);                                     // the four k groups of an in-feature
        colsum += __shfl_xor(colsum, 32);
#pragma unroll
        for (int o = 8; o; o >>= 1) colsum = fmaxf(colsum, __shfl_xor(colsum, o));   // the wave's 16 in-features
        if (lane == 0) gsc[wave] = colsum;
        __syncthreads();
        float c = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) c = fmaxf(c, gsc[w]);
        const float bound = amax_dz * c * 1.01f;                               // >= max|dzB|
        const unsigned eb = (__float_as_uint(bound) >> 23) & 0xFFu;
        const bool ok = eb >= 14u && eb < 255u;
        G2 = ok ? __uint_as_float((267u - eb) << 23) : 1.0f;
        invG2 = ok ? __uint_as_float((eb - 13u) << 23) : 1.0f;
        G2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(G2)));
        invG2 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(invG2)));
        __syncthreads();
    }
    const long nsteps = (a.M + 31) / 32;
    const int l32 = lane & (LPR - 1), prow = (32 / NW) * wave + lane / LPR;
    const int ppos = 16 * (l32 >> 2) + 8 * (l32 & 1) + 4 * ((l32 >> 1) & 1);
    const int zpos = 16 * wave + 8 * (q4 & 1) + 4 * (q4 >> 1);                 // where this lane's four in-features sit in an image row
    f32x4 dv[NLD], x1v[NLD], x2v[NLD];
    auto load_rows = [&](long step) {
        const long last = a.M - 1 - step * 32;
        const float* bz = a.dz + step * 32 * a.ldz;
        const float* b1 = a.x1 + step * 32 * a.ldx1;
        const float* b2 = a.x2 + step * 32 * a.ldx2;
        asm volatile("" : "+s"(bz), "+s"(b1), "+s"(b2));
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int rr = (int)min((long)(prow + RPI * k), last);            // (rows past the end: clamped here, zeroed below)
            dv[k] = lr_ldg4(bz, ((unsigned)rr * (unsigned)a.ldz + 4u * l32) * 4u, 0);
            x1v[k] = lr_ldg4(b1, ((unsigned)rr * (unsigned)a.ldx1 + 4u * l32) * 4u, 0);
            x2v[k] = lr_ldg4(b2, ((unsigned)rr * (unsigned)a.ldx2 + 4u * l32) * 4u, 0);
        }
    };
    f32x4 accwA[XB], accwB[XB];
#pragma unroll
    for (int xb = 0; xb < XB; ++xb) accwA[xb] = accwB[xb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsumA = 0.f, bsumB = 0.f, dxmax = 0.f;
    long step = blockIdx.x;
    if (step < nsteps) load_rows(step);
    // dW slab += Z^T X over the step's 32 rows, colsum(Z) (k_linear_bwd_r's weight-gradient block)
    auto wgrad = [&](const Img& Z, const Img& X, f32x4 (&accw)[XB], float& bsum) {
        const int r0 = 4 * q4 + (n16 >> 2), c0 = 4 * (n16 & 3);
        const int f = 16 * wave + c0;
        const u32x2 a0 = lr_tr(&Z[0][r0][f]), a1 = lr_tr(&Z[0][16 + r0][f]);
        const u32x2 b0 = lr_tr(&Z[1][r0][f]), b1 = lr_tr(&Z[1][16 + r0][f]);
        const f16x8 ah = __builtin_bit_cast(f16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
        const f16x8 al = __builtin_bit_cast(f16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
        float cs = lr_dot_ones(a0[0], 0.f);
        cs = lr_dot_ones(a0[1], cs), cs = lr_dot_ones(a1[0], cs), cs = lr_dot_ones(a1[1], cs);
        cs = lr_dot_ones(b0[0], cs), cs = lr_dot_ones(b0[1], cs), cs = lr_dot_ones(b1[0], cs), cs = lr_dot_ones(b1[1], cs);
        bsum += cs;
#pragma unroll
        for (int xb = 0; xb < XB; ++xb) {
            const u32x2 p0 = lr_tr(&X[0][r0][16 * xb + c0]), p1 = lr_tr(&X[0][16 + r0][16 * xb + c0]);
            const u32x2 q0 = lr_tr(&X[1][r0][16 * xb + c0]), q1 = lr_tr(&X[1][16 + r0][16 * xb + c0]);
            const f16x8 xh = __builtin_bit_cast(f16x8, u32x4{p0[0], p0[1], p1[0], p1[1]});
            const f16x8 xl = __builtin_bit_cast(f16x8, u32x4{q0[0], q0[1], q1[0], q1[1]});
            accw[xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh, accw[xb], 0, 0, 0);
            accw[xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl, accw[xb], 0, 0, 0);
            accw[xb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh, accw[xb], 0, 0, 0);
        }
    };
    // acc[rb] = 2^11 (Z W)^T for rows 16 rb .., in-features 16 wave ..
    auto gemm = [&](const Img& Z, const f16x8 (&wh)[KT], const f16x8 (&wl)[KT], f32x4 (&acc)[2]) {
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const f16x8 zh = *reinterpret_cast<const f16x8*>(&Z[0][16 * rb + n16][32 * kt + 8 * q4]);
                const f16x8 zl = *reinterpret_cast<const f16x8*>(&Z[1][16 * rb + n16][32 * kt + 8 * q4]);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[kt], zh, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kt], zl, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kt], zh, acc[rb], 0, 0, 0);
            }
    };
    for (int it = 0; step < nsteps; step += gridDim.x, ++it) {
        Buf& B = sb[it & 1];
        // ---- this lane's NLD x 4 values of dz, x1, x2 as fp16 planes, once for the whole workgroup
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const bool pok = step * 32 + prow + RPI * k < a.M;
            const f32x4 &d = dv[k], &u = x1v[k], &v = x2v[k];
            u32x2 zh, zl, uh, ul, vh, vl;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned h, l;
                split_f16_plain(pok ? d[2 * c] * G : 0.f, pok ? d[2 * c + 1] * G : 0.f, h, l);
                zh[c] = h, zl[c] = l;
                split_f16_plain(pok ? u[2 * c] : 0.f, pok ? u[2 * c + 1] : 0.f, h, l);
                uh[c] = h, ul[c] = l;
                split_f16_plain(pok ? v[2 * c] : 0.f, pok ? v[2 * c + 1] : 0.f, h, l);
                vh[c] = h, vl[c] = l;
            }
            const int r = prow + RPI * k;
            *reinterpret_cast<u32x2*>(&B.dz[0][r][ppos]) = zh;
            *reinterpret_cast<u32x2*>(&B.dz[1][r][ppos]) = zl;
            *reinterpret_cast<u32x2*>(&B.x1[0][r][ppos]) = uh;
            *reinterpret_cast<u32x2*>(&B.x1[1][r][ppos]) = ul;
            *reinterpret_cast<u32x2*>(&B.x2[0][r][ppos]) = vh;
            *reinterpret_cast<u32x2*>(&B.x2[1][r][ppos]) = vl;
            B.pos1[r][l32] = (unsigned char)((u[0] > 0.f ? 1 : 0) | (u[1] > 0.f ? 2 : 0) | (u[2] > 0.f ? 4 : 0) | (u[3] > 0.f ? 8 : 0));
            B.pos2[r][l32] = (unsigned char)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
        }
        float* const bdx = a.dx + step * 32 * a.lddx;
        unsigned odx[2];
        bool rok[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            rok[rb] = step * 32 + 16 * rb + n16 < a.M;
            odx[rb] = ((unsigned)(16 * rb + n16) * (unsigned)a.lddx + 16u * wave + 4u * q4) * 4u;
        }
        asm volatile("" ::: "memory");
        if (step + gridDim.x < nsteps) load_rows(step + gridDim.x);
        asm volatile("" ::: "memory");
        wg_barrier();                                                         // the step's planes of dz, x1, x2 are in LDS
        // ---- layer A: data GEMM, then dzB = mask1 * (dz W_A) as planes into the image; its weight gradient
        f32x4 acc[2];
        gemm(B.dz, wAh, wAl, acc);
        {
            const float sc = (1.0f / 2048.0f) * invG * G2;                    // 2^11 G (dz W)  ->  G2 dzB
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const unsigned pm = B.pos1[16 * rb + n16][4 * wave + q4];      // in-features 16 wave + 4 q4 ..
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = ((pm >> i) & 1u) && rok[rb] ? acc[rb][i] * sc : 0.f;
                unsigned h0, l0, h1, l1;
                split_f16_plain(t[0], t[1], h0, l0);
                split_f16_plain(t[2], t[3], h1, l1);
                const u32x2 h = {h0, h1}, l = {l0, l1};
                *reinterpret_cast<u32x2*>(&zbuf[0][16 * rb + n16][zpos]) = h;
                *reinterpret_cast<u32x2*>(&zbuf[1][16 * rb + n16][zpos]) = l;
            }
        }
        wgrad(B.dz, B.x1, accwA, bsumA);
        wg_barrier();                                                         // dzB of all 128 in-features staged
        // ---- layer B
        gemm(zbuf, wBh, wBl, acc);
        wgrad(zbuf, B.x2, accwB, bsumB);
        {
            float* bo = bdx;
            asm volatile("" : "+s"(bo));
            const float sc = (1.0f / 2048.0f) * invG2;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const unsigned pm = B.pos2[16 * rb + n16][4 * wave + q4];
                f32x4 t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[rb][i] * sc;
                    v = (pm >> i) & 1u ? v : 0.f;
                    t[i] = v;
                    dxmax = fmaxf(dxmax, fabsf(v));
                }
                if (rok[rb]) lr_stg4(bo, odx[rb], 0, t);
            }
        }
    }
    // ---- this workgroup's partial sums
    float* woA = a.wpartA + (size_t)blockIdx.x * WIDTH * WIDTH;
    float* woB = a.wpartB + (size_t)blockIdx.x * WIDTH * WIDTH;
#pragma unroll
    for (int xb = 0; xb < XB; ++xb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t o = (size_t)(16 * wave + lr_sw(4 * q4 + i)) * WIDTH + 16 * xb + lr_sw(n16);
            woA[o] = accwA[xb][i] * invG;
            woB[o] = accwB[xb][i] * invG2;
        }
    {
        float ba = bsumA, bb = bsumB;
        ba += __shfl_xor(ba, 16), bb += __shfl_xor(bb, 16);
        ba += __shfl_xor(ba, 32), bb += __shfl_xor(bb, 32);
        if (q4 == 0) {
            a.bpartA[(size_t)blockIdx.x * WIDTH + 16 * wave + lr_sw(n16)] = ba * invG;
            a.bpartB[(size_t)blockIdx.x * WIDTH + 16 * wave + lr_sw(n16)] = bb * invG2;
        }
    }
    if (a.amax_out) {
#pragma unroll
        for (int o = 32; o; o >>= 1) dxmax = fmaxf(dxmax, __shfl_xor(dxmax, o));
        __syncthreads();
        if (lane == 0) gsc[wave] = dxmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) m = fmaxf(m, gsc[w]);
            a.amax_out[blockIdx.x] = m;
        }
    }
}

int magnet_launch_reduce_partials2(const float* pa, long na, float* oa, const float* pb, long nb_, float* ob, int nblk, float scale,
                                   int accumulate, void* stream);      // train_kernels.hip: both reductions in one launch
int magnet_launch_reduce_partials2x2(const float* pa, long na, float* oa, const float* pb, long nb_, float* ob, const float* pa2, float* oa2,
                                     const float* pb2, float* ob2, int nblk, float scale, int accumulate, void* stream);

static int lb_grid(long M, int per_cu = 1) {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu = prop.multiProcessorCount;
    }
    const long steps = (M + 127) / 128;
    return (int)(steps < (long)ncu * per_cu ? steps : (long)ncu * per_cu);
}
#define LB_AMAX_BLOCKS 256
// which kernel serves magnet_linear_bwd128_f32 / 64: 1 (default) = k_linear_bwd_r (32-row steps, two waves per SIMD; 128-wide: <128, 2> from
// 65 536 rows, <128, 1> below), 0 = k_linear_bwd<NT>; 2 / 3 = <128, 2> / <128, 1> at every size (A/B runs)
static int g_lb_generation = 1;
extern "C" int magnet_linear_bwd128_generation(int generation) {
    const int prev = g_lb_generation;
    if (generation >= 0 && generation <= 3) g_lb_generation = generation;
    return prev;
}
// 128-wide: four waves with two feature blocks each, two workgroups per CU (11 % faster at 2.1 M rows, 5 % at 218 k) - but twice the
// partial sums to write and reduce, and a launch of a few hundred steps leaves them one step each (18 k rows: 22.9 against 17.2 us)
static bool lb_two_per_cu(long M) { return g_lb_generation == 2 || (g_lb_generation == 1 && M >= 65536); }
static int lb_grid_w(long M, int nt) {                                        // nt = width / 32
    const int per_cu = nt == 2 || lb_two_per_cu(M) ? 2 : 1;                   // (64-wide: three per CU fit and measured 6 % slower)
    if (g_lb_generation == 0) return lb_grid(M, per_cu);
    const long slots = (long)lb_grid(1L << 40) * per_cu, steps = (M + 31) / 32;
    return (int)(steps < slots ? steps : slots);
}
// workspace floats of magnet_linear_bwd{128,64}_f32: per-workgroup partial sums of dW and db, one set of block maxima
extern "C" int magnet_linear_bwd128_workspace(void) { return 2 * lb_grid(1L << 40) * (128 * 128 + 128) + LB_AMAX_BLOCKS; }   // (two workgroups per CU)
template <int NT>
static int launch_linear_bwd(const float* dz, int ldz, const float* x, int ldx, const void* wt_f16, long M, int mask, float* dx, int lddx,
                             float* dW, float* db, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                             float* workspace, void* stream) {
    constexpr int W = 32 * NT;
    (void)hipGetLastError();
    if (M <= 0) return 0;
    if (ldz < W || ldx < W || lddx < W || (ldz | ldx | lddx) % 4) return -2;
    const int grid = lb_grid_w(M, NT);
    LinBwdArgs a;
    a.dz = dz, a.ldz = ldz, a.x = x, a.ldx = ldx, a.wt = reinterpret_cast<const u32x4*>(wt_f16), a.dx = dx, a.lddx = lddx, a.M = M;
    a.mask = mask;
    a.wpart = workspace;
    a.bpart = a.wpart + (size_t)grid * W * W;
    float* own_amax = a.bpart + (size_t)grid * W;
    if (amax_in) {
        a.amax_in = amax_in, a.n_amax = n_amax_in;
    } else {
        if (ldz != W) return -2;                                          // (the maxima pass reads dz as one contiguous array)
        const long n4 = M * (W / 4);
        const int nab = (int)(n4 / 256 + 1 < LB_AMAX_BLOCKS ? n4 / 256 + 1 : LB_AMAX_BLOCKS);
        const int rc = magnet_launch_absmax_partial(dz, n4, own_amax, nab, stream);
        if (rc) return rc;
        a.amax_in = own_amax, a.n_amax = nab;
    }
    a.amax_out = amax_out;
    if (NT == 4 && lb_two_per_cu(M))
        hipLaunchKernelGGL((k_linear_bwd_r<128, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else if (g_lb_generation >= 1)
        hipLaunchKernelGGL((k_linear_bwd_r<W, 1>), dim3(grid), dim3(W * 4), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_linear_bwd<NT>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return magnet_launch_reduce_partials2(a.wpart, (long)W * W, dW, a.bpart, (long)W, db, grid, 1.0f, accumulate, stream);
}
// dx [M][lddx] = (mask ? x > 0 : 1) * (dz W),  dW [W][W] (+)= dz^T x,  db [W] (+)= colsum(dz) for one W -> W Linear (W = 128 / 64);
// wt_f16: fp16x3 planes of W^T packed as a Linear (packing.pack(W.t(), None, "fp16x3"); |w| < 31.98).  amax_in: n_amax_in block maxima
// of |dz| left by the previous layer's call (its amax_out), or nullptr: computed here with one more pass over dz.  amax_out: nullptr or
// room for magnet_linear_bwd128_grid(M) floats.
extern "C" int magnet_linear_bwd128_f32(const float* dz, int ldz, const float* x, int ldx, const void* wt_f16, long M, int mask,
                                        float* dx, int lddx, float* dW, float* db, int accumulate, const float* amax_in,
                                        int n_amax_in, float* amax_out, float* workspace, void* stream) {
    return launch_linear_bwd<4>(dz, ldz, x, ldx, wt_f16, M, mask, dx, lddx, dW, db, accumulate, amax_in, n_amax_in, amax_out, workspace,
                                stream);
}
extern "C" int magnet_linear_bwd64_f32(const float* dz, int ldz, const float* x, int ldx, const void* wt_f16, long M, int mask,
                                       float* dx, int lddx, float* dW, float* db, int accumulate, const float* amax_in,
                                       int n_amax_in, float* amax_out, float* workspace, void* stream) {
    return launch_linear_bwd<2>(dz, ldz, x, ldx, wt_f16, M, mask, dx, lddx, dW, db, accumulate, amax_in, n_amax_in, amax_out, workspace,
                                stream);
}
// Two consecutive hidden 128 -> 128 Linears in one kernel (k_linear_bwd_r2 above): layer A (dz, its input x1, W_A) and the layer below,
// B (its input x2, W_B) - dx [M][lddx] = (x2 > 0) * (((x1 > 0) * (dz W_A)) W_B), dW_A (+)= dz^T x1, db_A (+)= colsum(dz), dW_B (+)= dzB^T x2,
// db_B (+)= colsum(dzB).  wt*_f16, amax_in / amax_out (room for magnet_linear_bwd128_pair_grid(M) floats), workspace
// (magnet_linear_bwd128_workspace floats) as magnet_linear_bwd128_f32.
static int lb_pair_grid(long M, int per_cu = 1) {
    const long slots = (long)lb_grid(1L << 40) * per_cu, steps = (M + 31) / 32;
    return (int)(steps < slots ? steps : slots);
}
extern "C" int magnet_linear_bwd128_pair_grid(long M) { return lb_pair_grid(M); }
extern "C" int magnet_linear_bwd64_pair_grid(long M) { return lb_pair_grid(M, 2); }
template <int W>
static int launch_linear_bwd_pair(const float* dz, int ldz, const float* x1, int ldx1, const float* x2, int ldx2,
                                             const void* wtA_f16, const void* wtB_f16, long M, float* dx, int lddx, float* dWA, float* dbA,
                                             float* dWB, float* dbB, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                                             float* workspace, void* stream) {
    (void)hipGetLastError();
    if (M <= 0) return 0;
    if (ldz < W || ldx1 < W || ldx2 < W || lddx < W || (ldz | ldx1 | ldx2 | lddx) % 4) return -2;
    const int grid = lb_pair_grid(M, W == 128 ? 1 : 2);
    LinBwd2Args a;
    a.dz = dz, a.ldz = ldz, a.x1 = x1, a.ldx1 = ldx1, a.x2 = x2, a.ldx2 = ldx2, a.dx = dx, a.lddx = lddx, a.M = M;
    a.wtA = reinterpret_cast<const u32x4*>(wtA_f16), a.wtB = reinterpret_cast<const u32x4*>(wtB_f16);
    a.wpartA = workspace;
    a.bpartA = a.wpartA + (size_t)grid * W * W;
    a.wpartB = a.bpartA + (size_t)grid * W;
    a.bpartB = a.wpartB + (size_t)grid * W * W;
    float* own_amax = workspace + (size_t)2 * lb_grid(1L << 40) * (128 * 128 + 128);
    if (amax_in) {
        a.amax_in = amax_in, a.n_amax = n_amax_in;
    } else {
        if (ldz != W) return -2;
        const long n4 = M * (W / 4);
        const int nab = (int)(n4 / 256 + 1 < LB_AMAX_BLOCKS ? n4 / 256 + 1 : LB_AMAX_BLOCKS);
        const int rc = magnet_launch_absmax_partial(dz, n4, own_amax, nab, stream);
        if (rc) return rc;
        a.amax_in = own_amax, a.n_amax = nab;
    }
    a.amax_out = amax_out;
    hipLaunchKernelGGL(k_linear_bwd_r2<W>, dim3(grid), dim3(W * 4), 0, (hipStream_t)stream, a);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return magnet_launch_reduce_partials2x2(a.wpartA, (long)W * W, dWA, a.bpartA, (long)W, dbA, a.wpartB, dWB, a.bpartB, dbB, grid, 1.0f,
                                            accumulate, stream);
}
extern "C" int magnet_linear_bwd128_pair_f32(const float* dz, int ldz, const float* x1, int ldx1, const float* x2, int ldx2,
                                             const void* wtA_f16, const void* wtB_f16, long M, float* dx, int lddx, float* dWA, float* dbA,
                                             float* dWB, float* dbB, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                                             float* workspace, void* stream) {
    return launch_linear_bwd_pair<128>(dz, ldz, x1, ldx1, x2, ldx2, wtA_f16, wtB_f16, M, dx, lddx, dWA, dbA, dWB, dbB, accumulate, amax_in,
                                       n_amax_in, amax_out, workspace, stream);
}
extern "C" int magnet_linear_bwd64_pair_f32(const float* dz, int ldz, const float* x1, int ldx1, const float* x2, int ldx2,
                                            const void* wtA_f16, const void* wtB_f16, long M, float* dx, int lddx, float* dWA, float* dbA,
                                            float* dWB, float* dbB, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                                            float* workspace, void* stream) {
    return launch_linear_bwd_pair<64>(dz, ldz, x1, ldx1, x2, ldx2, wtA_f16, wtB_f16, M, dx, lddx, dWA, dbA, dWB, dbB, accumulate, amax_in,
                                      n_amax_in, amax_out, workspace, stream);
}
extern "C" int magnet_linear_bwd128_grid(long M) { return lb_grid_w(M, 4); }
extern "C" int magnet_linear_bwd64_grid(long M) { return lb_grid_w(M, 2); }
