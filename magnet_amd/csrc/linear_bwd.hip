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
        gemm_single<NT>(wl, dzr, acc, lane);
        // next step's dz rows: into the registers the GEMM has just released, in flight during everything below
        asm volatile("" ::: "memory");
        if (step + gridDim.x < nsteps) load_tile(dzr, a.dz, a.ldz, step + gridDim.x);
        wg_barrier();                                                         // G dz of all four tiles staged
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
            if (rd) wg_barrier();                                             // the previous blocks have been read by everybody
#pragma unroll
            for (int q = 0; q < XB; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    my_x[(q * 32 + 8 * (i >> 2) + 4 * half + (i & 3)) * ES + j] = row_ok ? xr[rd * XB + q][i] : 0.f;
            wg_barrier();
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
                for (int ks = 0; ks < 2; ++ks) stage_operand(sx + (t * 32 + j) * ES + 16 * ks + 8 * half, xh[ks], xl[ks]);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t][ks], xh[ks], cw, 0, 0, 0);
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t][ks], xl[ks], cw, 0, 0, 0);
                    cw = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t][ks], xh[ks], cw, 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        if (step + gridDim.x < nsteps) load_tile(xr, a.x, a.ldx, step + gridDim.x);
        asm volatile("" ::: "memory");
        if (row_ok) {
            float* o = a.dx + row * a.lddx + 4 * half;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(o + nt * 32 + 8 * g) = v;
                }
        }
        wg_barrier();                                                         // staging free for the next step
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

int magnet_launch_reduce_partials2(const float* pa, long na, float* oa, const float* pb, long nb_, float* ob, int nblk, float scale,
                                   int accumulate, void* stream);      // train_kernels.hip: both reductions in one launch

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
// workspace floats of magnet_linear_bwd{128,64}_f32: per-workgroup partial sums of dW and db, one set of block maxima
extern "C" int magnet_linear_bwd128_workspace(void) { return lb_grid(1L << 40) * (128 * 128 + 128) + LB_AMAX_BLOCKS; }
template <int NT>
static int launch_linear_bwd(const float* dz, int ldz, const float* x, int ldx, const void* wt_f16, long M, int mask, float* dx, int lddx,
                             float* dW, float* db, int accumulate, const float* amax_in, int n_amax_in, float* amax_out,
                             float* workspace, void* stream) {
    constexpr int W = 32 * NT;
    (void)hipGetLastError();
    if (M <= 0) return 0;
    if (ldz < W || ldx < W || lddx < W || (ldz | ldx | lddx) % 4) return -2;
    const int grid = lb_grid(M, NT == 2 ? 2 : 1);
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
extern "C" int magnet_linear_bwd128_grid(long M) { return lb_grid(M); }
extern "C" int magnet_linear_bwd64_grid(long M) { return lb_grid(M, 2); }
