// Fused MPNN edge kernel, generation 10: WAVE-SPECIALISED.  Same op as generations 6-9,
//
//   m_e = Swish(W2 * Swish(A[i] + Bm[j]) + b2),  agg[i] = mean_e m_e          (models/mpnn.py:73-79,46)
//
// but a different division of labour.  The ablations of generation 9 (profiles/r3/edge9_ablations.txt) showed that in
// every earlier generation the 96 MFMAs of a 32-edge tile hide completely behind the other ~1 000 instructions of the wave that issues
// them, and that this instruction stream retires one instruction per 7-9 cycles whatever its length: one or two fat waves (210-450
// registers) per SIMD cannot keep the VALU busy through their own dependency chains, LDS round trips and gather latencies.  Here the
// tile's work is split over EIGHT lean waves (<= 128 registers, two workgroups = 16 waves per CU, four per SIMD) with three roles,
// pipelined over consecutive tiles through LDS:
//
//   P  (2 waves, 16 edges each)   gather: every load instruction reads ONE WHOLE 512-byte row (lane = feature pair, the row index is
//                                 wave-uniform -> scalar base + lane offset; one cache-line access per line instead of four), first Swish,
//                                 the two fp16 pieces (v_fma_mix), written to the X tile in LDS.  The row registers are a rolling ring: the
//                                 rows of the same edge slot of the NEXT tile are requested as soon as a slot is consumed.
//   M  (4 waves, one 32-feature block each)   the GEMM transposed as in generation 9 (rows = edges): X fragments from LDS, the wave's slice of
//                                 W2 STAYS IN REGISTERS for the whole kernel (64 registers, no LDS weight traffic at all), one accumulator
//                                 (weight planes pre-scaled by 2^11, unscaled low activation piece), raw result to the Y tile in LDS.
//   E  (2 waves, 64 features each)   epilogue of generations 7 / 8: lane = feature, reads the Y tile edge by edge, bias, second Swish, running
//                                 sum carried across the tiles of the workgroup's range, run ends (wave-uniform) leave with a coalesced store.
//
// Step i of a workgroup: P fills X[i & 1] with tile i, M turns X[(i-1) & 1] into Y[(i-1) & 1], E consumes Y[(i-2) & 1]; one
// __syncthreads per step.  The second workgroup on the CU runs its own pipeline and fills the barrier waits.
#include "common.cuh"
#include "edge_tile.cuh"

typedef unsigned int u32;

#ifdef MAGNET_DEV_VARIANTS
__device__ long long g_ws_trace[32];       // development builds: cycle sums per phase of workgroup 0 (P: 0-7, M: 8-15, E: 16-23)
#define WS_T(slot) do { if (blockIdx.x == 0 && lane == 0) { const long long now_ = clock64(); g_ws_trace[slot] += now_ - tprev_; tprev_ = now_; } } while (0)
#define WS_T0() long long tprev_ = clock64()
#else
#define WS_T(slot) do { } while (0)
#define WS_T0() do { } while (0)
#endif

namespace {

constexpr int WS_XS = 68;                  // dwords per X row: 64 (= 128 fp16) + 4 padding -> conflict-free ds_read_b128 of the fragments
constexpr int WS_YS = 128;                 // dwords per Y row
constexpr float WS_SCALE = 2048.0f;        // every MFMA product carries 2^11 (host-scaled weight planes, packing.pack_edge32t_fp16x2)

__device__ __forceinline__ f32x16 ws_mfma(const u32x4& x, const u32x4& w, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, w), c, 0, 0, 0);
}

// Swish of the pair (v0, v1) -> one word of the high plane and one of the low plane (h = fp16(v r), l = fp16(v r - h))
template <bool L2E>
__device__ __forceinline__ void ws_swish_split(float v0, float v1, u32& hw, u32& lw) {
    const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -v0 : -v0 * MAGNET_LOG2E));
    const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(L2E ? -v1 : -v1 * MAGNET_LOG2E));
    u32 h, l;
    asm(MAGNET_TRANS_HEAD "v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "v"(r0));
    asm(MAGNET_TRANS_HEAD "v_fma_mixhi_f16 %0, %1, %2, 0" MAGNET_MIX_TAIL : "+v"(h) : "v"(v1), "v"(r1));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v0), "v"(r0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" MAGNET_MIX_TAIL : "+v"(l) : "v"(v1), "v"(r1), "v"(h));
    hw = h, lw = l;
}

__device__ __forceinline__ void ws_emit(float sv, int ti, int rs, int re, int e_lo, int e_hi, int range, float deg_scale,
                                        float* __restrict__ agg, float* __restrict__ partials, int fcol) {
    float* dst;
    if (rs >= e_lo && re <= e_hi) {
        sv = sv / ((float)(re - rs) * deg_scale);
        dst = agg + (size_t)ti * 128;
    } else {
        dst = partials + ((size_t)range * 2 + (rs < e_lo ? 0 : 1)) * 128;
    }
    dst[fcol] = sv;
}

}  // namespace

// ABL (development builds, results INVALID): 1 = the P waves only pass the barriers, 2 = the M waves, 4 = the E waves
template <bool L2E, int ABL = 0>
__global__ void __launch_bounds__(512, 4) k_mpnn_edge_ws(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         const int32_t* __restrict__ src, const int32_t* __restrict__ tgt,
                                                         const int32_t* __restrict__ rowptr, int E,
                                                         const u32x4* __restrict__ w2p /* [8 s][4 nt][2 planes][64] */,
                                                         const float* __restrict__ b2, float* __restrict__ agg,
                                                         float* __restrict__ partials) {
    constexpr float deg_scale = L2E ? MAGNET_LOG2E : 1.0f;
    __shared__ __attribute__((aligned(16))) u32 xb[2][2][32 * WS_XS];     // [buffer][plane][edge row][k pair]      34 KiB
    __shared__ __attribute__((aligned(16))) float yb[2][32 * WS_YS];      // [buffer][edge row][feature]            32 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform role index in an SGPR
    const int ntiles = (E + 31) / 32;
    const int nwg = gridDim.x;
    const int nb8 = nwg / 8;               // XCD-aware contiguous ranges, see k_mpnn_edge6 (one range per WORKGROUP here)
    const int range = (nwg % 8 == 0) ? (blockIdx.x % 8) * nb8 + blockIdx.x / 8 : blockIdx.x;
    const int tpw = (ntiles + nwg - 1) / nwg;
    const int t0 = range * tpw;
    const int t1 = min(t0 + tpw, ntiles);
    const int n = t1 - t0;                  // tiles of this workgroup (may be <= 0: then every role just leaves)
    if (n <= 0) return;
    const int e_lo = t0 * 32, e_hi = min(t1 * 32, E);
    const int nsteps = n + 2;

    if (wave < 4) {
        // ------------------------------------------------------------------ M: GEMM of one 32-feature block
        const int nt = wave, i = lane & 31, h = lane >> 5;
        u32x4 wreg[8][2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            wreg[s][0] = w2p[((s * 4 + nt) * 2 + 0) * 64 + lane];
            wreg[s][1] = w2p[((s * 4 + nt) * 2 + 1) * 64 + lane];
        }
        WS_T0();
        for (int st = 0; st < nsteps; ++st) {
            WS_T(8);                               // barrier wait + loop
            if (st >= 1 && st <= n && !(ABL & 2)) {
                const int b = (st - 1) & 1;
                const u32* xh = &xb[b][0][i * WS_XS + 4 * h];            // fragment of slot s: + 8 s dwords (16 bytes = 8 fp16)
                const u32* xl = &xb[b][1][i * WS_XS + 4 * h];
                f32x16 acc, acc2;                 // even / odd k slots: two independent accumulation chains
                u32x4 fh[2], fl[2];
                fh[0] = *reinterpret_cast<const u32x4*>(xh);
                fl[0] = *reinterpret_cast<const u32x4*>(xl);
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                fh[1] = *reinterpret_cast<const u32x4*>(xh + 8);
                fl[1] = *reinterpret_cast<const u32x4*>(xl + 8);
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {      // slots 2 sp (-> acc) and 2 sp + 1 (-> acc2), their MFMAs alternating
                    const int s0 = 2 * sp, s1 = 2 * sp + 1;
                    const u32x4 h0 = fh[0], l0 = fl[0], h1 = fh[1], l1 = fl[1];
                    if (sp + 1 < 4) {
                        fh[0] = *reinterpret_cast<const u32x4*>(xh + 8 * (s0 + 2));
                        fl[0] = *reinterpret_cast<const u32x4*>(xl + 8 * (s0 + 2));
                        fh[1] = *reinterpret_cast<const u32x4*>(xh + 8 * (s1 + 2));
                        fl[1] = *reinterpret_cast<const u32x4*>(xl + 8 * (s1 + 2));
                    }
                    acc = ws_mfma(h0, wreg[s0][1], sp == 0 ? z : acc);
                    acc2 = ws_mfma(h1, wreg[s1][1], sp == 0 ? z : acc2);
                    acc = ws_mfma(h0, wreg[s0][0], acc);
                    acc2 = ws_mfma(h1, wreg[s1][0], acc2);
                    acc = ws_mfma(l0, wreg[s0][0], acc);
                    acc2 = ws_mfma(l1, wreg[s1][0], acc2);
                }
                WS_T(9);                           // fragment reads + 24 MFMAs issued
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
                float* y = &yb[b][(4 * h) * WS_YS + 32 * nt + i];         // register r <-> edge (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
                for (int r = 0; r < 16; ++r) y[((r & 3) + 8 * (r >> 2)) * WS_YS] = acc[r];
                WS_T(10);                          // accumulator sum (waits for the MFMAs) + Y stores
            }
            __syncthreads();
        }
    } else if (wave < 6) {
        // ------------------------------------------------------------------ P: gather + first Swish + split, 16 edges per wave
        const int q = wave - 4;
        f32x2 ra[16], rb[16];                  // rolling ring: the A / Bm row of edge slot k (two features per lane)
        // the indices of a tile's 16 edges arrive one step ahead as a vector (lane k <-> edge slot k); v_readlane turns them into the
        // wave-uniform row bases when the rows are requested - a scalar load per request would stall every request for its latency
        // ... as BYTE OFFSETS of the rows: a request is then two v_readlane + two buffer loads (soffset = row offset, voffset = 8 lane) -
        // no 64-bit scalar address arithmetic between the readlane and the load
        auto indices = [&](int t, int& vt, int& vs) {
            const int e = min(t * 32 + 16 * q + (lane & 15), E - 1);
            vt = tgt[e] << 9;                                            // row * 512 bytes (n_rows * 512 < 2^32 is checked by the launcher)
            vs = src[e] << 9;
        };
        const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0x7FFFFFFF, 0x00020000);
        const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bm), 0, 0x7FFFFFFF, 0x00020000);
        auto request = [&](int vt, int vs, int k) {
            const int oa = __builtin_amdgcn_readlane(vt, k), ob = __builtin_amdgcn_readlane(vs, k);
            ra[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsA, 8 * lane, oa, 0));
            rb[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsB, 8 * lane, ob, 0));
        };
        int vt0, vs0, vt1, vs1;
        indices(t0, vt0, vs0);
        indices(min(t0 + 1, t1 - 1), vt1, vs1);
#pragma unroll
        for (int k = 0; k < 16; ++k) request(vt0, vs0, k);
        WS_T0();
        for (int st = 0; st < nsteps; ++st) {
            WS_T(0);                               // barrier wait + loop
            if (st < n && !(ABL & 1)) {
                const int b = st & 1;
                int vt2, vs2;
                indices(min(t0 + st + 2, t1 - 1), vt2, vs2);             // needed by the NEXT step's requests
                u32* xh = &xb[b][0][(16 * q) * WS_XS + lane];
                u32* xl = &xb[b][1][(16 * q) * WS_XS + lane];
                // eight edge slots at a time, stage by stage: inside a stage no instruction depends on another one (a lone wave
                // retires a dependent chain at ~10 cycles per instruction - the role ablations of DESIGN.md 10.5)
#pragma unroll
                for (int kb = 0; kb < 16; kb += 8) {
                    float v[16], r[16];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        v[2 * k] = ra[kb + k][0] + rb[kb + k][0];
                        v[2 * k + 1] = ra[kb + k][1] + rb[kb + k][1];
                    }
#pragma unroll
                    for (int c = 0; c < 16; ++c) r[c] = __builtin_amdgcn_exp2f(L2E ? -v[c] : -v[c] * MAGNET_LOG2E);
#pragma unroll
                    for (int c = 0; c < 16; ++c) r[c] = 1.0f + r[c];
#pragma unroll
                    for (int c = 0; c < 16; ++c) r[c] = __builtin_amdgcn_rcpf(r[c]);
                    WS_T(1 + (kb >> 3) * 4);           // wait for the rows + add / exp / add / rcp
                    u32 hw[8], lw[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        asm(MAGNET_TRANS_HEAD "v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hw[k]) : "v"(v[2 * k]), "v"(r[2 * k]));
                        asm(MAGNET_TRANS_HEAD "v_fma_mixhi_f16 %0, %1, %2, 0" MAGNET_MIX_TAIL : "+v"(hw[k]) : "v"(v[2 * k + 1]), "v"(r[2 * k + 1]));
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lw[k]) : "v"(v[2 * k]), "v"(r[2 * k]), "v"(hw[k]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" MAGNET_MIX_TAIL
                            : "+v"(lw[k])
                            : "v"(v[2 * k + 1]), "v"(r[2 * k + 1]), "v"(hw[k]));
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        xh[(kb + k) * WS_XS] = hw[k];
                        xl[(kb + k) * WS_XS] = lw[k];
                    }
                    WS_T(2 + (kb >> 3) * 4);           // pieces + LDS writes
#pragma unroll
                    for (int k = 0; k < 8; ++k) request(vt1, vs1, kb + k);     // the same slots of the next tile (clamped at the end)
                    WS_T(3 + (kb >> 3) * 4);           // requests
                }
                vt1 = vt2, vs1 = vs2;
            }
            __syncthreads();
        }
    } else {
        // ------------------------------------------------------------------ E: bias, second Swish, running sum, run ends (64 features per wave)
        const int p = wave - 6, j = lane & 31;
        const int fcol = 64 * p + lane;
        const float bias = b2[fcol] * deg_scale;
        float sum = 0.f;
        // meta of a tile, pipelined over two steps so that neither of its two dependent load round trips is waited for: the indices
        // of tile k are requested in step k (E works on tile k in step k + 2), its row bounds in step k + 1
        RunMeta md;
        TileIdx ixa = tile_idx<32>(t0, e_hi, src, tgt, j);               // tile t0: requested here, bounds below
        TileIdx ixb = tile_idx<32>(min(t0 + 1, t1 - 1), e_hi, src, tgt, j);
        int rsa = rowptr[ixa.ti], rea = rowptr[ixa.ti + 1];
        WS_T0();
        for (int st = 0; st < nsteps; ++st) {
            WS_T(16);                              // barrier wait + loop
            if (st >= 2 && !(ABL & 4)) {
                md.ti = ixa.ti, md.rs = rsa, md.re = rea;
                md.ends = (unsigned)__builtin_amdgcn_ballot_w64(ixa.e_ok && ixa.tnext != ixa.ti);   // both halves agree: low word
                ixa = ixb;                                                  // tile st - 1 (its indices arrived a step ago) ...
                rsa = rowptr[ixa.ti], rea = rowptr[ixa.ti + 1];            // ... its row bounds are used in the next step
                ixb = tile_idx<32>(min(t0 + st, t1 - 1), e_hi, src, tgt, j);   // tile st: used in step st + 2
                const int b = st & 1;                                       // (st - 2) & 1
                const float* y = &yb[b][fcol];
                WS_T(17);                          // meta
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {      // eight edges at a time, stage by stage (see the P waves)
                    float xv[8], rv[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) xv[c] = fmaf(y[(8 * g8 + c) * WS_YS], 1.0f / WS_SCALE, bias);
#pragma unroll
                    for (int c = 0; c < 8; ++c) rv[c] = __builtin_amdgcn_exp2f(L2E ? -xv[c] : -xv[c] * MAGNET_LOG2E);
#pragma unroll
                    for (int c = 0; c < 8; ++c) rv[c] = 1.0f + rv[c];
#pragma unroll
                    for (int c = 0; c < 8; ++c) rv[c] = __builtin_amdgcn_rcpf(rv[c]);
                    const unsigned m8 = (md.ends >> (8 * g8)) & 255u;
                    if (m8 == 0) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) sum = fmaf(xv[c], rv[c], sum);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            sum = fmaf(xv[c], rv[c], sum);
                            if (m8 & (1u << c)) {
                                const int e = 8 * g8 + c;
                                ws_emit(sum, __builtin_amdgcn_readlane(md.ti, e), __builtin_amdgcn_readlane(md.rs, e),
                                        __builtin_amdgcn_readlane(md.re, e), e_lo, e_hi, range, deg_scale, agg, partials, fcol);
                                sum = 0.f;
                            }
                        }
                    }
                }
                WS_T(18);                          // four groups of eight edges
            }
            __syncthreads();
        }
    }
}

#ifdef MAGNET_DEV_VARIANTS
extern "C" int magnet_debug_ws_trace(long long* out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_ws_trace), sizeof(long long) * 32) != hipSuccess) return -1;
    if (reset) {
        long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_trace), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
int g_edge_ws_ablation = 0;        // development builds only: magnet_debug_set_variant(3, flags)
#endif

// w2_t32: the planes of packing.pack_edge32t_fp16x2 (both pre-scaled by 2^11).  partials: ceil(n_edges / 32) * 2 * 128 floats at least.
extern "C" int magnet_mpnn_edge_fp16x3_ws_f32(const float* A, const float* Bm, const int32_t* src, const int32_t* tgt,
                                              const int32_t* rowptr, int n_edges, int n_rows, const void* w2_t32, const float* b2,
                                              float* agg, float* partials, int a_bm_log2e_scaled, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev;
        hipGetDevice(&dev);
        hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount;
    }
    if ((long)n_rows * 512 >= (1L << 31)) return -2;                // the gather waves address rows by 32-bit byte offsets
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (n_edges + 31) / 32;
    const int nwg = tiles > 0 ? min(tiles, 2 * ncu) : 0;          // two workgroups (16 waves) per CU
    if (nwg > 0) {
        const dim3 g(nwg), b(512);
        const u32x4* ws = reinterpret_cast<const u32x4*>(w2_t32);
#ifdef MAGNET_DEV_VARIANTS
#define WS_ABL_CASE(v) case v: hipLaunchKernelGGL((k_mpnn_edge_ws<true, v>), g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials); break;
        if (g_edge_ws_ablation && a_bm_log2e_scaled) {
            switch (g_edge_ws_ablation) {
                WS_ABL_CASE(1) WS_ABL_CASE(2) WS_ABL_CASE(3) WS_ABL_CASE(4) WS_ABL_CASE(5) WS_ABL_CASE(6) WS_ABL_CASE(7)
                default: return -2;
            }
        } else
#endif
        if (a_bm_log2e_scaled) hipLaunchKernelGGL(k_mpnn_edge_ws<true>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
        else hipLaunchKernelGGL(k_mpnn_edge_ws<false>, g, b, 0, s, A, Bm, src, tgt, rowptr, n_edges, ws, b2, agg, partials);
    }
    const int tpw = nwg ? (tiles + nwg - 1) / nwg : 1;               // as computed inside the kernel
    return magnet_launch_edge_range_fixup(tgt, rowptr, n_rows, n_edges, tpw, 5, nwg, a_bm_log2e_scaled ? MAGNET_LOG2E : 1.0f,
                                          partials, agg, stream);
}
