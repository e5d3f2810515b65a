// Weight packing on the device: torch Linear [n_out, k_in] (any strides, optional row / column selection, zero padding)
// -> the fragment orders the GEMM kernels read (include/magnet_hip.h, magnet_amd/packing.py has the same layouts as torch
// index shuffles for CPU tensors).  One launch per Linear instead of ~30 small torch kernels and a host synchronisation
// (the fp16 range check): a training step re-packs every weight it has touched, and for the MAgNet models that was ~60 000 launches
// and ~600 `.item()` round trips per step.
//
// Source element of padded position (r, c):  r' = row_src ? row_src[r] : (r < n ? r : -1),  c' likewise; value = w[r' sr + c' sc] or 0.
#include "common.cuh"

namespace {
enum { PACK_FP32 = 0, PACK_FP16X2 = 1, PACK_BF16X3 = 2, PACK_BF16X1 = 3, PACK_FP16X2S = 4 };   // 4: fp16x3s - high plane and bias times 2^11

struct PackSrc {
    const float* w;
    long sr, sc;
    int n, k;
    const int32_t* row_src;
    const int32_t* col_src;
};

__device__ __forceinline__ float pack_fetch(const PackSrc& s, int r, int c) {
    const int rr = s.row_src ? s.row_src[r] : (r < s.n ? r : -1);
    const int cc = s.col_src ? s.col_src[c] : (c < s.k ? c : -1);
    return (rr >= 0 && cc >= 0) ? s.w[(long)rr * s.sr + (long)cc * s.sc] : 0.f;
}

__device__ __forceinline__ unsigned short f2h_bits(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }   // RNE
__device__ __forceinline__ unsigned short bf16_rne_bits(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// 16-bit planes [kt][nt][m][NP][lane 64][8]: element e of lane (i = lane & 31, h = lane >> 5), k slot m  <->
//   W[nt*32 + i][kt*32 + 16 m + 8 (e >> 2) + 4 h + (e & 3)]
template <int MODE>
__global__ void __launch_bounds__(256) k_pack16(PackSrc s, int NT, int KT, unsigned short* __restrict__ out, int* __restrict__ status,
                                                int range_bit) {
    constexpr int NP = (MODE == PACK_FP16X2 || MODE == PACK_FP16X2S) ? 2 : (MODE == PACK_BF16X3 ? 3 : 1);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)KT * NT * 2 * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63), m = (int)((t >> 6) & 1);
    const long tile = t >> 7;
    const int nt = (int)(tile % NT), kt = (int)(tile / NT);
    const int i = lane & 31, h = lane >> 5;
    unsigned short pl[NP][8];
    bool bad = false, big = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = pack_fetch(s, nt * 32 + i, kt * 32 + 16 * m + 8 * (e >> 2) + 4 * h + (e & 3));
        if (MODE == PACK_FP16X2) {
            bad |= !(fabsf(v) < 65504.0f);                                  // also catches NaN / inf
            big |= !(fabsf(v) < 31.98f);                                    // fits these planes, not the 2^11-scaled ones of the training kernels
            const _Float16 hv = (_Float16)v;
            pl[0][e] = __builtin_bit_cast(unsigned short, hv);
            pl[1][e] = f2h_bits((v - (float)hv) * 2048.0f);
        } else if (MODE == PACK_FP16X2S) {
            bad |= !(fabsf(v) < 31.98f);                                    // 2^11 hw has to stay below the fp16 maximum
            const _Float16 hv = (_Float16)v;
            pl[0][e] = f2h_bits((float)hv * 2048.0f);                       // exact: a power of two, in range
            pl[1][e] = f2h_bits((v - (float)hv) * 2048.0f);
        } else if (MODE == PACK_BF16X3) {
            const unsigned u = __builtin_bit_cast(unsigned, v);
            const float hi = __builtin_bit_cast(float, u & 0xFFFF0000u);
            const float r1 = v - hi;
            const unsigned u1 = __builtin_bit_cast(unsigned, r1);
            const float mid = __builtin_bit_cast(float, u1 & 0xFFFF0000u);
            const float lo = r1 - mid;
            pl[0][e] = (unsigned short)(u >> 16);
            pl[1][e] = (unsigned short)(u1 >> 16);
            pl[2][e] = (unsigned short)(__builtin_bit_cast(unsigned, lo) >> 16);
        } else {
            pl[0][e] = bf16_rne_bits(v);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        u32x4 w4;
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = (unsigned)pl[p][2 * q] | ((unsigned)pl[p][2 * q + 1] << 16);
        *reinterpret_cast<u32x4*>(out + ((((tile * 2 + m) * NP + p) * 64 + lane) * 8)) = w4;
    }
    if ((MODE == PACK_FP16X2 || MODE == PACK_FP16X2S) && status && bad) atomicOr(status, range_bit);
    // range_bit << 1: a weight in [31.98, 65504) - harmless for this pack, but the one-kernel backward of a hidden Linear and the edge
    // backward keep 2^11 x the fp16 high plane of the SAME weight; a training step reads this before its backward runs (packing.bad_weights)
    // (square 128- / 64-wide Linears only: message_net_2 and the hidden Linears of the chains are what those kernels take - a first layer
    //  or a 32-wide output Linear with such a weight must not widen the whole process)
    else if (MODE == PACK_FP16X2 && status && big && s.n == s.k && (s.n == 128 || s.n == 64)) atomicOr(status, range_bit << 1);
}

// fp32 fragments [kt][nt][g][lane 64][4]:  W[nt*32 + (lane & 31)][kt*32 + 8 g + 4 (lane >> 5) + c]
__global__ void __launch_bounds__(256) k_pack32(PackSrc s, int NT, int KT, float* __restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)KT * NT * 4 * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63), g = (int)((t >> 6) & 3);
    const long tile = t >> 8;
    const int nt = (int)(tile % NT), kt = (int)(tile / NT);
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = pack_fetch(s, nt * 32 + (lane & 31), kt * 32 + 8 * g + 4 * (lane >> 5) + c);
    *reinterpret_cast<f32x4*>(out + t * 4) = v;
}

__global__ void k_pad_bias(const float* __restrict__ b, int n, int np, float scale, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np) out[i] = (b && i < n) ? b[i] * scale : 0.f;
}
}  // namespace

extern "C" int magnet_pack_linear_f32(const float* w, long stride_r, long stride_c, int n, int k, const int32_t* row_src,
                                      const int32_t* col_src, const float* bias, int n_bias, int np, int kp, int mode, void* out,
                                      float* bias_out, int* status, int range_bit, void* stream) {
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls (e.g. torch's device probing)
    if (np <= 0 || kp <= 0 || np % 32 || kp % 32) return -2;
    const PackSrc s = {w, stride_r, stride_c, n, k, row_src, col_src};
    const int NT = np / 32, KT = kp / 32;
    hipStream_t st = (hipStream_t)stream;
    if (mode == PACK_FP32) {
        const long total = (long)KT * NT * 4 * 64;
        hipLaunchKernelGGL(k_pack32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s, NT, KT, (float*)out);
    } else {
        const long total = (long)KT * NT * 2 * 64;
        const dim3 g((unsigned)((total + 255) / 256)), b(256);
        unsigned short* o = (unsigned short*)out;
        if (mode == PACK_FP16X2) hipLaunchKernelGGL(k_pack16<PACK_FP16X2>, g, b, 0, st, s, NT, KT, o, status, range_bit);
        else if (mode == PACK_BF16X3) hipLaunchKernelGGL(k_pack16<PACK_BF16X3>, g, b, 0, st, s, NT, KT, o, status, range_bit);
        else if (mode == PACK_BF16X1) hipLaunchKernelGGL(k_pack16<PACK_BF16X1>, g, b, 0, st, s, NT, KT, o, status, range_bit);
        else if (mode == PACK_FP16X2S) hipLaunchKernelGGL(k_pack16<PACK_FP16X2S>, g, b, 0, st, s, NT, KT, o, status, range_bit);
        else return -2;
    }
    if (bias_out) hipLaunchKernelGGL(k_pad_bias, dim3((np + 255) / 256), dim3(256), 0, st, bias, n_bias, np, mode == PACK_FP16X2S ? 2048.0f : 1.0f, bias_out);
    return (int)hipGetLastError();
}
