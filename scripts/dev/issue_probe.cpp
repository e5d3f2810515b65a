// Does a second wave on the same SIMD raise the instruction issue rate?  One workgroup per CU, 1 / 2 / 4 waves per SIMD,
// every wave does the same work: if the time stays the same when the waves double, throughput doubled.
//   kind 0: independent v_fma_f32            kind 1: v_exp_f32 (+ mul)          kind 2: Swish  x * rcp(1 + exp(-x))
//   kind 3: MFMA 32x32x16 f16 + 8 fma per MFMA (one instruction stream)        kind 4: MFMA only
//   kind 5: one MFMA group of the edge kernel: 3 MFMA + 20 fma + 4 exp + 4 rcp, interleaved
//   kind 6: kind 5 with the MFMA A operands read from LDS (2 ds_read_b128 per group, requested one group ahead)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int TPB>
__global__ void __launch_bounds__(TPB) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) u32x4 wl[KIND == 6 ? 4096 : 1];
    if (KIND == 6) {
        for (int i = threadIdx.x; i < 4096; i += TPB) wl[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
        __syncthreads();
    }
    u32x4 wa = {0, 0, 0, 0}, wb = {0, 0, 0, 0};
    f16x8 x, y;
    for (int c = 0; c < 8; ++c) { x[c] = (_Float16)(threadIdx.x * 0.01f + c); y[c] = (_Float16)(c * 0.3f); }
    f32x16 c0 = {0}, c1 = {0};
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i * 0.1f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 3 || KIND == 4) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
            if (KIND == 3) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaf(v[q], b, a);
            }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0);
            if (KIND == 3) {
#pragma unroll
                for (int q = 8; q < 16; ++q) v[q] = fmaf(v[q], b, a);
            }
        } else if (KIND == 6) {
            const u32x4 na = wl[((i * 2) & 63) * 64 + (threadIdx.x & 63)], nb = wl[((i * 2 + 1) & 63) * 64 + (threadIdx.x & 63)];
            __builtin_amdgcn_sched_barrier(0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa), y, c0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 7; ++q) v[q] = fmaf(v[q], b, a);
            v[12] = __builtin_amdgcn_exp2f(-v[12]) * 0.5f; v[13] = __builtin_amdgcn_exp2f(-v[13]) * 0.5f;
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wb), y, c1, 0, 0, 0);
#pragma unroll
            for (int q = 7; q < 12; ++q) v[q] = fmaf(v[q], b, a);
            v[12] = __builtin_amdgcn_rcpf(1.0f + v[12]); v[13] = __builtin_amdgcn_rcpf(1.0f + v[13]);
            v[14] = __builtin_amdgcn_exp2f(-v[14]) * 0.5f; v[15] = __builtin_amdgcn_exp2f(-v[15]) * 0.5f;
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa), x, c0, 0, 0, 0);
            v[14] = __builtin_amdgcn_rcpf(1.0f + v[14]); v[15] = __builtin_amdgcn_rcpf(1.0f + v[15]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q + 4], b, v[q]);
            wa = na; wb = nb;
            __builtin_amdgcn_sched_barrier(0);
        } else if (KIND == 5) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 7; ++q) v[q] = fmaf(v[q], b, a);
            v[12] = __builtin_amdgcn_exp2f(-v[12]) * 0.5f; v[13] = __builtin_amdgcn_exp2f(-v[13]) * 0.5f;
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c1, 0, 0, 0);
#pragma unroll
            for (int q = 7; q < 12; ++q) v[q] = fmaf(v[q], b, a);
            v[12] = __builtin_amdgcn_rcpf(1.0f + v[12]); v[13] = __builtin_amdgcn_rcpf(1.0f + v[13]);
            v[14] = __builtin_amdgcn_exp2f(-v[14]) * 0.5f; v[15] = __builtin_amdgcn_exp2f(-v[15]) * 0.5f;
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, c0, 0, 0, 0);
            v[14] = __builtin_amdgcn_rcpf(1.0f + v[14]); v[15] = __builtin_amdgcn_rcpf(1.0f + v[15]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaf(v[q + 4], b, v[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (KIND == 0) v[q] = fmaf(v[q], b, a);
                else if (KIND == 1) v[q] = __expf(v[q]) * 0.5f;
                else v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[q])) + a;
            }
        }
    }
    float s = c0[0] + c1[1];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x + blockIdx.x * TPB] = s;
}
template <int KIND, int TPB> float run(float* d, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<KIND, TPB>), dim3(256), dim3(TPB), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KIND, TPB>), dim3(256), dim3(TPB), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 3 * 1e6f / iters;   // ns per loop iteration of every wave
}
#define ROW(KIND, NAME) printf("%-44s 1 wave/SIMD %7.1f   2 waves %7.1f   4 waves %7.1f   ns per iteration\n", NAME, \
    run<KIND, 256>(d, it), run<KIND, 512>(d, it), run<KIND, 1024>(d, it));
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    const int it = 20000;
    ROW(0, "16 independent fma")
    ROW(1, "16 x (exp + 2 mul)")
    ROW(2, "16 x swish (+fma): 2 trans + 4 valu")
    ROW(3, "2 MFMA f16 32x32x16 + 16 fma")
    ROW(4, "2 MFMA only")
    ROW(5, "edge group: 3 MFMA + 20 fma + 4 exp + 4 rcp")
    ROW(6, "  + 2 ds_read_b128 feeding the MFMAs")
    return 0;
}
