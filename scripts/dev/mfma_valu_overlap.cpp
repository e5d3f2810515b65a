// Does v_mfma_f32_32x32x2_f32 overlap with VALU work of a co-resident wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: waves 0-3 MFMA, 4-7 idle; 1: 0-3 idle, 4-7 VALU; 2: both; 3: 4-7 trans (exp)
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    if (wave < 4) {
        if (MODE == 0 || MODE == 2 || MODE == 4) {
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
            }
            out[threadIdx.x + blockIdx.x * 512] = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (MODE == 1 || MODE == 2) {
            float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
            for (int i = 0; i < iters * 16; ++i) {     // 8 independent fma chains
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
            }
            out[threadIdx.x + blockIdx.x * 512] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
        if (MODE == 3 || MODE == 4) {
            float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
            for (int i = 0; i < iters * 8; ++i) {
                x0 = __expf(x0) * 0.1f; x1 = __expf(x1) * 0.1f; x2 = __expf(x2) * 0.1f; x3 = __expf(x3) * 0.1f;
            }
            out[threadIdx.x + blockIdx.x * 512] = x0 + x1 + x2 + x3;
        }
    }
}
// same wave: NV independent FMAs (+ optional exp) between consecutive fp32 MFMAs
template <int NV, int NEXP>
__global__ void __launch_bounds__(256) k_same(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a + i;
    for (int i = 0; i < iters; ++i) {
#define GAP() { _Pragma("unroll") for (int q = 0; q < NV; ++q) x[q] = fmaf(x[q], b, a); _Pragma("unroll") for (int q = 0; q < NEXP; ++q) x[8 + q] = __expf(x[8 + q]) * 0.1f; }
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); GAP();
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0); GAP();
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0); GAP();
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0); GAP();
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[threadIdx.x + blockIdx.x * 256] = s;
}
template <int NV, int NEXP> float run_same(float* d, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k_same<NV, NEXP>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_same<NV, NEXP>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 5;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 5;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    double mfma_flops = 256.0 * 4 * iters * 4 * 4096.0;
    printf("MFMA only      : %.3f ms (%.1f TF)\n", t0, mfma_flops / t0 / 1e9);
    printf("VALU fma only  : %.3f ms (%.1f TF)\n", t1, 256.0 * 4 * 64 * iters * 16 * 8 * 2 / t1 / 1e9);
    printf("MFMA + VALU fma: %.3f ms  (sum would be %.3f, max %.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
    printf("exp only       : %.3f ms\n", t3);
    printf("MFMA + exp     : %.3f ms  (sum would be %.3f, max %.3f)\n", t4, t0 + t3, t0 > t3 ? t0 : t3);
    printf("same wave, fp32 MFMA + N FMA per gap: N=0 %.3f  N=4 %.3f  N=8 %.3f  N=8+2exp %.3f  N=8+4exp %.3f ms\n",
           run_same<0, 0>(d, iters), run_same<4, 0>(d, iters), run_same<8, 0>(d, iters), run_same<8, 2>(d, iters), run_same<8, 4>(d, iters));
    return 0;
}
