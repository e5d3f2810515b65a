// How many VALU instructions of which kind fit in the shadow of back-to-back v_mfma_f32_32x32x16_bf16 of the SAME
// wave (one wave per SIMD)?  Reports ns per MFMA for N fillers per MFMA gap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int N, int KIND>   // KIND 0: independent fma, 1: one dependent fma chain, 2: v_exp, 3: dpp mov + fma (scan step), 4: and+sub (split)
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    s16x8 x, y;
    for (int c = 0; c < 8; ++c) { x[c] = (short)(threadIdx.x + c); y[c] = (short)(c * 3); }
    f32x16 c0 = {0}, c1 = {0};
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    for (int i = 0; i < iters; ++i) {
#define FILL()                                                                                              \
    _Pragma("unroll") for (int q = 0; q < N; ++q) {                                                         \
        if (KIND == 0) v[q % 16] = fmaf(v[q % 16], b, a);                                                     \
        else if (KIND == 1) v[0] = fmaf(v[0], b, a);                                                          \
        else if (KIND == 2) v[q % 16] = __expf(v[q % 16]) * 0.5f;                                             \
        else if (KIND == 3) v[q % 16] += b * __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[q % 16]), 0x111, 0xF, 0xF, true)); \
        else { float t = __uint_as_float(__float_as_uint(v[q % 16]) & 0xFFFF0000u); v[q % 16] = v[q % 16] - t + a; } \
    }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0); FILL();
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0); FILL();
    }
    float s = c0[0] + c1[1];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x + blockIdx.x * 256] = s;
}
template <int N, int KIND> float run(float* d, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<N, KIND>), dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 3 * 1e6f / (iters * 2.0f);   // ns per MFMA
}
#define ROW(KIND, NAME) printf("%-28s N=0 %.1f  2 %.1f  4 %.1f  6 %.1f  8 %.1f  12 %.1f  16 %.1f  ns/MFMA\n", NAME, \
    run<0, KIND>(d, it), run<2, KIND>(d, it), run<4, KIND>(d, it), run<6, KIND>(d, it), run<8, KIND>(d, it), run<12, KIND>(d, it), run<16, KIND>(d, it));
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const int it = 40000;
    ROW(0, "independent fma")
    ROW(1, "dependent fma chain")
    ROW(2, "exp + mul")
    ROW(3, "dpp mov + fma")
    ROW(4, "and + sub + add")
    return 0;
}
