// Does a LARGE straight-line loop body change what an MFMA hides?  Same streams as overlap_probe (a), but the loop body is the
// pattern repeated U times (code size U x 4 x (1 + N) instructions): a tiny body replays from the instruction buffer, a 20-40 KiB
// body (what the unrolled edge kernels are) has to be fetched every iteration.
// hipcc --offload-arch=gfx950 -O3 -w -o fetch_probe fetch_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));

template <int N, int U, int TPB>
__global__ void __launch_bounds__(TPB) k(float* out, int iters) {
    f16x8 x, y;
    for (int c = 0; c < 8; ++c) { x[c] = (_Float16)(threadIdx.x * 0.01f + c); y[c] = (_Float16)(c * 0.3f); }
    f32x16 c32[4] = {{0}, {0}, {0}, {0}};
    float a = threadIdx.x * 1e-6f, b = 0.999f, v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i * 0.1f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c32[m]) : "v"(x), "v"(y));
#pragma unroll
                for (int i = 0; i < N; ++i) { FMA(i) }
            }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += c32[m][0];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x + blockIdx.x * TPB] = s;
}
template <int N, int U, int TPB> float run(float* d, int total_mfma_per_wave) {
    const int iters = total_mfma_per_wave / (4 * U);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<N, U, TPB>), dim3(256), dim3(TPB), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<N, U, TPB>), dim3(256), dim3(TPB), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 3 * 1e6f / (iters * 4 * U);   // ns per MFMA (+ fillers) of one wave
}
template <int N> void row(float* d) {
    const int T = 1 << 18;
    printf("32x32 MFMA + %2d fma:  body x1 (%5d B): %6.2f | x16: %6.2f | x128 (%6d B): %6.2f | x512 (%7d B): %6.2f ns   [2 waves/SIMD x128: %6.2f]\n", N,
           4 * (8 + 8 * N), run<N, 1, 256>(d, T), run<N, 16, 256>(d, T), 128 * 4 * (8 + 8 * N), run<N, 128, 256>(d, T),
           512 * 4 * (8 + 8 * N), run<N, 512, 256>(d, T), run<N, 128, 512>(d, T) / 2);
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    row<0>(d); row<3>(d); row<5>(d); row<8>(d); row<12>(d);
    return 0;
}
