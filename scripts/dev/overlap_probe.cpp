// Ground truth for the edge-kernel redesign: does VALU / transcendental work overlap with MFMA on ONE SIMD of gfx950
//   (a) inside one wave: N independent fillers pinned (asm volatile) between consecutive MFMAs, N = 0..10
//   (b) across waves: waves 0-3 of a 512-thread workgroup run an MFMA-only loop, waves 4-7 (same SIMDs: a workgroup's
//       waves go to the SIMDs cyclically) a VALU-only / transcendental-only loop; time vs each half alone
// hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));

template <int N, int KIND> __device__ __forceinline__ void fill(float (&v)[16], float a, float b) {
    // KIND 0: plain fma fillers, 1: v_exp fillers
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (KIND == 0) { FMA(i) } else { EXP(i) }
    }
}

// (a) one instruction stream: MFMA, N fillers, MFMA, N fillers ... (4 independent accumulators)
template <int SHAPE, int N, int KIND, int TPB>
__global__ void __launch_bounds__(TPB) k_same(float* out, int iters) {
    f16x8 x, y;
    for (int c = 0; c < 8; ++c) { x[c] = (_Float16)(threadIdx.x * 0.01f + c); y[c] = (_Float16)(c * 0.3f); }
    f32x16 c32[4] = {{0}, {0}, {0}, {0}};
    f32x4 c16[4] = {{0}, {0}, {0}, {0}};
    float a = threadIdx.x * 1e-6f, b = 0.999f, v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i * 0.1f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c32[m]) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c16[m]) : "v"(x), "v"(y));
            fill<N, KIND>(v, a, b);
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += c32[m][0] + c16[m][1];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x + blockIdx.x * TPB] = s;
}

// (b) wave specialisation: ROLE bit 0 = the MFMA half runs, bit 1 = the filler half runs
template <int SHAPE, int KIND, int ROLE>
__global__ void __launch_bounds__(512) k_cross(float* out, int iters) {
    // role by the SIMD the wave really runs on: the first wave to arrive on a SIMD takes the MFMA role, the second the filler role
    __shared__ int cnt[4];
    __shared__ int simd_of[8];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (hwid >> 4) & 3;
    int rank = 0;
    if ((threadIdx.x & 63) == 0) { rank = atomicAdd(&cnt[simd], 1); simd_of[threadIdx.x >> 6] = simd * 16 + rank; }
    rank = __shfl(rank, 0);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && iters < 0) for (int w = 0; w < 8; ++w) printf("wave %d simd %d rank %d\n", w, simd_of[w] >> 4, simd_of[w] & 15);
    if (iters < 0) return;
    f16x8 x, y;
    for (int c = 0; c < 8; ++c) { x[c] = (_Float16)(threadIdx.x * 0.01f + c); y[c] = (_Float16)(c * 0.3f); }
    f32x16 c32[4] = {{0}, {0}, {0}, {0}};
    f32x4 c16[4] = {{0}, {0}, {0}, {0}};
    float a = threadIdx.x * 1e-6f, b = 0.999f, v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i * 0.1f;
    if (rank == 0) {
        if (ROLE & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c32[m]) : "v"(x), "v"(y));
                    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c16[m]) : "v"(x), "v"(y));
                }
            }
    } else {
        if (ROLE & 2)
            for (int it = 0; it < iters; ++it) fill<16, KIND>(v, a, b);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += c32[m][0] + c16[m][1];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x + blockIdx.x * 512] = s;
}

template <typename F> float timeit(F launch) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    launch();
    hipEventRecord(s);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 3;
}

template <int SHAPE, int N, int KIND> void row_same(float* d, int it) {
    const float t1 = timeit([&] { hipLaunchKernelGGL((k_same<SHAPE, N, KIND, 256>), dim3(256), dim3(256), 0, 0, d, it); });
    const float t2 = timeit([&] { hipLaunchKernelGGL((k_same<SHAPE, N, KIND, 512>), dim3(256), dim3(512), 0, 0, d, it); });
    // ns per MFMA of one wave (4 MFMAs per iteration)
    printf("  %dx%d MFMA + %2d %s fillers each:  1 wave/SIMD %6.2f ns per MFMA+fillers   2 waves/SIMD %6.2f (per wave: x2 work)\n", SHAPE, SHAPE, N,
           KIND ? "exp" : "fma", t1 * 1e6f / it / 4, t2 * 1e6f / it / 4);
}
template <int SHAPE, int KIND> void row_cross(float* d, int it) {
    const float tm = timeit([&] { hipLaunchKernelGGL((k_cross<SHAPE, KIND, 1>), dim3(256), dim3(512), 0, 0, d, it); });
    const float tv = timeit([&] { hipLaunchKernelGGL((k_cross<SHAPE, KIND, 2>), dim3(256), dim3(512), 0, 0, d, it); });
    const float tb = timeit([&] { hipLaunchKernelGGL((k_cross<SHAPE, KIND, 3>), dim3(256), dim3(512), 0, 0, d, it); });
    printf("  cross-wave %dx%d: MFMA-only half %7.3f ms, %s-only half %7.3f ms, both %7.3f ms  (max = %.3f, sum = %.3f)\n", SHAPE, SHAPE, tm,
           KIND ? "exp" : "fma", tv, tb, tm > tv ? tm : tv, tm + tv);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    const int it = 20000;
    printf("(a) fillers pinned between the MFMAs of ONE wave\n");
    row_same<32, 0, 0>(d, it); row_same<32, 2, 0>(d, it); row_same<32, 4, 0>(d, it); row_same<32, 5, 0>(d, it); row_same<32, 6, 0>(d, it);
    row_same<32, 8, 0>(d, it); row_same<32, 10, 0>(d, it);
    row_same<32, 2, 1>(d, it); row_same<32, 4, 1>(d, it); row_same<32, 6, 1>(d, it);
    row_same<16, 0, 0>(d, it); row_same<16, 1, 0>(d, it); row_same<16, 2, 0>(d, it); row_same<16, 3, 0>(d, it); row_same<16, 4, 0>(d, it);
    row_same<16, 6, 0>(d, it); row_same<16, 2, 1>(d, it); row_same<16, 4, 1>(d, it);
    printf("(b) one MFMA-only wave beside one filler-only wave on EVERY SIMD (roles by HW_ID.SIMD_ID)\n");
    hipLaunchKernelGGL((k_cross<32, 0, 3>), dim3(1), dim3(512), 0, 0, d, -1);
    hipDeviceSynchronize();
    row_cross<32, 0>(d, it); row_cross<32, 1>(d, it); row_cross<16, 0>(d, it); row_cross<16, 1>(d, it);
    return 0;
}
