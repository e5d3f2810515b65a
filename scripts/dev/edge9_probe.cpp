// Hardware facts the generation-9 edge kernel (k_mpnn_edge32t) relies on, checked on the box before the kernel is trusted:
//  (1) v_mfma_f32_32x32x16_f16 operand / result layouts: A lane l = row l&31, k = 8(l>>5)+0..7; B lane l = column l&31, same k;
//      D lane l = column l&31, rows (r&3) + 8(r>>2) + 4(l>>5), r = 0..15
//  (2) fp16 SUBNORMAL inputs of that MFMA are not flushed (the single-accumulator split keeps the low piece unscaled)
//  (3) v_fma_mixlo_f16 / v_fma_mixhi_f16: fp16(a*b + c) with ONE rounding, subnormal results kept
//  (4) v_permlane32_swap: upper half of the first register <-> lower half of the second
// hipcc --offload-arch=gfx950 -O3 -o edge9_probe scripts/dev/edge9_probe.cpp && ./edge9_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k_layout(const float* A /*[32][16]*/, const float* B /*[16][32]*/, float* D /*[32][32]*/, float scaleA) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f16x8 a, b;
    for (int c = 0; c < 8; ++c) {
        a[c] = (_Float16)(A[i * 16 + 8 * h + c] * scaleA);
        b[c] = (_Float16)B[(8 * h + c) * 32 + i];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

__global__ void k_mix(const float* x, const float* y, const float* z, uint32_t* out_lo_hi, int n) {
    const int t = threadIdx.x;
    if (t >= n) return;
    uint32_t w = 0;
    // lo half = fp16(x*y + z), hi half = fp16(x*y - z)
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(w) : "v"(x[t]), "v"(y[t]), "v"(z[t]));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3" : "+v"(w) : "v"(x[t]), "v"(y[t]), "v"(z[t]));
    out_lo_hi[t] = w;
}

__global__ void k_swap(uint32_t* out) {
    uint32_t a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
#else
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
#endif
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
}

static float h2f(uint16_t h) {
    _Float16 v;
    memcpy(&v, &h, 2);
    return (float)v;
}

int main() {
    // ---- (1) layouts, asymmetric A and B
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), ref(32 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5) * 0.25f;
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 2) % 13 - 6) * 0.125f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1.0f);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(D[i] - ref[i]));
    printf("(1) 32x32x16 f16 layout: max |D - A.B| = %.3g  %s\n", e, e < 1e-6 ? "OK" : "MISMATCH");
    // ---- (2) subnormal A inputs: scale A by 2^-18 so that every entry (multiples of 0.25 * 2^-18 = 2^-20) is an fp16 subnormal
    const float sc = ldexpf(1.f, -18);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, sc);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    e = 0; double m = 0; for (int i = 0; i < 1024; ++i) { e = fmax(e, fabs(D[i] - ref[i] * sc)); m = fmax(m, fabs(ref[i] * sc)); }
    printf("(2) fp16-subnormal A operand: max |D - exact| = %.3g (exact max %.3g)  %s\n", e, m, e < 1e-3 * m ? "subnormals KEPT" : "subnormals FLUSHED");
    // ---- (3) fma_mix
    const int n = 8;
    float x[n] = {1.0009765625f, 3.0f, 1e-3f, 0.7f, -2.5f, 1.5f, 6e-5f, 1.0f}, y[n] = {1.0009765625f, 0.333333f, 1e-3f, 0.9f, 0.3f, 2e-5f, 0.5f, 5.9e-8f},
          z[n] = {-1.0f, 0.f, 0.f, -0.63f, 0.75f, 0.f, 0.f, 0.f};
    float *dx, *dy, *dz; uint32_t* dw, w[n];
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dz, n * 4); hipMalloc(&dw, 128 * 4);
    hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice); hipMemcpy(dz, z, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mix, dim3(1), dim3(64), 0, 0, dx, dy, dz, dw, n);
    hipMemcpy(w, dw, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const double lo = (double)x[i] * y[i] + z[i], hi = (double)x[i] * y[i] - z[i];
        printf("(3) fma_mix %d: lo %.9g (exact %.9g)  hi %.9g (exact %.9g)\n", i, h2f(w[i] & 0xFFFF), lo, h2f(w[i] >> 16), hi);
    }
    // ---- (4) permlane32_swap
    uint32_t s[128];
    hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, dw);
    hipMemcpy(s, dw, 128 * 4, hipMemcpyDeviceToHost);
    printf("(4) permlane32_swap(a = 1000 + lane, b = 2000 + lane): a[0] %u a[31] %u a[32] %u a[63] %u | b[0] %u b[31] %u b[32] %u b[63] %u\n",
           s[0], s[31], s[32], s[63], s[64], s[95], s[96], s[127]);
    printf("    expected (upper half of a <-> lower half of b): a = 1000..1031, 2000..2031 ; b = 1032..1063, 2032..2063\n");
    return 0;
}
