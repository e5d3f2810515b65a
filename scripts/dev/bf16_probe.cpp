// Probes for the bf16-split plan: (1) operand layout of v_mfma_f32_32x32x16_bf16, (2) overlap with VALU,
// (3) accuracy of the 6-term (and 3-term) bf16 split of an fp32 product against fp64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short top16(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }

// D[i][j] = sum_k A[i][k] * B[k][j], A [32][16], B [16][32] given as fp32 (values exactly bf16)
__global__ void k_layout(const float* A, const float* B, float* D, int variant) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    s16x8 a, b;
    for (int c = 0; c < 8; ++c) {
        int k = variant == 0 ? h * 8 + c : (c >> 2) * 8 + h * 4 + (c & 3);   // candidate k maps
        a[c] = (short)top16(A[i * 16 + k]);
        b[c] = (short)top16(B[k * 32 + i]);
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

template <int MODE>
__global__ void __launch_bounds__(512) k_overlap(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    if (MODE == 3 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(3);
    if (MODE == 4 && __builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(3);
    if (MODE == 5) {   // one wave does both: 8 independent FMAs between consecutive MFMAs
        if (wave < 4) {
            s16x8 x, y;
            for (int c = 0; c < 8; ++c) { x[c] = (short)(threadIdx.x + c); y[c] = (short)(c * 3); }
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c2, 0, 0, 0);
                x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c3, 0, 0, 0);
                x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
            }
            out[threadIdx.x + blockIdx.x * 512] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
        return;
    }
    if (wave < 4) {
        if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) {
            s16x8 x, y;
            for (int c = 0; c < 8; ++c) { x[c] = (short)(threadIdx.x + c); y[c] = (short)(c * 3); }
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c3, 0, 0, 0);
            }
            out[threadIdx.x + blockIdx.x * 512] = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 4) {
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters * 8; ++i) {
            x0 = fmaf(x0, b, a); x1 = fmaf(x1, b, a); x2 = fmaf(x2, b, a); x3 = fmaf(x3, b, a);
            x4 = fmaf(x4, b, a); x5 = fmaf(x5, b, a); x6 = fmaf(x6, b, a); x7 = fmaf(x7, b, a);
        }
        out[threadIdx.x + blockIdx.x * 512] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k_overlap<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_overlap<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 5;
}

// split accuracy: one wave computes D = W[32 x K] * X^T[K x 32] with the bf16 split, K = 128
__global__ void k_split(const float* W, const float* X, float* D, int K, int nterms, int variant) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        s16x8 w[3], x[3];
        for (int c = 0; c < 8; ++c) {
            int k = k0 + (variant == 0 ? h * 8 + c : (c >> 2) * 8 + h * 4 + (c & 3));
            float wv = W[i * K + k], xv = X[i * K + k];
            float whi = __uint_as_float(__float_as_uint(wv) & 0xFFFF0000u), wr = wv - whi;
            float wmi = __uint_as_float(__float_as_uint(wr) & 0xFFFF0000u), wlo = wr - wmi;
            float xhi = __uint_as_float(__float_as_uint(xv) & 0xFFFF0000u), xr = xv - xhi;
            float xmi = __uint_as_float(__float_as_uint(xr) & 0xFFFF0000u), xlo = xr - xmi;
            w[0][c] = (short)top16(whi); w[1][c] = (short)top16(wmi); w[2][c] = (short)top16(wlo);
            x[0][c] = (short)top16(xhi); x[1][c] = (short)top16(xmi); x[2][c] = (short)top16(xlo);
        }
        if (nterms >= 6) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) k_dep(float* out, int iters) {
    s16x8 x, y;
    for (int c = 0; c < 8; ++c) { x[c] = (short)(threadIdx.x + c); y[c] = (short)(c * 3); }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 12; ++q) c[q % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c[q % NACC], 0, 0, 0);
    }
    out[threadIdx.x + blockIdx.x * 256] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
template <int NACC> float run_dep(float* d, int iters) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k_dep<NACC>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_dep<NACC>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / 5;
}
int main() {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    // (1) layout
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
    for (auto& v : A) v = (float)((int)(U(rng) * 8));          // small ints: exactly bf16
    for (auto& v : B) v = (float)((int)(U(rng) * 8));
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, variant);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; err = fmax(err, fabs(s - D[i * 32 + j])); }
        printf("layout variant %d (any consistent k map must be exact): max err %.3g\n", variant, err);
    }
    // (2) overlap
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 40000;
    float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
    float t3 = run<3>(d, iters), t4 = run<4>(d, iters), t5 = run<5>(d, iters);
    printf("both, VALU waves prio 3: %.3f ms; both, MFMA waves prio 3: %.3f ms; same wave 4 FMA per MFMA gap (%d FMAs vs %d in VALU-only): %.3f ms\n", t3, t4, iters * 16, iters * 64, t5);
    printf("bf16 MFMA only: %.3f ms (%.0f TF)  VALU only: %.3f ms  both: %.3f ms (sum %.3f, max %.3f)\n", t0,
           256.0 * 4 * iters * 4 * 32768.0 / t0 / 1e9, t1, t2, t0 + t1, fmax(t0, t1));
    printf("dependent bf16 MFMA chains, 12 MFMA/iter: 1 acc %.3f ms, 2 acc %.3f ms, 3 acc %.3f ms, 4 acc %.3f ms\n",
           run_dep<1>(d, 20000), run_dep<2>(d, 20000), run_dep<3>(d, 20000), run_dep<4>(d, 20000));
    // (3) split accuracy
    const int K = 128;
    std::vector<float> W(32 * K), X(32 * K), R(32 * 32);
    for (auto& v : W) v = U(rng) * 0.2f;
    for (auto& v : X) v = U(rng) * 3.f;
    float *dW, *dX; hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    for (int nterms : {6, 3}) {
        hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dW, dX, dD, K, nterms, 1);
        hipMemcpy(R.data(), dD, R.size() * 4, hipMemcpyDeviceToHost);
        double num = 0, den = 0, num32 = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; float s32 = 0;
            for (int k = 0; k < K; ++k) { s += (double)W[i * K + k] * X[j * K + k]; s32 = fmaf(W[i * K + k], X[j * K + k], s32); }
            num += (R[i * 32 + j] - s) * (R[i * 32 + j] - s); den += s * s; num32 += (s32 - s) * (s32 - s);
        }
        printf("split %d terms: rel L2 vs fp64 %.3e   (plain fp32 fmaf chain: %.3e)\n", nterms, sqrt(num / den), sqrt(num32 / den));
    }
    return 0;
}
