// Operand / result layout of v_mfma_f32_16x16x32_f16 as assumed by mpnn_edge16.hip:
//   A (16 x 32): lane l holds row i = l & 15, k = 8 * (l >> 4) + 0..7        B (32 x 16): lane l holds column j = l & 15, same k
//   D (16 x 16): lane l holds column j = l & 15, rows i = 4 * (l >> 4) + 0..3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    f16x8 a, b;
    for (int c = 0; c < 8; ++c) { a[c] = (_Float16)A[i * 32 + 8 * q + c]; b[c] = (_Float16)B[(8 * q + c) * 16 + i]; }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = d[r];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6) / 4.f; hB[i] = (float)((i * 5) % 11 - 5) / 8.f; }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)hA[i * 32 + kk] * hB[kk * 16 + j];
        worst = fmax(worst, fabs(s - hD[i * 16 + j]));
    }
    printf("16x16x32 f16 layout check: max abs error %.3g (%s)\n", worst, worst < 1e-3 ? "layout as assumed" : "LAYOUT DIFFERS");
    return 0;
}
