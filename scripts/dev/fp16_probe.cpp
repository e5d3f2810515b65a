// fp16x3 split accuracy probe: x = h + l/2048 (h = RTZ fp16, l = RNE fp16 of the scaled remainder),
// product = hh + 2^-11 (h l' + l' h); compared with fp64 and the fp32 fmaf chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
__global__ void k_split(const float* W, const float* X, float* D, int K) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 accM = {0}, accC = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 wh, wl, xh, xl;
        for (int c = 0; c < 8; c += 2) {
            int ka = k0 + (c >> 2) * 8 + h * 4 + (c & 3), kb = ka + 1;
            float w0 = W[i * K + ka], w1 = W[i * K + kb], x0 = X[i * K + ka], x1 = X[i * K + kb];
            fp16x2 hw = __builtin_amdgcn_cvt_pkrtz(w0, w1), hx = __builtin_amdgcn_cvt_pkrtz(x0, x1);
            wh[c] = (_Float16)hw[0]; wh[c + 1] = (_Float16)hw[1]; xh[c] = (_Float16)hx[0]; xh[c + 1] = (_Float16)hx[1];
            wl[c] = (_Float16)((w0 - (float)hw[0]) * 2048.f); wl[c + 1] = (_Float16)((w1 - (float)hw[1]) * 2048.f);
            xl[c] = (_Float16)((x0 - (float)hx[0]) * 2048.f); xl[c + 1] = (_Float16)((x1 - (float)hx[1]) * 2048.f);
        }
        accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accM, 0, 0, 0);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accC, 0, 0, 0);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accC, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = accM[r] + accC[r] * (1.f / 2048.f);
}
int main() {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    for (float wscale : {0.2f, 0.01f, 1e-4f}) for (float xscale : {3.f, 100.f, 1e-3f}) {
        const int K = 128;
        std::vector<float> W(32 * K), X(32 * K), R(32 * 32);
        for (auto& v : W) v = U(rng) * wscale;
        for (auto& v : X) v = U(rng) * xscale;
        float *dW, *dX, *dD; hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dD, R.size() * 4);
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dW, dX, dD, K);
        hipMemcpy(R.data(), dD, R.size() * 4, hipMemcpyDeviceToHost);
        double num = 0, den = 0, num32 = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; float s32 = 0;
            for (int k = 0; k < K; ++k) { s += (double)W[i * K + k] * X[j * K + k]; s32 = fmaf(W[i * K + k], X[j * K + k], s32); }
            num += (R[i * 32 + j] - s) * (R[i * 32 + j] - s); den += s * s; num32 += (s32 - s) * (s32 - s);
        }
        printf("w~%g x~%g: fp16x3 rel L2 vs fp64 %.3e   (fp32 fmaf chain %.3e)\n", wscale, xscale, sqrt(num / den), sqrt(num32 / den));
    }
    return 0;
}
