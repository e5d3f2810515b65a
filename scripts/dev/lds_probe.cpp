// Can a workgroup own all 160 KiB of a gfx950 CU's LDS?  (static allocation of exactly 163840 bytes, and a few sizes below)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ void __launch_bounds__(256) k(float* out) {
    __shared__ __attribute__((aligned(16))) float buf[BYTES / 4];
    for (int i = threadIdx.x; i < BYTES / 4; i += 256) buf[i] = (float)i;
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < BYTES / 4; i += 256) s += buf[(i * 7 + 3) % (BYTES / 4)];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int BYTES> void run(float* d) {
    hipLaunchKernelGGL(k<BYTES>, dim3(512), dim3(256), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    hipError_t e2 = hipGetLastError();
    printf("%7d bytes: %s / %s\n", BYTES, hipGetErrorString(e), hipGetErrorString(e2));
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<131072>(d); run<160 * 1024 - 2048>(d); run<160 * 1024 - 256>(d); run<160 * 1024>(d);
    return 0;
}
