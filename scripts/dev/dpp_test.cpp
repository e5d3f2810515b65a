#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xF, true); }
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int v = 100 + lane;
    out[lane] = dpp_i<0x111, 0xF>(v);
    out[64 + lane] = dpp_i<0x118, 0xF>(v);
    out[128 + lane] = dpp_i<0x142, 0xA>(v);
    out[192 + lane] = dpp_i<0x142, 0xF>(v);
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"shr1", "shr8", "bcast15 mask A", "bcast15 mask F"};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i]); printf("\n"); }
    return 0;
}
