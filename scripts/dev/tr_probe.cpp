// Dev probe: what does ds_read_b64_tr_b16 return?  LDS holds halfword i at byte 2 i; every lane reads from its own address.
//   hipcc --offload-arch=gfx950 -O2 -o tr_probe scripts/dev/tr_probe.cpp && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int* addr_of_lane, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(&lds[0]) + (unsigned)addr_of_lane[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v[0] & 0xFFFF;
    out[threadIdx.x * 4 + 1] = v[0] >> 16;
    out[threadIdx.x * 4 + 2] = v[1] & 0xFFFF;
    out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main() {
    int h_addr[64];
    unsigned short h_out[256];
    int* d_addr;
    unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr));
    hipMalloc(&d_out, sizeof(h_out));
    const char* names[3] = {"lane l reads bytes 8 l (contiguous)", "lane l reads row (l & 15) of a [16][64-halfword] image, column block l >> 4: byte 128 (l & 15) + 8 (l >> 4)",
                            "lane l reads row (l >> 2 & 3)+4(l>>4) .. : byte 64 (l & 15) + 8 (l >> 4)"};
    for (int t = 0; t < 3; ++t) {
        for (int l = 0; l < 64; ++l) h_addr[l] = t == 0 ? 8 * l : t == 1 ? 128 * (l & 15) + 8 * (l >> 4) : 64 * (l & 15) + 8 * (l >> 4);
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("---- %s\n", names[t]);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d (addr halfword %4d): %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
        }
    }
    return 0;
}
