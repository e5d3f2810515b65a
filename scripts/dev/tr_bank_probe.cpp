// Dev probe: LDS cycles per ds_read_b64_tr_b16 (and ds_read_b64) for candidate address patterns, all 8 waves of a 512-thread workgroup
// reading at once (the LDS array is the bottleneck then): cycles per wave-instruction = elapsed / (reads per wave x 8 waves).
//   hipcc --offload-arch=gfx950 -O2 -o scripts/dev/tr_bank_probe scripts/dev/tr_bank_probe.cpp && ./scripts/dev/tr_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int TR>
__global__ void __launch_bounds__(512) k(const int* addr_of_lane, unsigned* out, long long* cycles, int reps) {
    __shared__ unsigned lds[160 * 1024 / 4 - 64];
    for (int i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += 512) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned a = (unsigned)(size_t)(&lds[0]) + (unsigned)addr_of_lane[lane] + wave * 16384u;   // (each wave its own 16 KiB: same banks)
    u32x2 acc = {0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
        if (TR) {
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:4608\n\tds_read_b64_tr_b16 %2, %8 offset:9216\n\t"
                         "ds_read_b64_tr_b16 %3, %8 offset:13824\n\tds_read_b64_tr_b16 %4, %8 offset:2304\n\tds_read_b64_tr_b16 %5, %8 offset:6912\n\t"
                         "ds_read_b64_tr_b16 %6, %8 offset:11520\n\tds_read_b64_tr_b16 %7, %8 offset:1152\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:4608\n\tds_read_b64 %2, %8 offset:9216\n\t"
                         "ds_read_b64 %3, %8 offset:13824\n\tds_read_b64 %4, %8 offset:2304\n\tds_read_b64 %5, %8 offset:6912\n\t"
                         "ds_read_b64 %6, %8 offset:11520\n\tds_read_b64 %7, %8 offset:1152\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
        }
        acc += v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[threadIdx.x] = acc[0] ^ acc[1];
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
int main() {
    int* d_addr; unsigned* d_out; long long* d_cyc;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 512 * 4); hipMalloc(&d_cyc, 8);
    struct P { std::string name; int (*f)(int g, int a, int b); };
    // lane l = 16 g + s, s = 4 a + b: the lane reads 8 bytes at f(g, a, b)
    std::vector<P> ps = {
        {"rows 4g+a, row stride 288 B (shipped k_linear_bwd_r)", [](int g, int a, int b) { return (4 * g + a) * 288 + 8 * b; }},
        {"rows 4g+a, row stride 272 B", [](int g, int a, int b) { return (4 * g + a) * 272 + 8 * b; }},
        {"rows 4g+a, row stride 320 B", [](int g, int a, int b) { return (4 * g + a) * 320 + 8 * b; }},
        {"rows 4g+a, row stride 264 B", [](int g, int a, int b) { return (4 * g + a) * 264 + 8 * b; }},
        {"rows 4g+a, row stride 256 B", [](int g, int a, int b) { return (4 * g + a) * 256 + 8 * b; }},
        {"rows 8g+a, row stride 288 B", [](int g, int a, int b) { return (8 * g + a) * 288 + 8 * b; }},
        {"[4][16] subtile contiguous (128 B), groups 128 B apart", [](int g, int a, int b) { return g * 128 + (4 * a + b) * 8; }},
        {"subtile contiguous, groups 256 B apart", [](int g, int a, int b) { return g * 256 + (4 * a + b) * 8; }},
        {"subtile contiguous, groups 512 B apart (guide layout 1)", [](int g, int a, int b) { return g * 512 + (4 * a + b) * 8; }},
        {"subtile contiguous, groups 640 B apart", [](int g, int a, int b) { return g * 640 + (4 * a + b) * 8; }},
        {"subtile contiguous, groups 1152 B apart", [](int g, int a, int b) { return g * 1152 + (4 * a + b) * 8; }},
        {"row stride 32 B within a group (a*32), groups 144 B apart", [](int g, int a, int b) { return g * 144 + a * 32 + b * 8; }},
        {"row stride 64 B within a group, groups 32 B apart", [](int g, int a, int b) { return g * 32 + a * 64 + b * 8; }},
        {"row stride 288, b chunks, groups +1152 (rows 4g) swizzle b^a", [](int g, int a, int b) { return (4 * g + a) * 288 + 8 * (b ^ a); }},
        {"row stride 272, swizzle b^a", [](int g, int a, int b) { return (4 * g + a) * 272 + 8 * (b ^ a); }},
        {"row stride 144 B (72 halfwords)", [](int g, int a, int b) { return (4 * g + a) * 144 + 8 * b; }},
        {"row stride 160 B", [](int g, int a, int b) { return (4 * g + a) * 160 + 8 * b; }},
        {"row stride 96 B", [](int g, int a, int b) { return (4 * g + a) * 96 + 8 * b; }},
    };
    const int reps = 2000;
    for (int tr = 1; tr >= 0; --tr)
        for (auto& p : ps) {
            int h[64];
            for (int l = 0; l < 64; ++l) h[l] = p.f(l >> 4, (l & 15) >> 2, l & 3);
            hipMemcpy(d_addr, h, sizeof(h), hipMemcpyHostToDevice);
            long long c = 0;
            for (int w = 0; w < 2; ++w) {
                if (tr) hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, d_addr, d_out, d_cyc, reps);
                else hipLaunchKernelGGL(k<0>, dim3(1), dim3(512), 0, 0, d_addr, d_out, d_cyc, reps);
                hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("%s  %-62s %6.2f cycles per wave-instruction\n", tr ? "tr_b16 " : "b64    ", p.name.c_str(), (double)c / (reps * 8.0 * 8.0));
        }
    return 0;
}
