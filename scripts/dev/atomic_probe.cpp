// Dev probe (round 5): what does a by-source reduction of the edge backward's dp cost when it is done with
// global_atomic_add_f32 instead of 4.4 GB written + re-read?  Mesh-stencil graph like the metric graph (4 x 256^2 nodes, 33 edges each).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe scripts/dev/atomic_probe.cpp && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wave per 32-edge tile (edges of one target row), lane = feature pair; the values are made up from registers (no read)
template <int MODE>   // 0: plain stores to dp[e][128]   1: atomics to out[src[e]][128]   2: atomics, return-less builtin with agent scope   3: nothing (skeleton)
__global__ void __launch_bounds__(256) k_probe(const int* __restrict__ src, long E, float* __restrict__ dp, float* __restrict__ out, float seed) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    const long ntiles = (E + 31) / 32;
    for (long t = wave; t < ntiles; t += nw) {
        const long e0 = t * 32;
        const int n = (int)min(32L, E - e0);
        int my = lane < n ? src[e0 + lane] : 0;
        float v0 = seed + (float)lane, v1 = seed - (float)lane;
#pragma unroll 4
        for (int k = 0; k < n; ++k) {
            const int j = __builtin_amdgcn_readlane(my, k);
            if (MODE == 0) {
                float* p = dp + (e0 + k) * 128;
                p[lane] = v0, p[lane + 64] = v1;
            } else if (MODE == 1) {
                float* p = out + (long)j * 128;
                unsafeAtomicAdd(p + lane, v0), unsafeAtomicAdd(p + lane + 64, v1);
            } else if (MODE == 2) {
                float* p = out + (long)j * 128;
                __hip_atomic_fetch_add(p + lane, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(p + lane + 64, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            v0 += 1.0f, v1 -= 1.0f;
        }
        if (MODE == 3 && v0 == 12345.0f) out[0] = v1;
    }
}

// the by-source segment sum as the library does it today (rows of dp gathered through the by-source permutation): read 4.4 GB
__global__ void __launch_bounds__(256) k_segsum(const float* __restrict__ dp, const int* __restrict__ perm, const int* __restrict__ rowptr, long N, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (long r = wave; r < N; r += nw) {
        float a0 = 0, a1 = 0;
        for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float* p = dp + (long)perm[e] * 128;
            a0 += p[lane], a1 += p[lane + 64];
        }
        out[r * 128 + lane] = a0, out[r * 128 + lane + 64] = a1;
    }
}

int main() {
    const int W = 256, B = 4;
    std::vector<std::pair<int, int>> offs;
    for (int dy = -4; dy <= 4; ++dy) for (int dx = -4; dx <= 4; ++dx) if (dx * dx + dy * dy <= 10) offs.push_back({dx, dy});
    printf("stencil of %zu offsets\n", offs.size());
    std::vector<int> src, rowptr{0};
    for (int b = 0; b < B; ++b) for (int y = 0; y < W; ++y) for (int x = 0; x < W; ++x) {
        for (auto [dx, dy] : offs) { int xx = x + dx, yy = y + dy; if (xx >= 0 && xx < W && yy >= 0 && yy < W) src.push_back((b * W + yy) * W + xx); }
        rowptr.push_back((int)src.size());
    }
    const long E = src.size(), N = (long)B * W * W;
    // by-source permutation (the graph is symmetric: CSR by source has the same rowptr)
    std::vector<int> perm(E), fill(rowptr.begin(), rowptr.end() - 1);
    for (long i = 0; i < N; ++i) for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) perm[fill[src[e]]++] = e;
    printf("N = %ld, E = %ld (%.2f per row), dp = %.2f GB\n", N, E, (double)E / N, E * 512e-9);
    int *d_src, *d_perm, *d_rp; float *d_dp, *d_out;
    CK(hipMalloc(&d_src, E * 4)); CK(hipMalloc(&d_perm, E * 4)); CK(hipMalloc(&d_rp, (N + 1) * 4));
    CK(hipMalloc(&d_dp, E * 512)); CK(hipMalloc(&d_out, N * 512));
    CK(hipMemcpy(d_src, src.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_perm, perm.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rp, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0, N * 512)); CK(hipMemset(d_dp, 0, E * 512));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char* name, auto fn) {
        for (int i = 0; i < 2; ++i) fn();
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) fn();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-58s %8.3f ms\n", name, ms / 10);
    };
    for (int grid : {1024, 4096, 16384}) {
        printf("grid %d x 256 threads\n", grid);
        time("  skeleton (index loads + readlane loop)", [&] { hipLaunchKernelGGL(k_probe<3>, dim3(grid), dim3(256), 0, 0, d_src, E, d_dp, d_out, 1.0f); });
        time("  plain stores dp[e][128] (E x 512 B)", [&] { hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(256), 0, 0, d_src, E, d_dp, d_out, 1.0f); });
        time("  unsafeAtomicAdd to out[src[e]][128]", [&] { hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(256), 0, 0, d_src, E, d_dp, d_out, 1.0f); });
        time("  __hip_atomic_fetch_add relaxed agent scope", [&] { hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(256), 0, 0, d_src, E, d_dp, d_out, 1.0f); });
    }
    time("k_segsum: gather E x 512 B by source (16384 x 256)", [&] { hipLaunchKernelGGL(k_segsum, dim3(16384), dim3(256), 0, 0, d_dp, d_perm, d_rp, N, d_out); });
    // correctness of the atomic path: out must equal the gathered sum of what the store path wrote
    CK(hipMemset(d_out, 0, N * 512));
    hipLaunchKernelGGL(k_probe<0>, dim3(4096), dim3(256), 0, 0, d_src, E, d_dp, d_out, 1.0f);
    hipLaunchKernelGGL(k_probe<1>, dim3(4096), dim3(256), 0, 0, d_src, E, d_dp, d_out, 1.0f);
    std::vector<float> got(N * 128), ref(N * 128);
    CK(hipMemcpy(got.data(), d_out, N * 512, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_segsum, dim3(16384), dim3(256), 0, 0, d_dp, d_perm, d_rp, N, d_out);
    CK(hipMemcpy(ref.data(), d_out, N * 512, hipMemcpyDeviceToHost));
    double worst = 0; long bad = 0;
    for (long i = 0; i < N * 128; ++i) { double d = std::fabs((double)got[i] - ref[i]) / std::max(1.0, std::fabs((double)ref[i])); worst = std::max(worst, d); bad += d > 1e-5; }
    printf("atomic sum vs gathered sum: worst relative difference %.3g, %ld elements above 1e-5\n", worst, bad);
    return 0;
}
