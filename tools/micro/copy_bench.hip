// HBM bandwidth probe for the roofline denominator ("verify numbers on the box"): a streaming copy and a streaming read of
// 4 GiB with 16-byte accesses.  hipcc --offload-arch=gfx950 -O3 tools/micro/copy_bench.hip -o tools/micro/copy_bench && ./copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_read(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) out[0] = acc;  // keeps the loads alive
}
int main() {
    const size_t bytes = 4ull << 30, n = bytes / sizeof(float4);
    float4 *a, *b;
    float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        for (int which = 0; which < 2; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double moved = which == 0 ? 2.0 * bytes : 1.0 * bytes;
            printf("%s grid=%d: %.3f ms, %.2f TB/s\n", which == 0 ? "copy (read + write)" : "read only", grid, best, moved / best / 1e9);
        }
    }
    return 0;
}
