// micro-benchmark: cost of one "__syncthreads + dependent LDS read" step for different workgroup sizes (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(float* out, long long* cyc, int iters) {
    __shared__ float sh[2048];
    const int tid = threadIdx.x;
    sh[tid] = tid;
    __syncthreads();
    float acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // barrier only
            __syncthreads();
        } else if (MODE == 1) {  // write, barrier, dependent read
            sh[tid] = acc + i;
            __syncthreads();
            acc += sh[(tid + 1) % blockDim.x];
        } else if (MODE == 2) {  // dependent LDS read chain, no barrier
            acc += sh[((int)acc + tid + i) & 1023];
        } else if (MODE == 3) {  // raw s_barrier without the fence
            __builtin_amdgcn_s_barrier();
        }
    }
    long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + tid] = acc;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8 * 1024);
    const int iters = 2000;
    for (int nt : {64, 256, 512, 1024}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int grid) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(nt), 0, 0, out, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(nt), 0, 0, out, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(nt), 0, 0, out, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(nt), 0, 0, out, cyc, iters);
            };
            launch(1);
            hipDeviceSynchronize();
            hipEventRecord(e0); launch(1); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("nt=%4d mode=%d: %.1f clock64/iter, %.1f ns/iter (wall, 1 block)\n", nt, mode, (double)c / iters, ms * 1e6 / iters);
        }
    }
    return 0;
}
