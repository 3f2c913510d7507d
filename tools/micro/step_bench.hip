// What does one elimination step cost?  Stripped variants of wide.hip::spd_solve_t on one 1024-lane workgroup (n = 64, nct = 130).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NT = 1024, NWV = 16;
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
template <int MODE> __global__ void __launch_bounds__(NT) k(float* out, long long* cyc, int n, int nct, int iters) {
    __shared__ float rowbuf[2 * 260];
    __shared__ float piv[128];
    const int tid = threadIdx.x, ti = tid >> 6, tj = tid & 63;
    float z[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) z[a][b] = 1.0f + 0.001f * (tid + a * 7 + b * 3);
    for (int i = tid; i < 520; i += NT) rowbuf[i] = 0.001f * i + 1.0f;
    __syncthreads();
    const int rb = nct + 1;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
        for (int k = 0; k < n; ++k) {
            const int ka = k / NWV, src = k & 63;
            float* rbuf = rowbuf + (k & 1) * rb;
            if (MODE >= 3 && ti == k - ka * NWV) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (a == ka) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            if (tj + 64 * b < nct) rbuf[tj + 64 * b] = z[a][b];
                            if (b == 0 && tj == src) {
                                rbuf[nct] = 1.0f / z[a][b];
                                piv[k] = z[a][b];
                            }
                        }
                    }
            }
            __syncthreads();
            if (MODE >= 1) {
                const float inv = rbuf[nct];
                float zk[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) zk[b] = (tj + 64 * b < nct) ? rbuf[tj + 64 * b] : 0.f;
                if (MODE >= 2) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const float f = (ti + NWV * a != k) ? bcast(z[a][0], src) * inv : 0.f;
#pragma unroll
                        for (int b = 0; b < 4; ++b) z[a][b] -= f * zk[b];
                    }
                } else {
                    z[0][0] += inv + zk[0] + zk[1] + zk[2] + zk[3];
                }
            }
        }
    long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    float acc = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) acc += z[a][b];
    out[tid] = acc + piv[tid & 63];
}
template <int NB, bool EARLY> __global__ void __launch_bounds__(NT) k2(float* out, long long* cyc, int n, int nct, int iters) {
    __shared__ float rowbuf[2 * 260];
    __shared__ float piv[128];
    const int tid = threadIdx.x, ti = tid >> 6, tj = tid & 63;
    float z[4][NB];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < NB; ++b) z[a][b] = 1.0f + 0.001f * (tid + a * 7 + b * 3);
    for (int i = tid; i < 520; i += NT) rowbuf[i] = 0.001f * i + 1.0f;
    __syncthreads();
    const int rb = nct + 1;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (EARLY && ti == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (tj + 64 * b < nct) rowbuf[tj + 64 * b] = z[0][b];
            if (tj == 0) rowbuf[nct] = 1.0f / z[0][0], piv[0] = z[0][0];
        }
        for (int k = 0; k < n; ++k) {
            const int ka = k / NWV, src = k & 63;
            float* rbuf = rowbuf + (k & 1) * rb;
            if (!EARLY && ti == k - ka * NWV) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (a == ka) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            if (tj + 64 * b < nct) rbuf[tj + 64 * b] = z[a][b];
                            if (b == 0 && tj == src) {
                                rbuf[nct] = 1.0f / z[a][b];
                                piv[k] = z[a][b];
                            }
                        }
                    }
            }
            __syncthreads();
            const float inv = rbuf[nct];
            float zk[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) zk[b] = (tj + 64 * b < nct) ? rbuf[tj + 64 * b] : 0.f;
            const int an = (k + 1) / NWV;
            const bool own_next = EARLY && k + 1 < n && ti == k + 1 - an * NWV;
            float* nbuf = rowbuf + ((k + 1) & 1) * rb;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float f = (ti + NWV * a != k) ? bcast(z[a][0], src) * inv : 0.f;
#pragma unroll
                for (int b = 0; b < NB; ++b) z[a][b] -= f * zk[b];
                if (own_next && a == an) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (tj + 64 * b < nct) nbuf[tj + 64 * b] = z[a][b];
                        if (b == 0 && tj == ((k + 1) & 63)) {
                            nbuf[nct] = 1.0f / z[a][b];
                            piv[k + 1] = z[a][b];
                        }
                    }
                }
            }
        }
    }
    long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    float acc = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < NB; ++b) acc += z[a][b];
    out[tid] = acc + piv[tid & 63];
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1 << 16); (void)hipMalloc(&cyc, 64);
    const int n = 64, nct = 130, iters = 20;
    const char* names[] = {"barrier only", "+ 5 LDS reads", "+ readlane + 16 FMA", "+ owner publish"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            (void)hipDeviceSynchronize();
        }
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-24s %.0f cycles per step\n", names[mode], (double)c / iters / n);
    }
    const char* n2[] = {"NB=4 late publish", "NB=3 late publish", "NB=4 early publish", "NB=3 early publish"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL((k2<4, false>), dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            if (mode == 1) hipLaunchKernelGGL((k2<3, false>), dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            if (mode == 2) hipLaunchKernelGGL((k2<4, true>), dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            if (mode == 3) hipLaunchKernelGGL((k2<3, true>), dim3(1), dim3(NT), 0, 0, out, cyc, n, nct, iters);
            (void)hipDeviceSynchronize();
        }
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-24s %.0f cycles per step\n", n2[mode], (double)c / iters / n);
    }
    return 0;
}
