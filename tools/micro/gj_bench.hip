// micro-benchmark of wide.hip's cooperative solves on one workgroup (d = 64): cycles per call
#include "../../aux_ssm_samplers_amd/csrc/wide.hip"
namespace ax { void set_error(const char*, ...) {} void* ws_take(auxssm_ctx*, size_t) { return nullptr; } }
using namespace ax::wide;
template <typename R> __global__ void __launch_bounds__(NT) kb(R* out, long long* cyc, int d, int iters, int mode, int nsolve) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    Bump L{smem};
    const int nct = 3 * d + 1, ldz = ldp_(nct), ncs = nsolve > 0 ? nsolve : nct;
    R* Z = L.take<R>(d * ldz);
    R* rowbuf = L.take<R>(2 * (nct + 1) + NWV);
    R* pinv = L.take<R>(d);
    int* iperm = L.take<int>(d);
    unsigned int* key = L.take<unsigned int>(2);
    R* invd = L.take<R>(d);
    R* dg = L.take<R>(d);
    int* flag = L.take<int>(1);
    long long tot = 0;
    for (int it = 0; it < iters; ++it) {
        for (int r = tid / 64; r < d; r += NWV)
            for (int c = tid & 63; c < nct; c += 64) Z[r * ldz + c] = (r == c ? (R)(d + 1) : (R)0) + (R)(((r * 131 + c * 71 + it) % 17) - 8) * (R)0.05;
        if (mode == 1 || mode == 4)  // SPD for the Cholesky: make the leading block symmetric
            for (int r = tid / 64; r < d; r += NWV)
                for (int c = tid & 63; c < r; c += 64) Z[r * ldz + c] = Z[c * ldz + r];
        __syncthreads();
        const long long t0 = clock64();
        if (mode == 0) lu_solve<R>(Z, ldz, d, ncs, rowbuf, pinv, iperm, key, tid);
        if (mode == 1) (void)chol<R>(Z, ldz, d, nullptr, invd, dg, flag, tid);
        if (mode == 2) trsm_l<R>(Z, ldz, d, pinv, Z + d, ldz, d + 2, tid);
        if (mode == 3) gemm<false, false>(d, d, d, Z, ldz, Z + d, ldz, Z + 2 * d, ldz, (R)1, (R)0, tid);
        if (mode == 5) {  // eight independent products back to back, one barrier: the steady-state cost of a product
#pragma unroll 1
            for (int q = 0; q < 8; ++q) gemm<false, false>(d, d, d, Z, ldz, Z + d, ldz, Z + 2 * d, ldz, (R)1, (R)0, tid, (const R*)nullptr, 0, false);
            __syncthreads();
        }
        if (mode == 6) {  // matrix cores alone: 8 x 16 MFMAs per wave on two accumulators, operands already in registers
            const float a0 = (float)Z[tid], b0 = (float)Z[tid + 7];
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll 1
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, a0, c1, 0, 0, 0);
                }
            }
            Z[tid] = (R)(c0[0] + c1[1] + c0[2] + c1[3]);
            __syncthreads();
        }
        if (mode == 7) {  // fragment loads alone: 8 x 32 LDS reads per lane in the K = 64 pattern
            const int lane = tid & 63, wv = tid >> 6, lo = lane & 15, hi = lane >> 4, i0 = (wv >> 2) << 4, j0 = (wv & 3) << 4;
            float acc = 0;
#pragma unroll 1
            for (int q = 0; q < 8; ++q) {
                const float* qa = (const float*)Z + (i0 + lo) * ldz + 16 * hi + q;
                const float* qb = (const float*)Z + d + 16 * hi * ldz + j0 + lo + q;
                float fa[16], fb[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) fa[u] = qa[u], fb[u] = qb[u * ldz];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += fa[u] * fb[u];
            }
            Z[tid] = (R)acc;
            __syncthreads();
        }
        if (mode == 4) (void)spd_solve<R>(Z, ldz, d, nsolve > 0 ? nsolve : 2 * d + 2, nullptr, rowbuf, pinv, (R*)nullptr, tid, true);
        tot += clock64() - t0;
    }
    if (tid == 0) cyc[0] = tot;
    out[tid] = Z[(tid % d) * ldz + d + (tid % d)];
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const int d = 64, iters = 50;
    const size_t lds = 120 * 1024;
    hipFuncSetAttribute((const void*)kb<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char* names[] = {"gj_solve [W|A|C|v]", "chol", "trsm_l (d+2 cols)", "gemm 64^3", "spd_solve [S|H|r|r]", "8 x gemm 64^3, one barrier", "8 x 16 MFMAs per wave, no loads", "8 x 32 fragment loads per lane, no MFMA"};
    for (int mode = 0; mode < 8; ++mode) {
        hipLaunchKernelGGL(kb<float>, dim3(1), dim3(NT), lds, 0, out, cyc, d, iters, mode, 0);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-22s d=%d NT=%d: %.0f cycles/call (%.1f per step)\n", names[mode], d, NT, (double)c / iters, (double)c / iters / d);
    }
    const int ncols[] = {65, 129, 130, 193};
    for (int mode : {0, 4})
        for (int nc : ncols) {
            hipLaunchKernelGGL(kb<float>, dim3(1), dim3(NT), lds, 0, out, cyc, d, iters, mode, nc);
            hipDeviceSynchronize();
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-22s nct=%d (z_free) d=%d: %.0f cycles/call\n", mode == 0 ? "lu_solve" : "spd_solve", nc, d, (double)c / iters);
        }
    return 0;
}
