// phase profile of wide.hip's fold_step (d = 64, fp32): cycles per phase, one workgroup
#define AUXSSM_FOLD_PROF 1
#include "../../aux_ssm_samplers_amd/csrc/wide.hip"
namespace ax { void set_error(const char*, ...) {} void* ws_take(auxssm_ctx*, size_t) { return nullptr; } }
using namespace ax::wide;
template <typename R, bool FULL> __global__ void __launch_bounds__(NT) kb(const R* src, R* out, long long* cyc, int d, int iters, int fetch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, ldd = ldp_(d);
    Bump L{smem};
    Fold<R> g;
    carve_fold<R>(L, g, d, FULL);
    for (int k = 0; k < 16; ++k) g.ph[k] = 0;
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            if (FULL) g.A[r * ldd + q] = r == q, g.J[r * ldd + q] = 0;
            g.C[r * ldd + q] = r == q ? (R)0.5 : (R)0;
        }
    for (int k = tid; k < d; k += NT) g.b[k] = 0, g.eta[k] = 0;
    g.z = 0;
    __syncthreads();
    const long long ni = info_size(d);
    const StepSrc<R> nx{src, src + d * d, src + 2 * d * d + ni, src + 2 * d * d};
    {
        StepRegs<R> sr;
        step_fetch<R>(sr, nx, d, tid);
        if (FULL) step_drop<R>(g, sr, g.F, g.Z + d, g.ldz, d, tid);
        else step_drop<R>(g, sr, g.F, g.A, ldd, d, tid);
    }
    __syncthreads();
    const long long t0 = clock64();
    g.t0 = t0;
    for (int it = 0; it < iters; ++it) {
        if (FULL) fold_step<R, true>(g, fetch ? &nx : nullptr, d, tid);
        else fold_step_down<R>(g, fetch ? &nx : nullptr, d, tid);
    }
    const long long tot = clock64() - t0;
    if (tid == 0) {
        cyc[0] = tot;
        for (int k = 0; k < 16; ++k) cyc[1 + k] = g.ph[k];
    }
    out[tid] = g.C[(tid % d) * ldd + (tid % d)] + g.b[tid % d];
}
int main() {
    const int d = 64, iters = 50;
    const long long ni = info_size(d);
    std::vector<float> hs(2 * d * d + ni + d, 0.f);
    for (int r = 0; r < d; ++r)
        for (int q = 0; q < d; ++q) {
            hs[r * d + q] = (r == q ? 0.9f : 0.f) + 0.01f * (float)(((r * 31 + q * 17) % 13) - 6) / 6.f;  // F
            hs[d * d + r * d + q] = r == q ? 0.3f : 0.f;                                                   // Q
            hs[2 * d * d + r * d + q] = r == q ? 4.f : 0.f;                                               // Lam
        }
    for (int k = 0; k < d; ++k) hs[2 * d * d + d * d + k] = 0.1f * k;  // g0
    hs[2 * d * d + d * d + d] = 1.f, hs[2 * d * d + d * d + d + 1] = 0.5f, hs[2 * d * d + d * d + d + 2] = (float)d;
    float *src, *out; long long* cyc;
    (void)hipMalloc(&src, hs.size() * 4); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 256);
    (void)hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const char* names[] = {"-", "FC, mb, Pp", "FA, W, lm, g", "lu_solve (2d+1)", "fetch, v, PM, pg, tv, eta, MFA, A', C'", "drop, z, b, J, sym C", "sym J"};
    for (int fetch = 1; fetch >= 0; --fetch)
    for (int full = 1; full >= 0; --full) {
        const size_t lds = lds_fold(4, d, full);
        if (full) { (void)hipFuncSetAttribute((const void*)kb<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL((kb<float, true>), dim3(1), dim3(NT), lds, 0, src, out, cyc, d, iters, fetch); }
        else { (void)hipFuncSetAttribute((const void*)kb<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL((kb<float, false>), dim3(1), dim3(NT), lds, 0, src, out, cyc, d, iters, fetch); }
        (void)hipDeviceSynchronize();
        long long c[17]; (void)hipMemcpy(c, cyc, 17 * 8, hipMemcpyDeviceToHost);
        printf("fold_step<%s> fetch=%d d=%d: %.0f cycles/step (lds %zu)\n", full ? "FULL" : "down", fetch, d, (double)c[0] / iters, lds);
        for (int k = 0; k < 7; ++k) printf("   %-28s %8.0f\n", names[k], (double)c[1 + k] / iters);
        printf("   %-28s %8.0f\n", "(loop / fetch issue)", (double)c[16] / iters);
    }
    return 0;
}
