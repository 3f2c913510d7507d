// wide.hip::spd_solve on random SPD systems around the blocked elimination's limit (n = 64), both dtypes, against a host Gauss-Jordan in long double.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -o spd_check spd_check.hip && ./spd_check
#include <cstdio>
#include <random>
#include <vector>

#include "../../aux_ssm_samplers_amd/csrc/wide.hip"
namespace ax { void set_error(const char*, ...) {} void* ws_take(auxssm_ctx*, size_t) { return nullptr; } }
using namespace ax::wide;

template <typename R> __global__ void __launch_bounds__(NT) k_spd(const R* __restrict__ in, R* __restrict__ out, int n, int nct, const unsigned char* __restrict__ skipg, R* __restrict__ hl_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, ld = ldp_(nct);
    Bump L{smem};
    R* Z = L.take<R>(n * ld);
    R* piv = L.take<R>(n);
    R* rowbuf = L.take<R>(2 * (nct + 1) + NWV);
    unsigned char* skip = L.take<unsigned char>(n);
    for (int e = tid; e < n * nct; e += NT) Z[(e / nct) * ld + e % nct] = in[e];
    for (int k = tid; k < n; k += NT) skip[k] = skipg[k];
    __syncthreads();
    R hl = 0;
    const bool ok = spd_split_fits(n, nct, ld) ? spd_solve_split<R>(Z, ld, n, nct, skip, rowbuf, piv, &hl, tid) : spd_solve<R>(Z, ld, n, nct, skip, rowbuf, piv, &hl, tid, true);
    for (int e = tid; e < n * nct; e += NT) out[e] = Z[(e / nct) * ld + e % nct];
    if (tid == 0) hl_out[0] = ok ? hl : (R)-12345;
}

template <typename R> static void run(int n, int nr, int nskip) {
    const int nct = n + nr, ld = ldp_(nct);
    std::mt19937 g(n * 131 + nr);
    std::normal_distribution<double> N01(0, 1);
    std::vector<double> A((size_t)n * n), S((size_t)n * n, 0.0), RHS((size_t)n * nr);
    std::vector<unsigned char> skip(n, 0);
    for (int k = 0; k < nskip; ++k) skip[(k * 7 + 3) % n] = 1;
    for (auto& v : A) v = N01(g) / std::sqrt((double)n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = i == j ? 0.7 : 0.0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * A[j * n + k];
            S[i * n + j] = (skip[i] || skip[j]) ? 0.0 : s;
        }
    for (auto& v : RHS) v = N01(g);
    std::vector<R> in((size_t)n * nct), out((size_t)n * nct);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) in[i * nct + j] = (R)S[i * n + j];
        for (int j = 0; j < nr; ++j) in[i * nct + n + j] = (R)RHS[i * nr + j];
    }
    // host reference: unit rows for the deleted indices
    std::vector<long double> M((size_t)n * nct);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < nct; ++j) M[i * nct + j] = j < n ? (skip[i] ? (i == j ? 1.0L : 0.0L) : (long double)S[i * n + j]) : (long double)RHS[i * nr + (j - n)];
    long double hl = 0;
    for (int k = 0; k < n; ++k) {
        const long double p = M[k * nct + k];
        if (!skip[k]) hl += 0.5L * logl(p);
        for (int j = 0; j < nct; ++j) M[k * nct + j] /= p;
        for (int i = 0; i < n; ++i)
            if (i != k) {
                const long double f = M[i * nct + k];
                for (int j = 0; j < nct; ++j) M[i * nct + j] -= f * M[k * nct + j];
            }
    }
    R *din, *dout, *dhl;
    unsigned char* dsk;
    hipMalloc(&din, in.size() * sizeof(R)); hipMalloc(&dout, in.size() * sizeof(R)); hipMalloc(&dhl, sizeof(R)); hipMalloc(&dsk, n);
    hipMemcpy(din, in.data(), in.size() * sizeof(R), hipMemcpyHostToDevice);
    hipMemcpy(dsk, skip.data(), n, hipMemcpyHostToDevice);
    const size_t lds = al16((size_t)n * ld * sizeof(R)) + al16(n * sizeof(R)) + al16((2 * (nct + 1) + NWV) * sizeof(R)) + al16(n) + 64;
    hipFuncSetAttribute((const void*)k_spd<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_spd<R>), dim3(1), dim3(NT), lds, 0, din, dout, n, nct, dsk, dhl);
    R hlg;
    hipMemcpy(out.data(), dout, in.size() * sizeof(R), hipMemcpyDeviceToHost);
    hipMemcpy(&hlg, dhl, sizeof(R), hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < n; ++i)
        for (int j = n; j < nct; ++j) err = std::max(err, (double)fabsl((long double)out[i * nct + j] - M[i * nct + j]));
    printf("%s n=%3d nr=%3d skip=%d lds=%zu: max|dX| %.3e  hl %.6f (ref %.6f)\n", sizeof(R) == 4 ? "f32" : "f64", n, nr, nskip, lds, err, (double)hlg, (double)hl);
    hipFree(din); hipFree(dout); hipFree(dhl); hipFree(dsk);
}

int main() {
    for (int n : {40, 64, 65, 68, 96, 128})
        for (int nr : {9, 72}) {
            if (n + nr <= 256) {
                run<float>(n, nr, 0);
                run<float>(n, nr, 5);
                if ((size_t)n * ldp_(n + nr) * 8 < 150000) {
                    run<double>(n, nr, 0);
                    run<double>(n, nr, 5);
                }
            }
        }
    return 0;
}
