// phase profile of the blocked Gauss-Jordan solve AS IT WAS before its trailing update moved onto the matrix cores (a copy of that gj_solve_blk with s_memtime at
// the phase boundaries): the measurement that showed the VALU rank-16 update at 35 k of 64 k cycles and the single-wave panel phase at 28 k
#include "../../aux_ssm_samplers_amd/csrc/wide.hip"
namespace ax { void set_error(const char*, ...) {} void* ws_take(auxssm_ctx*, size_t) { return nullptr; } }
using namespace ax::wide;
template <typename R>
__device__ __forceinline__ void gj_prof(R* Z, int ld, int n, int nct, R* pinv, int* iperm, int tid, long long* ph) {
    constexpr int NRR = 4, NB = 16, PS = NB + 1;
    const int ti = tid >> 6, tj = tid & 63;
    R z[NRR][4];
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        const int r = ti + NWV * a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = tj + 64 * b;
            z[a][b] = (r < n && c < nct) ? Z[r * ld + c] : (R)0;
        }
    }
    __syncthreads();  // Z is in registers: its LDS image is scratch until the write-back
    R* panel = Z;                 // [n][PS]
    R* Dm = panel + n * PS;       // [n][NB]
    R* Zp = Dm + n * NB;          // [NB][nct]
    int* pos = (int*)(Zp + NB * nct);  // [n] position of row r among the block's pivots, -1 if none
    bool used_lane = false;       // (wave 0) row tj already served as a pivot row
    long long tA = clock64();
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        const int kb = k0 >> 6, c0 = k0 & 63;
        if (tj >= c0 && tj < c0 + nb) {
#pragma unroll
            for (int a = 0; a < NRR; ++a) {
                const int r = ti + NWV * a;
                const R v = kb == 0 ? z[a][0] : (kb == 1 ? z[a][1] : (kb == 2 ? z[a][2] : z[a][3]));
                if (r < n) panel[r * PS + (tj - c0)] = v;
            }
        }
        __syncthreads();
        { long long t = clock64(); ph[0] += t - tA; tA = t; }
        if (ti == 0) {
            const int r = tj;
            const bool valid = r < n;
            R pz[NB], g[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                pz[j] = (valid && j < nb) ? panel[r * PS + j] : (R)0;
                g[j] = 0;
            }
            int mypos = -1;
            R myinv = 0;
            auto step = [&](int j) {
                const unsigned int ky = (valid && !used_lane) ? piv_key(pz[j], r) : 0u;
                const unsigned int best = wave_umax_dpp(ky);
                const int pr = 127 - (int)(best & 0x7fu);
                const R inv = rcp_nr(bcast(pz[j], pr));
                const bool me = r == pr;
                const R f = me ? (R)0 : pz[j] * inv;
#pragma unroll
                for (int jj = j + 1; jj < NB; ++jj) pz[jj] -= f * bcast(pz[jj], pr);
#pragma unroll
                for (int i = 0; i < j; ++i) g[i] -= f * bcast(g[i], pr);
                g[j] = -f;
                used_lane = used_lane || me;
                mypos = me ? j : mypos;
                myinv = me ? inv : myinv;
            };
            if (nb == NB) {  // full block: no per-step branches
#pragma unroll
                for (int j = 0; j < NB; ++j) step(j);
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) step(j);
            }
            if (valid && mypos >= 0) {  // this lane's row was a pivot of the block: its reciprocal pivot and its position
                pinv[r] = myinv;
                iperm[r] = k0 + mypos;
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < NB; ++j) Dm[r * NB + j] = g[j];
                pos[r] = mypos;
            }
        }
        { long long t = clock64(); ph[1] += t - tA; tA = t; }
        __syncthreads();
        { long long t = clock64(); ph[4] += t - tA; tA = t; }
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
            const int r = ti + NWV * a;
            const int pp = r < n ? pos[r] : -1;
            if (pp >= 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (tj + 64 * b < nct) Zp[pp * nct + tj + 64 * b] = z[a][b];
            }
        }
        __syncthreads();
        { long long t = clock64(); ph[2] += t - tA; tA = t; }
        for (int j = 0; j < nb; ++j) {
            R zk[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) zk[b] = (tj + 64 * b < nct) ? Zp[j * nct + tj + 64 * b] : (R)0;
#pragma unroll
            for (int a = 0; a < NRR; ++a) {
                const int r = ti + NWV * a;
                const R dd = r < n ? Dm[r * NB + j] : (R)0;
#pragma unroll
                for (int b = 0; b < 4; ++b) z[a][b] += dd * zk[b];
            }
        }
        { long long t = clock64(); ph[3] += t - tA; tA = t; }
        // the next block's panel writes touch `panel` only; D / Zp / pos are rewritten after its first barrier, which no wave passes
        // before every wave has finished the update above
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        const int r = ti + NWV * a;
        if (r < n) {
            const R inv = pinv[r];
            const int kr = iperm[r];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = tj + 64 * b;
                if (c >= n && c < nct) Z[kr * ld + c] = z[a][b] * inv;
            }
        }
    }
    __syncthreads();
}

template <typename R> __global__ void __launch_bounds__(NT) kb(R* out, long long* cyc, int d, int nct, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    Bump L{smem};
    const int ldz = ldp_(nct);
    R* Z = L.take<R>(d * ldz);
    R* pinv = L.take<R>(d);
    int* iperm = L.take<int>(d);
    long long ph[5] = {0, 0, 0, 0, 0}, tot = 0;
    for (int it = 0; it < iters; ++it) {
        for (int r = tid / 64; r < d; r += NWV)
            for (int c = tid & 63; c < nct; c += 64) Z[r * ldz + c] = (r == c ? (R)(d + 1) : (R)0) + (R)(((r * 131 + c * 71 + it) % 17) - 8) * (R)0.05;
        __syncthreads();
        const long long t0 = clock64();
        gj_prof<R>(Z, ldz, d, nct, pinv, iperm, tid, ph);
        tot += clock64() - t0;
    }
    if (tid == 0 || tid == 64) {
        long long* o = cyc + (tid ? 8 : 0);
        o[0] = tot;
        for (int i = 0; i < 5; ++i) o[1 + i] = ph[i];
    }
    out[tid] = Z[(tid % d) * ldz + d + (tid % d)];
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 256);
    const int d = 64, iters = 50;
    const size_t lds = 120 * 1024;
    (void)hipFuncSetAttribute((const void*)kb<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int nct : {65, 129, 193}) {
        hipLaunchKernelGGL(kb<float>, dim3(1), dim3(NT), lds, 0, out, cyc, d, nct, iters);
        (void)hipDeviceSynchronize();
        long long c[16]; (void)hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
        for (int w = 0; w < 2; ++w)
            printf("nct=%d wave %d: total %.0f | A panel-drop+barrier %.0f | B panel %.0f | wait-after-B %.0f | publish+barrier %.0f | C update %.0f\n", nct, w, (double)c[8*w] / iters,
                   (double)c[8*w+1] / iters, (double)c[8*w+2] / iters, (double)c[8*w+5] / iters, (double)c[8*w+3] / iters, (double)c[8*w+4] / iters);
    }
    return 0;
}
