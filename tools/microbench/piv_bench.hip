// Microbenchmark of the 64x64 pivot block (W = L^-1 of an SPD tile): wg_pivot_block (VALU, NWPT waves) against
// wg_pivot_block_mfma (matrix cores, 3 waves), one workgroup of BENCH_NT threads per CU.  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 -fno-slp-vectorize \
//         -DNWPT=16 -DNT=1024 -o piv_bench tools/microbench/piv_bench.hip && ./piv_bench
// Prints us per block for both and the largest difference of W / W^T against a double-precision Cholesky inverse.
#include "../../lqp_py_amd/csrc/lqp_boxqp.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace lqp;
#ifndef NWPT
#define NWPT 16
#endif
#ifndef BENCH_NT
#define BENCH_NT 1024
#endif
template <bool MFMA>
__global__ __launch_bounds__(BENCH_NT) void k_piv(const float* src, float* out, float* outT, int reps) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    float* W = (float*)smem; float* WT = W + 64 * SPD_LS; float* pcol = WT + 64 * SPD_LS; int* flag = (int*)(pcol + PIV_LDS);
    if (threadIdx.x == 0) flag[0] = 0;
    __syncthreads();
    for (int r = 0; r < reps; ++r) {
        if constexpr (MFMA) wg_pivot_block_mfma<false>(src + blockIdx.x * 4096, W, WT, pcol, flag, 0);
        else wg_pivot_block<NWPT>(src + blockIdx.x * 4096, W, WT, pcol, flag, 0);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 64 * 64; i += BENCH_NT) {
        out[blockIdx.x * 4096 + i] = W[(i >> 6) * SPD_LS + (i & 63)];
        outT[blockIdx.x * 4096 + i] = WT[(i >> 6) * SPD_LS + (i & 63)];
    }
    if (threadIdx.x == 0 && flag[0] != 0) out[blockIdx.x * 4096] = -12345.f;
}
int main() {
    const int B = 256;
#ifdef LQP_PIV_STAMPS
    {
        unsigned long long* st; (void)hipMalloc(&st, 128 * 8); (void)hipMemset(st, 0, 128 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_piv_stamps), &st, sizeof(st));
    }
#endif
    std::vector<float> h(B * 4096);
    for (int b = 0; b < B; ++b) for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) {
        float v = 0.3f * std::cos(0.37f * (i + 1) * (j + 1) + b) / (1.f + 0.1f * std::abs(i - j));
        if (i == j) v = 3.f + 0.05f * i;
        h[b * 4096 + i * 64 + j] = v; h[b * 4096 + j * 64 + i] = v;
    }
    // reference: W = L^-1 in double for problem 0 and B-1
    auto ref = [&](int b, std::vector<double>& Wd) {
        std::vector<double> L(4096, 0.0);
        for (int j = 0; j < 64; ++j) {
            double d = h[b * 4096 + j * 64 + j];
            for (int k = 0; k < j; ++k) d -= L[j * 64 + k] * L[j * 64 + k];
            d = std::sqrt(d); L[j * 64 + j] = d;
            for (int i = j + 1; i < 64; ++i) {
                double v = h[b * 4096 + i * 64 + j];
                for (int k = 0; k < j; ++k) v -= L[i * 64 + k] * L[j * 64 + k];
                L[i * 64 + j] = v / d;
            }
        }
        Wd.assign(4096, 0.0);
        for (int c = 0; c < 64; ++c) {          // solve L w = e_c
            for (int i = c; i < 64; ++i) {
                double v = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) v -= L[i * 64 + k] * Wd[k * 64 + c];
                Wd[i * 64 + c] = v / L[i * 64 + i];
            }
        }
    };
    float *d, *o, *oT; hipMalloc(&d, h.size() * 4); hipMalloc(&o, h.size() * 4); hipMalloc(&oT, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int lds = (2 * 64 * SPD_LS + PIV_LDS + 8) * 4;
    hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
    for (int mf = 0; mf < 2; ++mf) {
        for (int reps : {1, 1, 16, 16}) {
            hipEventRecord(a);
            if (mf) hipLaunchKernelGGL(k_piv<true>, dim3(B), dim3(BENCH_NT), lds, 0, d, o, oT, reps);
            else hipLaunchKernelGGL(k_piv<false>, dim3(B), dim3(BENCH_NT), lds, 0, d, o, oT, reps);
            hipEventRecord(b2); hipEventSynchronize(b2);
            float ms; hipEventElapsedTime(&ms, a, b2);
            printf("%s reps %2d: %.2f us per pivot block (%.1f us total)\n", mf ? "mfma" : "valu", reps, ms * 1e3 / reps, ms * 1e3);
        }
        std::vector<float> r(h.size()), rT(h.size());
        hipMemcpy(r.data(), o, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(rT.data(), oT, h.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, etr = 0, upper = 0;
        for (int b : {0, 1, B - 1}) {
            std::vector<double> Wd; ref(b, Wd);
            for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) {
                emax = std::max(emax, std::abs((double)r[b * 4096 + i * 64 + j] - Wd[i * 64 + j]));
                etr = std::max(etr, std::abs((double)rT[b * 4096 + j * 64 + i] - (double)r[b * 4096 + i * 64 + j]));
                if (j > i) upper = std::max(upper, (double)std::abs(r[b * 4096 + i * 64 + j]));
            }
        }
        printf("%s: max |W - L^-1 (double)| = %.3g, max |W^T - W'| = %.3g, max above the diagonal = %.3g, W[0][0]=%g W[63][0]=%g W[63][63]=%g\n",
               mf ? "mfma" : "valu", emax, etr, upper, r[0], r[63 * 64], r[63 * 64 + 63]);
    }
#ifdef LQP_PIV_STAMPS
    {
        unsigned long long* st; (void)hipMemcpyFromSymbol(&st, HIP_SYMBOL(g_piv_stamps), sizeof(st));
        unsigned long long h2[128]; (void)hipMemcpy(h2, st, sizeof(h2), hipMemcpyDeviceToHost);
        const unsigned long long t0 = h2[0];
        printf("stamps (cycles since wave 0 entered its role; last call of workgroup 0)\n wave0:");
        for (int i = 0; i <= 9; ++i) printf(" %lld", (long long)(h2[i] - t0));
        printf(" | end %lld", (long long)(h2[20] - t0));
        printf("\n wave3:"); for (int i = 0; i <= 9; ++i) printf(" %lld", (long long)(h2[96 + i] - t0));
        printf(" | end %lld", (long long)(h2[96 + 20] - t0));
        printf("\n wave1: enter %lld stores-begin %lld stores-end %lld; wave2: enter %lld stores-begin %lld\n", (long long)(h2[32] - t0),
               (long long)(h2[33] - t0), (long long)(h2[34] - t0), (long long)(h2[64] - t0), (long long)(h2[65] - t0));
    }
#endif
    return 0;
}
