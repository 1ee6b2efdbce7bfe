// debug aid for wg_pivot_block_mfma: one workgroup, one block; dumps the scales and the coefficient queue next to a
// float emulation of the same elimination (first rows only).  Build with -DLQP_PIV_DEBUG_STOP (the function then returns
// before its final stores, so the queue survives in the W^T area).
#include "../../lqp_py_amd/csrc/lqp_boxqp.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace lqp;
__global__ __launch_bounds__(256) void k_dbg(const float* src, float* out) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    float* W = (float*)smem; float* WT = W + 64 * SPD_LS; float* pcol = WT + 64 * SPD_LS; int* flag = (int*)(pcol + PIV_LDS);
    if (threadIdx.x == 0) flag[0] = 0;
    __syncthreads();
    wg_pivot_block_mfma<false>(src, W, WT, pcol, flag, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) out[i] = WT[i];          // queue [16][4][64]
    for (int i = threadIdx.x; i < 64; i += 256) out[4096 + i] = pcol[i];   // scales
    if (threadIdx.x == 0) out[4096 + 64] = (float)flag[0];
    for (int i = threadIdx.x; i < 2048; i += 256) out[4096 + 128 + i] = W[i];    // pivot-row queue [8][4][64]
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) {
        float v = 0.3f * std::cos(0.37f * (i + 1) * (j + 1)) / (1.f + 0.1f * std::abs(i - j));
        if (i == j) v = 3.f + 0.05f * i;
        h[i * 64 + j] = v; h[j * 64 + i] = v;
    }
    // float emulation: plain right-looking elimination
    std::vector<float> X(h), coefs(4096, 0.f), sc(64), Xp0;
    for (int c = 0; c < 64; ++c) {
        const float d = X[c * 64 + c], s = 1.f / std::sqrt(d);
        sc[c] = s;
        for (int r = c + 1; r < 64; ++r) coefs[c * 64 + r] = X[r * 64 + c] * s * s;
        for (int r = c + 1; r < 64; ++r) for (int e = 0; e < 64; ++e) X[r * 64 + e] -= coefs[c * 64 + r] * X[c * 64 + e];
        if (c == 3) Xp0 = X;
    }
    float *d, *o; (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&o, 8192 * 4);
    (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int lds = (2 * 64 * SPD_LS + PIV_LDS + 8) * 4;
    hipLaunchKernelGGL(k_dbg, dim3(1), dim3(256), lds, 0, d, o);
    std::vector<float> r(8192); (void)hipMemcpy(r.data(), o, 8192 * 4, hipMemcpyDeviceToHost);
    printf("flag %g\n", r[4096 + 64]);
    for (int p = 0; p < 3; ++p) for (int t = 0; t < 4; ++t) {
        printf("x'[panel %d][%d]:", p, t);
        for (int l = 0; l < 64; ++l) printf(" %.3g", r[4096 + 128 + (p * 4 + t) * 64 + l]);
        printf("\n");
    }
    for (int rr = 4; rr < 6; ++rr) { printf("ref row %d after panel 0:", rr); for (int e = 0; e < 64; ++e) printf(" %.3g", Xp0[rr * 64 + e]); printf("\n"); }
    for (int c = 0; c < 8; ++c) {
        double e = 0; int worst = -1;
        for (int rr = 0; rr < 64; ++rr) { double dd = std::abs((double)r[c * 64 + rr] - coefs[c * 64 + rr]); if (!(dd <= e)) { e = dd; worst = rr; } }
        printf("col %2d: scale gpu %.6g ref %.6g | coef max diff %.3g at row %d (gpu %.6g ref %.6g)\n", c, r[4096 + c], sc[c], e, worst,
               worst >= 0 ? r[c * 64 + worst] : 0.f, worst >= 0 ? coefs[c * 64 + worst] : 0.f);
    }
    return 0;
}
