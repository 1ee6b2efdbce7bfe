#include "../../lqp_py_amd/csrc/lqp_boxqp.cuh"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace lqp;
#ifndef TAGV
#define TAGV false
#endif
__global__ __launch_bounds__(LQP_NT) void k_piv(const float* src, float* out, int reps, unsigned long long* dbg = nullptr) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    float* W = (float*)smem; float* WT = W + 64*SPD_LS; float* pcol = WT + 64*SPD_LS; int* flag = (int*)(pcol + PIV_LDS);
    if (threadIdx.x == 0) flag[0] = 0;
    for (int r = 0; r < reps; ++r) { wg_pivot_block<NWPT>(src + blockIdx.x * 4096, W, WT, pcol, flag, 0); __syncthreads(); }
    __syncthreads();
    for (int i = threadIdx.x; i < 64*64; i += LQP_NT) out[blockIdx.x*4096 + i] = W[(i>>6)*SPD_LS + (i&63)];
}
int main() {
    const int B = 128;
    std::vector<float> h(B * 4096);
    for (int b = 0; b < B; ++b) for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) {
        float v = 0.01f * std::cos(0.37f * (i + 1) * (j + 1) + b); v = (i == j) ? 2.f + 0.01f * i : v;
        h[b*4096 + i*64 + j] = (i >= j) ? v : 0.01f * std::cos(0.37f * (j + 1) * (i + 1) + b);
    }
    float *d, *o; hipMalloc(&d, h.size()*4); hipMalloc(&o, h.size()*4);
    hipMemcpy(d, h.data(), h.size()*4, hipMemcpyHostToDevice);
    const int lds = (2*64*SPD_LS + PIV_LDS + 8) * 4;
    hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
    for (int reps : {1, 1, 8, 8}) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_piv, dim3(B), dim3(LQP_NT), lds, 0, d, o, reps, (unsigned long long*)nullptr);
        hipEventRecord(b2); hipEventSynchronize(b2);
        float ms; hipEventElapsedTime(&ms, a, b2);
        printf("reps %d: %.2f us per pivot block (%.1f us total)\n", reps, ms * 1e3 / reps, ms*1e3);
    }
    std::vector<float> r(4096); hipMemcpy(r.data(), o, 4096*4, hipMemcpyDeviceToHost);
    printf("W[0][0]=%g W[63][63]=%g W[63][0]=%g\n", r[0], r[63*64+63], r[63*64]);
    return 0;
}
