// probes v_permlane32_swap and the operand layout of v_mfma_f32_32x32x2_f32 on the device (prints, no assertions)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* o) {
    const int l = threadIdx.x;
    const unsigned a = l, b = 100 + l;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    // A[row][k] = 1000 row + k (row = l % 32, k = l / 32); B[k][col] = (k == 0 ? 1 : 0.001) for col = l % 32 only at col 5
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const float av = 1000.f * (l % 32) + (l / 32);
    const float bv = (l % 32) == 5 ? ((l / 32) == 0 ? 1.f : 0.001f) : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    for (int q = 0; q < 16; ++q) o[128 + q * 64 + l] = acc[q];
}
int main() {
    float* d; (void)hipMalloc(&d, (128 + 1024) * 4);
    k<<<1, 64>>>(d);
    float h[128 + 1024]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0:"); for (int i = 0; i < 64; i += 8) printf(" %g", h[i]);
    printf("\nr1:"); for (int i = 0; i < 64; i += 8) printf(" %g", h[64 + i]);
    // column 5 of the result: D[row][5] = A[row][0] * 1 + A[row][1] * 0.001 = 1000 row + 0.001
    printf("\nlane 5 (lh 0) regs:"); for (int q = 0; q < 16; ++q) printf(" %g", h[128 + q * 64 + 5]);
    printf("\nlane 37 (lh 1) regs:"); for (int q = 0; q < 16; ++q) printf(" %g", h[128 + q * 64 + 37]);
    printf("\n");
    return 0;
}
