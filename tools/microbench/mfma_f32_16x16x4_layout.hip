// which lane / register holds D[i][j] of v_mfma_f32_16x16x4_f32?  (A[i][k] = 100 i + k, B[k][j] = (k == 0) ? j + 1 : 0 ... probes)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int mode) {
    const int l = threadIdx.x;
    // hypothesis: A lane l = A[i = l & 15][k = l >> 4], B lane l = B[k = l >> 4][j = l & 15]
    const int i = l & 15, k = l >> 4, j = l & 15;
    float a, b;
    if (mode == 0) { a = (k == 0) ? (float)(i + 1) : 0.f; b = (k == 0) ? (float)(100 * (j + 1)) : 0.f; }   // D[i][j] = (i+1) * 100 (j+1) if hypothesis right
    else { a = (float)(k + 1); b = (k == mode - 1) ? 1.f : 0.f; }                                           // D[i][j] = mode  (k-index pairing check)
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) out[l * 4 + q] = c[q];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    float h[256];
    for (int mode = 0; mode < 5; ++mode) {
        probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        if (mode == 0) {
            int ok1 = 1, ok2 = 1;
            for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
                const double v = h[l * 4 + q];
                const int jj = (int)(v / 100.0 + 0.5) / 1;   // v = (i+1) * 100 * (j+1): decode
                (void)jj;
                const int i1 = 4 * (l >> 4) + q, j1 = l & 15;          // hypothesis 1
                const int i2 = 4 * q + (l >> 4), j2 = l & 15;          // hypothesis 2
                if (v != (double)((i1 + 1) * 100 * (j1 + 1))) ok1 = 0;
                if (v != (double)((i2 + 1) * 100 * (j2 + 1))) ok2 = 0;
            }
            printf("D layout: rows 4*(l>>4)+q: %d   rows 4*q+(l>>4): %d\n", ok1, ok2);
            printf("lane 0: %g %g %g %g   lane 16: %g %g %g %g  lane 17: %g %g %g %g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67], h[68], h[69], h[70], h[71]);
        } else {
            printf("mode %d (expect %d everywhere): lane0 %g %g lane 33 %g %g\n", mode, mode, h[0], h[3], h[33 * 4], h[33 * 4 + 2]);
        }
    }
    return 0;
}
