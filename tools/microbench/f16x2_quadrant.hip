// float32 products as two-half splits on the float16 matrix pipe (csrc/lqp_f16x2.hpp): accuracy against float64 next to
// v_mfma_f32_32x32x2_f32, and cycles per 32x32x64 quadrant product from LDS images.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lqp_py_amd/csrc tools/microbench/f16x2_quadrant.hip -o /tmp/f16x2 && /tmp/f16x2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "lqp_f16x2.hpp"
using namespace lqp;

// X, Z: 32 x 64 float32 (row-major).  out[0]: split product, out[1]: float32 matrix instruction; accumulator layout.
__global__ __launch_bounds__(64) void k_quadrant(const float* X, const float* Z, float* out, int reps, unsigned long long* cyc, int four) {
    __shared__ __attribute__((aligned(32))) char img[2 * 32 * F2_ROW];
    __shared__ __attribute__((aligned(16))) float raw[2 * 32 * 68];
    const int l = threadIdx.x, li = l & 31, lh = l >> 5;
    float inv[2];
    for (int m = 0; m < 2; ++m) {
        const float* src = m ? Z : X;
        for (int i = l; i < 32 * 64; i += 64) raw[m * 32 * 68 + (i >> 6) * 68 + (i & 63)] = src[i];
        __syncthreads();
        // lane (li, lh) converts the cells (s, lh) of row li
        float v[4][8], mx = 0.f;
        for (int s = 0; s < 4; ++s) {
            f2_load_cell_f32(raw + m * 32 * 68 + li * 68, s, lh, v[s]);
            for (int j = 0; j < 8; ++j) mx = tmax(mx, tabs(v[s][j]));
        }
        mx = wave_max(mx);
        float sc;
        f2_scale_of(mx, sc, inv[m]);
        for (int s = 0; s < 4; ++s) {
            for (int j = 0; j < 8; ++j) v[s][j] *= sc;
            h16x8 hi, mid;
            f2_split8(v[s], hi, mid);
            f2_write_cell(img + m * 32 * F2_ROW + li * F2_ROW + 64 * s + 32 * lh, hi, mid);
        }
    }
    __syncthreads();
    const char* xa = img + li * F2_ROW + 32 * lh;
    const char* zb = img + 32 * F2_ROW + li * F2_ROW + 32 * lh;
    f32x16 a = f2_quadrant<0, 4>(xa, zb);
    if (four) {       // + mid x mid
        for (int s = 0; s < 4; ++s) {
            const F2Cell ca = f2_read_cell(xa + 64 * s), cb = f2_read_cell(zb + 64 * s);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.mid, cb.mid, a, 0, 0, 0);
        }
    }
    const float un = inv[0] * inv[1];
    for (int q = 0; q < 16; ++q) out[l * 16 + q] = a[q] * un;
    // float32 reference: lane feeds row li, k = 32 lh + t
    f32x16 b;
    for (int q = 0; q < 16; ++q) b[q] = 0.f;
    for (int t = 0; t < 32; ++t)
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(raw[li * 68 + 32 * lh + t], raw[32 * 68 + li * 68 + 32 * lh + t], b, 0, 0, 0);
    for (int q = 0; q < 16; ++q) out[1024 + l * 16 + q] = b[q];
    // ---- timing: reps quadrant products back to back ----
    if (reps > 0) {
        f32x16 t0acc;
        for (int q = 0; q < 16; ++q) t0acc[q] = 0.f;
        __syncthreads();
        unsigned long long c0 = clock64();
        for (int r = 0; r < reps; ++r) {
            const f32x16 p = f2_quadrant<0, 4>(xa, zb);
            for (int q = 0; q < 16; ++q) t0acc[q] = __builtin_fmaf(p[q], -un, t0acc[q]);
        }
        unsigned long long c1 = clock64();
        f32x16 t1acc;
        for (int q = 0; q < 16; ++q) t1acc[q] = 0.f;
        for (int r = 0; r < reps; ++r) {
            f32x16 p;
            for (int q = 0; q < 16; ++q) p[q] = 0.f;
            const float* xr = raw + li * 68 + 32 * lh;
            const float* zr = raw + 32 * 68 + li * 68 + 32 * lh;
            for (int t = 0; t < 8; ++t) {
                const V4<float> av = *(const V4<float>*)(xr + 4 * t), bv = *(const V4<float>*)(zr + 4 * t);
                for (int e = 0; e < 4; ++e) p = __builtin_amdgcn_mfma_f32_32x32x2f32(av.v[e], bv.v[e], p, 0, 0, 0);
            }
            t1acc -= p;
        }
        unsigned long long c2 = clock64();
        if (l == 0) { cyc[0] = c1 - c0; cyc[1] = c2 - c1; }
        out[2048 + l] = t0acc[0] + t1acc[0];
    }
}

static double frand() { return (double)rand() / RAND_MAX; }
int main() {
    float *dX, *dZ, *dO;
    unsigned long long* dC;
    hipMalloc(&dX, 32 * 64 * 4); hipMalloc(&dZ, 32 * 64 * 4); hipMalloc(&dO, (2048 + 64) * 4); hipMalloc(&dC, 16);
    std::vector<float> X(2048), Z(2048), O(2048 + 64);
    const char* names[] = {"uniform [-1,1)", "normal-ish x 1e-3 (small block)", "wide range 10^U(-6,0)", "one huge entry (1e4) + O(1)", "tiny 1e-30", "with subnormal halves (1e-7 rel)"};
    for (int four = 0; four < 2; ++four)
    for (int dist = 0; dist < 6; ++dist) {
        srand(7 + dist);
        for (int i = 0; i < 2048; ++i) {
            double a = 2 * frand() - 1, b = 2 * frand() - 1;
            if (dist == 1) { a *= 1e-3; b *= 1e-3; }
            if (dist == 2) { a *= pow(10.0, -6 * frand()); b *= pow(10.0, -6 * frand()); }
            if (dist == 3) { if (i == 5) a = 1e4; if (i == 77) b = -1e4; }
            if (dist == 4) { a *= 1e-30; b *= 1e-30; }
            if (dist == 5) { if (i % 7) { a *= 1e-7; b *= 1e-7; } }
            X[i] = (float)a; Z[i] = (float)b;
        }
        hipMemcpy(dX, X.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(dZ, Z.data(), 8192, hipMemcpyHostToDevice);
        k_quadrant<<<1, 64>>>(dX, dZ, dO, 0, dC, four);
        hipMemcpy(O.data(), dO, 2048 * 4, hipMemcpyDeviceToHost);
        double e16 = 0, e32 = 0, ref_max = 0, s16 = 0, s32 = 0;
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * (l >> 5), col = l & 31;
            double r = 0, ra = 0;
            for (int k = 0; k < 64; ++k) { r += (double)X[row * 64 + k] * (double)Z[col * 64 + k]; ra += fabs((double)X[row * 64 + k] * (double)Z[col * 64 + k]); }
            ref_max = fmax(ref_max, fabs(r));
            const double d16 = fabs(O[l * 16 + q] - r), d32 = fabs(O[1024 + l * 16 + q] - r);
            e16 = fmax(e16, d16 / ra); e32 = fmax(e32, d32 / ra);
            s16 += d16 * d16 / (ra * ra); s32 += d32 * d32 / (ra * ra);
        }
        printf("%s  %-38s max |err| / sum|ab|: split %.2e  f32 mfma %.2e   rms: split %.2e  f32 %.2e   (max |ref| %.2e)\n",
               four ? "4 terms" : "3 terms", names[dist], e16, e32, sqrt(s16 / 1024), sqrt(s32 / 1024), ref_max);
    }
    const int reps = 2000;
    k_quadrant<<<1, 64>>>(dX, dZ, dO, reps, dC, 0);
    unsigned long long c[2];
    hipMemcpy(c, dC, 16, hipMemcpyDeviceToHost);
    printf("cycles per quadrant product (one wave, operands from LDS): split %.0f   float32 %.0f\n", (double)c[0] / reps, (double)c[1] / reps);
    return 0;
}
