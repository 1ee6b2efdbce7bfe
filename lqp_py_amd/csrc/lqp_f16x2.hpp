// float32 products on the float16 matrix pipe: every operand value a is carried as TWO halves, a s = hi + mid
// (hi = half(a s), mid = half(a s - hi): 22-24 significant bits), s a power of two per 32-row operand block chosen so that
// the block's largest entry lands just below 2^15 (nothing overflows, the residuals stay normal halves), and a product
// sum_k a_k b_k is taken as
//     sum_k  mid_a hi_b  +  hi_a mid_b  +  hi_a hi_b        (three v_mfma_f32_32x32x16_f16, float32 accumulation),
// i.e. everything but mid_a mid_b <= 2^-22 |a b|, then multiplied by 1 / (s_a s_b) (exact).  gfx950 has no reduced
// precision float32 matrix instruction (no xf32); v_mfma_f32_32x32x2_f32 runs at the vector rate (64 cycles for 4 k flop),
// the float16 instruction delivers 32 k flop in 32 cycles: three of them per 16-deep slice are 5.3 x the float32 rate at
// float32-like accuracy (measured against float64: tools/microbench/f16x2_quadrant.hip).
//
// LDS image of an operand row (64 values): eight CELLS of 32 bytes, cell (s, h) = slice s of the matrix instruction,
// lane half h: [8 halves hi | 8 halves mid] -- 256 bytes, the size of the float32 row it replaces (row stride SPD_LS
// floats = 272 bytes: conflict-free 16-byte reads by 16 consecutive lanes).  Element j of cell (s, h) is column
//     kappa(s, h, j) = 16 s + 8 (j >> 2) + 4 h + (j & 3)
// of the row: the order in which a lane of a 32x32 ACCUMULATOR holds the entries of its column (register q = row
// (q & 3) + 8 (q >> 2) + 4 h), so a product that comes out of the matrix cores transposed -- the rows of the new operand
// across the lanes -- is split and stored by the lane that holds it, sixteen bytes at a time, no lane ever needs a
// neighbour's value.  Any k order is a valid schedule as long as both operands of an instruction use the same one.
#pragma once
#include "lqp_common.hpp"

namespace lqp {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int F2_ROW = 272;          // bytes per operand row (= SPD_LS floats)

// scale of a block whose largest magnitude is mx: s = 2^(14 - floor(log2 mx)) (mx s in [2^14, 2^15)); inv = 1 / s.
// mx = 0 and magnitudes outside 2^-112 .. 2^126 take the nearest admissible exponent.
__device__ __forceinline__ void f2_scale_of(const float mx, float& s, float& inv) {
    int e = (int)((__float_as_uint(mx) >> 23) & 0xFFu);
    e = e < 15 ? 15 : (e > 253 ? 253 : e);
    s = __uint_as_float((unsigned int)(268 - e) << 23);
    inv = __uint_as_float((unsigned int)(e - 14) << 23);
}

// 8 values (already in the block's scale) -> hi, mid
__device__ __forceinline__ void f2_split8(const float (&v)[8], h16x8& hi, h16x8& mid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        hi[j] = h;
        mid[j] = (_Float16)(v[j] - (float)h);
    }
}

// the two float32 runs of cell (s, h) of a row held as 64 floats: columns 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3
__device__ __forceinline__ void f2_load_cell_f32(const float* __restrict__ row, const int s, const int h, float (&v)[8]) {
    const V4<float> a = *(const V4<float>*)(row + 16 * s + 4 * h);
    const V4<float> b = *(const V4<float>*)(row + 16 * s + 8 + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a.v[e]; v[4 + e] = b.v[e]; }
}

struct F2Cell { h16x8 hi, mid; };
__device__ __forceinline__ F2Cell f2_read_cell(const char* __restrict__ p) {
    F2Cell c;
    c.hi = *(const h16x8*)p;
    c.mid = *(const h16x8*)(p + 16);
    return c;
}
__device__ __forceinline__ void f2_write_cell(char* __restrict__ p, const h16x8& hi, const h16x8& mid) {
    *(h16x8*)p = hi;
    *(h16x8*)(p + 16) = mid;
}

// acc += A B^T over one 16-deep slice (small terms first)
__device__ __forceinline__ f32x16 f2_mma(const F2Cell& a, const F2Cell& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.mid, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}

// one 32x32 quadrant: sum over slices [S0, S1) of X rows x Z rows, both split images in LDS.  xa / zb: the LANE's cell
// address of slice 0 (image + (row0 + (l & 31)) * F2_ROW + 32 (l >> 5)).  In the scales of the two operands.
template <int S0 = 0, int S1 = 4>
__device__ __forceinline__ f32x16 f2_quadrant(const char* __restrict__ xa, const char* __restrict__ zb) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // (measured: requesting the cells of slice s + 1 before the instructions of slice s costs the ten-tile waves of the K = 8
    //  sweep another tile in scratch memory -- 0.295 against 0.280 ms; left to the compiler)
#pragma unroll
    for (int s = S0; s < S1; ++s) acc = f2_mma(f2_read_cell(xa + 64 * s), f2_read_cell(zb + 64 * s), acc);
    return acc;
}

}  // namespace lqp
