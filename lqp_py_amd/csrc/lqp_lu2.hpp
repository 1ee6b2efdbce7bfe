// Batched LU with partial pivoting shared by TWO workgroups per matrix (small batches: 2 B <= #CUs, the second half
// of the chip would idle).  Same algorithm, pivot rule, LAPACK layout and -- per element -- the same arithmetic as
// wg_lu_factor (lqp_lu.hpp): replaces torch.linalg.lu_factor at lqp_py/solve_box_qp_admm_torch.py:215,254 and
// lqp_py/lu_layer.py:10,31.
//
// Column blocks of PB columns are dealt out alternately (block c belongs to workgroup c & 1).  Updates of different
// columns never depend on each other, so all that travels is the factored panel:
//   owner of panel k+1   receives panel k, applies its interchanges / U12 / update to the PB columns of block k+1 only,
//                        factors panel k+1 in registers and PUBLISHES it (write-through stores of the panel at its
//                        final position in the matrix itself + a 100-word message: gather map of the interchanges,
//                        first zero pivot), THEN brings the rest of its columns up to date with panel k;
//   the other workgroup  meanwhile runs panel k's interchanges / U12 / trailing update on ITS columns.
// A classical look-ahead of depth one with the panel chain on alternating CUs.  One direction per hand-off, no
// acknowledgement: the owner of panel k+2 is the workgroup that consumed panel k before it published k+1, so neither the
// message slots (two, by panel parity) nor the panel's columns are rewritten before they are read.
// Hand-off form: MI355X_MICROARCH.md, R1 (every handed-off byte stored sc1, every storing wave drains, barrier, ONE flag
// store; the reader polls the flag with an sc1 load, barrier, sc1 loads of the bytes).
//
// Panel factorisation: 256 threads (ONE wave per SIMD), two rows per thread -- the column step is a dependent chain of
// ~140 instructions; two waves per SIMD interleave two copies of that chain and take twice as long (lqp_lu.hpp:
// 2027 cycles per column with eight waves).  The other four waves only join the barriers.
#pragma once
#include "lqp_lu.hpp"

namespace lqp {

constexpr int LU2_NT = 512;
constexpr int LU2_NW = LU2_NT / 64;
constexpr int LU2_PW = 4;                    // waves that hold panel rows
constexpr int LU2_MSG = 128;                 // ints per message: [0] ne, [1] first zero pivot (1-based; 0: none), [2, 2+PB) src of the new
                                             // top rows, [34, 34+PB) xdst, [66, 66+PB) xsrc  (PB <= 32)
constexpr int LU2_SCR_WORDS = 8 + LU2_MSG;   // 64-bit words per problem: [0], [1] panels published by workgroup 0 / 1; [2], [3] their XCDs
                                             // {launch epoch : 32 | count : 32}; [8..): two messages of LU2_MSG ints

template <typename T, int PB> struct Lu2Lds {
    int lt, up, l11, lit, rowp, wval, wrcp, widx, wtid, pidx, src, xdst, xsrc, pxdst, pxsrc, cnt, total;
    __host__ __device__ explicit Lu2Lds(int Mpad) {
        int o = 0;
        lt = o;   o += PB * Mpad * (int)sizeof(T);
        up = o;   o += PB * Mpad * (int)sizeof(T);
        l11 = o;
        lit = o;  o += PB * PB * (int)sizeof(T);       // (L11^-1)^T: [k * PB + i] = Linv[i][k], the A operand of the U12 products
        rowp = o; o += 2 * LU2_PW * PB * (int)sizeof(T);
        wval = o; o += round_up(2 * LU2_PW * (int)sizeof(T), 32);
        wrcp = o; o += round_up(2 * LU2_PW * (int)sizeof(T), 32);
        widx = o; o += round_up(2 * LU2_PW * 4, 32);
        wtid = o; o += round_up(2 * LU2_PW * 4, 32);
        pidx = o; o += round_up(PB * 4, 32);
        src = o;  o += round_up(PB * 4, 32);
        xdst = o; o += round_up(PB * 4, 32);
        xsrc = o; o += round_up(PB * 4, 32);
        pxdst = o; o += round_up(PB * 4, 32);    // the same two lists of the panel being PUBLISHED (the received panel's are still
        pxsrc = o; o += round_up(PB * 4, 32);    // in use: its interchanges on the other columns follow the publication)
        cnt = o;  o += 32;                       // [0] displaced rows of the panel being published, [1] first zero pivot, [2] timeout
        total = o;
    }
};

// ---- write-through (sc1) stores / L1-bypassing (sc1) loads of handed-off bytes ----
__device__ __forceinline__ void st_sc1(int* p, const int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, const float v) {
    __hip_atomic_store((unsigned int*)p, __builtin_bit_cast(unsigned int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, const double v) {
    __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 16 bytes, one write-through store instruction (a 4-byte sc1 store is a fabric write of its own: ~6x the time per byte)
__device__ __forceinline__ void st16_sc1(void* p, const u32x4 v) {
    // (s_nop: a store of more than 64 bits reads its data registers over several cycles and the next instruction must not
    //  write them meanwhile -- the hazard recogniser does not see into inline assembly; without it the first 8 bytes of some
    //  rows arrived as the NEXT store's address)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_vec_sc1(float* p, const V4<float>& v) {
    st16_sc1(p, u32x4{__builtin_bit_cast(unsigned int, v.v[0]), __builtin_bit_cast(unsigned int, v.v[1]),
                      __builtin_bit_cast(unsigned int, v.v[2]), __builtin_bit_cast(unsigned int, v.v[3])});
}
__device__ __forceinline__ void st_vec_sc1(double* p, const V4<double>& v) {
    const unsigned long long a = __builtin_bit_cast(unsigned long long, v.v[0]), b = __builtin_bit_cast(unsigned long long, v.v[1]);
    const unsigned long long c = __builtin_bit_cast(unsigned long long, v.v[2]), d = __builtin_bit_cast(unsigned long long, v.v[3]);
    st16_sc1(p, u32x4{(unsigned int)a, (unsigned int)(a >> 32), (unsigned int)b, (unsigned int)(b >> 32)});
    st16_sc1(p + 2, u32x4{(unsigned int)c, (unsigned int)(c >> 32), (unsigned int)d, (unsigned int)(d >> 32)});
}
// (8-byte agent-scope loads: the compiler counts them, unlike loads issued from inline assembly)
__device__ __forceinline__ V4<float> ld_vec_sc1(const float* p) {
    const unsigned long long a = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    V4<float> v;
    v.v[0] = __uint_as_float((unsigned int)a); v.v[1] = __uint_as_float((unsigned int)(a >> 32));
    v.v[2] = __uint_as_float((unsigned int)b); v.v[3] = __uint_as_float((unsigned int)(b >> 32));
    return v;
}
__device__ __forceinline__ V4<double> ld_vec_sc1(const double* p) {
    V4<double> v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v.v[e] = ld_sc1(p + e);
    return v;
}

// one 128-byte row another workgroup has published: eight 16-byte sc1 loads (past this CU's L1) in flight at once.  Issued
// from inline assembly (the compiler has no 16-byte agent-scope load), so the wait is explicit and carries the registers.
__device__ __forceinline__ void ld_row128_sc1(const void* p, u32x4 (&v)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
                 "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p) : "memory");
}
template <typename T, int PB> __device__ __forceinline__ void row128_unpack(const u32x4 (&v)[8], T (&r)[PB]) {
    static_assert(PB * sizeof(T) == 128, "a panel row is 128 bytes");
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            const unsigned int u = v[c >> 2][c & 3];      // (a bit_cast straight from the vector element yields element 0: hipcc 7.2)
            r[c] = __uint_as_float(u);
        }
    } else {
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            const unsigned int lo = v[c >> 1][2 * (c & 1)], hi = v[c >> 1][2 * (c & 1) + 1];
            r[c] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
        }
    }
}

// f64 trailing update on the matrix cores: v_mfma_f64_16x16x4_f64, D = C - L21 U12 over one 16x16 tile.
//   A operand (16x4 slice of L21): lane l holds L21[i = l & 15][k = l >> 4]
//   B operand (4x16 slice of U12): lane l holds U12[k = l >> 4][j = l & 15]
//   C/D: lane l holds rows 4 q + (l >> 4), q = 0..3, of column l & 15 (measured: tools/microbench/mfma_f64_layout.hip)
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Trailing update of the tile columns this workgroup owns.  A22: the M2 x M2 trailing matrix (row stride ld); LT / UP:
// L21^T and U12 in LDS (row kk at kk * Mpad); tile columns tj with ((tj + tj_par) & 1) == 0 and tj >= tj_first are taken.
// f32: 32x32 tiles (PB == 32 columns = one block per tile column); f64: 16x16 tiles (PB == 16).
template <typename T, int PB>
__device__ __forceinline__ void lu2_trailing(T* __restrict__ A22, const int ld, const int M2, const T* __restrict__ LT,
                                             const T* __restrict__ UP, const int Mpad, const int tj_par, const int tj_first,
                                             const int only_tj = -1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if constexpr (sizeof(T) == 4) {
        static_assert(sizeof(T) == 8 || PB == 32, "one 32-column tile per block");
        const int li = lane & 31, lh = lane >> 5;
        const int nt = (M2 + 31) >> 5;
        // my tile columns: tj = tj0, tj0 + 2, ...
        int tj0 = tj_par & 1;                 // smallest tj with (tj + tj_par) even
        while (tj0 < tj_first) tj0 += 2;
        int ncol = tj0 < nt ? (nt - tj0 + 1) >> 1 : 0;
        if (only_tj >= 0) { tj0 = only_tj; ncol = 1; }
        const int ntiles = nt * ncol;
        const int voff = 4 * lh * ld + li;
        for (int t = __builtin_amdgcn_readfirstlane(w); t < ntiles; t += LU2_NW) {
            const int ti = t / ncol, tj = tj0 + 2 * (t - ti * ncol);
            const int i0 = ti << 5, j0 = tj << 5;
            float* base = (float*)A22 + (size_t)i0 * ld + j0;
            const bool colok = j0 + li < M2;
            const int rlim = M2 - i0 - 4 * lh;
            f32x16 cur;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int qrow = (q & 3) + 8 * (q >> 2);
                cur[q] = (colok && qrow < rlim) ? base[(size_t)qrow * ld + voff] : 0.f;
            }
            const float* lt = (const float*)LT + i0 + li + lh * Mpad;
            const float* up = (const float*)UP + j0 + li + lh * Mpad;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int kk = 0; kk < PB; kk += 2) {
                const float a = lt[kk * Mpad];
                const float b = up[kk * Mpad];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            cur -= acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int qrow = (q & 3) + 8 * (q >> 2);
                if (colok && qrow < rlim) base[(size_t)qrow * ld + voff] = cur[q];
            }
        }
    } else {
        static_assert(sizeof(T) == 4 || PB == 16, "one 16-column tile per block");
        const int li = lane & 15, lh = lane >> 4;
        const int nt = (M2 + 15) >> 4;
        int tj0 = tj_par & 1;
        while (tj0 < tj_first) tj0 += 2;
        int ncol = tj0 < nt ? (nt - tj0 + 1) >> 1 : 0;
        if (only_tj >= 0) { tj0 = only_tj; ncol = 1; }
        const int ntiles = nt * ncol;
        for (int t = __builtin_amdgcn_readfirstlane(w); t < ntiles; t += LU2_NW) {
            const int ti = t / ncol, tj = tj0 + 2 * (t - ti * ncol);
            const int i0 = ti << 4, j0 = tj << 4;
            double* base = (double*)A22 + (size_t)(i0 + lh) * ld + j0 + li;
            const bool colok = j0 + li < M2;
            const int rlim = M2 - i0 - lh;            // register q holds row 4 q + lh (measured: tools/microbench/mfma_f64_layout.hip)
            f64x4 cur;
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = (colok && 4 * q < rlim) ? base[(size_t)(4 * q) * ld] : 0.0;
            const double* lt = (const double*)LT + i0 + li + lh * Mpad;
            const double* up = (const double*)UP + j0 + li + lh * Mpad;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < PB; kk += 4) {
                const double a = lt[kk * Mpad];
                const double b = up[kk * Mpad];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            cur -= acc;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (colok && 4 * q < rlim) base[(size_t)(4 * q) * ld] = cur[q];
        }
    }
}

// (L11^-1)^T by the wave whose lanes 0 .. PB-1 hold the rows of the panel's top block (lane i, register j: L[i][j]): lane c
// carries column c of the inverse of the unit lower triangle; L[i][j] reaches every lane as a scalar (v_readlane), so the
// substitution is 2 instructions per term with no memory operand (through LDS broadcast reads the 496 terms of a 32-wide
// panel waited ~240 times for the LDS: 23 k cycles per panel).  The U12 of every column then is a plain product, and those
// run on the matrix cores.
template <typename T, int PB>
__device__ __forceinline__ void lu2_invert_l11(const T (&r)[PB], T* __restrict__ LIT) {
    const int c = threadIdx.x & 63;
    T x[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) x[i] = (i == c) ? T(1) : T(0);
#pragma unroll
    for (int j = 0; j < PB - 1; ++j) {
        const T xj = x[j];
#pragma unroll
        for (int i = j + 1; i < PB; ++i) x[i] -= readlane_t(r[j], i) * xj;
    }
    if (c < PB) {
#pragma unroll
        for (int i = 0; i < PB; ++i) LIT[c * PB + i] = x[i];
    }
}

// One tile = the PB columns of one block.  Lane l: column l % PB, k-group l / PB (G = 64 / PB groups = the k-depth of one
// matrix instruction: f32 32x32x2, f64 16x16x4).
template <typename T, int PB> struct Lu2Tile {
    static constexpr int G = 64 / PB;
    T b[PB / G];             // rows lu2_tile_row(kk, group) of (P A)12, this lane's column
    int cb;                  // block (-1: none)
    bool right;
};
// f64: the B operand of v_mfma_f64_16x16x4 (row 4 kk + group); f32: the ACCUMULATOR layout of v_mfma_f32_32x32x2 (register kk of
// lane group g: row (kk & 3) + 8 (kk >> 2) + 4 g) -- the f32 tile is solved in place, see lu2_tile_finish
template <typename T, int PB> __device__ __forceinline__ constexpr int lu2_tile_row(const int kk, const int lg) {
    return sizeof(T) == 4 ? (kk & 3) + 8 * (kk >> 2) + 4 * lg : (64 / PB) * kk + lg;
}

// interchanges of panel k on the tile's columns: the new top rows into registers (gather map `src`), the displaced top rows
// down to their new places.  Only this wave touches these columns.
template <typename T, int PB>
__device__ __forceinline__ void lu2_tile_load(Lu2Tile<T, PB>& t, T* __restrict__ A, const int ld, const int N, const int k0,
                                              const int pb, const int ne, const int* __restrict__ src,
                                              const int* __restrict__ xdst, const int* __restrict__ xsrc) {
    constexpr int G = 64 / PB;
    if (t.cb < 0) return;
    const int lane = threadIdx.x & 63, li = lane % PB, lg = lane / PB;
    const int col = t.cb * PB + li;
    const bool colok = col < N;
    T* Ac = A + (size_t)k0 * ld + col;
#pragma unroll
    for (int kk = 0; kk < PB / G; ++kk) {
        const int kr = lu2_tile_row<T, PB>(kk, lg);
        t.b[kk] = (kr < pb && colok) ? Ac[(size_t)src[kr] * ld] : T(0);
    }
    // (all old rows first -- at most PB / G per lane --, ONE wait, then the stores: a wait per chunk cost a memory round trip each)
    T ext[PB / G];
#pragma unroll
    for (int q = 0; q < PB / G; ++q) {
        const int e = G * q + lg;
        ext[q] = (e < ne && colok) ? Ac[(size_t)xsrc[e] * ld] : T(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (every load of the old rows before the first store)
#pragma unroll
    for (int q = 0; q < PB / G; ++q) {
        const int e = G * q + lg;
        if (e < ne && colok) Ac[(size_t)xdst[e] * ld] = ext[q];
    }
}

// right of the panel: U12 = L11^-1 (P A)12 on the matrix cores, to the matrix and to UP (row i at i * Mpad, column relative
// to k1); left of it: the new top rows go back as they are
template <typename T, int PB>
__device__ __forceinline__ void lu2_tile_finish(const Lu2Tile<T, PB>& t, T* __restrict__ A, const int ld, const int N, const int k0,
                                                const int pb, const int* __restrict__ src, const T* __restrict__ LIT,
                                                T* __restrict__ UP, const int Mpad) {
    constexpr int G = 64 / PB;
    if (t.cb < 0) return;
    const int lane = threadIdx.x & 63, li = lane % PB, lg = lane / PB;
    const int col = t.cb * PB + li;
    const bool colok = col < N;
    T* Ac = A + (size_t)k0 * ld + col;
    if (t.right) {
        const int ucol = col - (k0 + pb);
        if constexpr (sizeof(T) == 4) {
            // Forward substitution IN the accumulators, two rows per matrix instruction: rows 2p, 2p+1 of a column sit in
            // registers q0, q0 + 1 of one lane (group (p >> 1) & 1); row 2p+1 takes its one term on the vector unit, ONE
            // v_permlane32_swap turns the two finished rows into the B operand (row 2p in lanes 0-31, row 2p+1 in lanes 32-63)
            // and a single v_mfma_f32_32x32x2 subtracts L[:, 2p : 2p+2] times them from every row below.  LIT here holds L11
            // itself ([k * PB + i] = L[i][k]): no inverse (one wave inverting a 32 x 32 triangle, whether against LDS broadcast
            // reads or v_readlane scalars, took 23-25 k cycles per panel: ~5 instructions per term, 496 terms).
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = t.b[q];
            // (every operand that comes from LDS first: inside the chain each read would be a round trip of its own)
            float leff[PB / 2], aop[PB / 2];
#pragma unroll
            for (int p = 0; p < PB / 2; ++p) {
                const int lgp = (p >> 1) & 1;
                const float lp = LIT[(2 * p) * PB + 2 * p + 1];
                leff[p] = (lg == lgp) ? -lp : 0.f;
                const float lraw = LIT[(2 * p + lg) * PB + li];
                aop[p] = (li > 2 * p + 1) ? -lraw : 0.f;
            }
#pragma unroll
            for (int p = 0; p < PB / 2; ++p) {
                const int lgp = (p >> 1) & 1, q0 = ((2 * p) & 3) + 4 * (p >> 2);
                acc[q0 + 1] = __builtin_fmaf(leff[p], acc[q0], acc[q0 + 1]);
                if (p < PB / 2 - 1) {
                    float a0 = acc[q0], a1 = acc[q0 + 1];
                    lane_swap32(a0, a1);                          // a0 = [a0 lanes 0-31 | a1 lanes 0-31], a1 = [a0 lanes 32-63 | a1 lanes 32-63]
                    const float bop = lgp == 0 ? a0 : a1;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[p], bop, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * lg;
                UP[i * Mpad + ucol] = acc[q];
                if (colok) Ac[(size_t)i * ld] = acc[q];
            }
        } else {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < PB / G; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(LIT[(G * kk + lg) * PB + li], t.b[kk], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 4 * q + lg;
                UP[i * Mpad + ucol] = acc[q];
                if (colok) Ac[(size_t)i * ld] = acc[q];
            }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < PB / G; ++kk) {
            const int kr = lu2_tile_row<T, PB>(kk, lg);
            if (kr < pb && colok && src[kr] != kr) Ac[(size_t)kr * ld] = t.b[kk];
        }
    }
}

template <typename T> struct Panel2Lds {
    T* rowP; T* wval; T* wrcp; int* widx; int* wtid; int* pidx; int* cnt;
};

// The PB columns of one panel on LU2_PW waves: row a = relative row t and -- TWO: panels of more than 256 rows -- row b =
// t + 256.  Same per-element arithmetic and the same pivot rule as lu_panel_columns (max |.|, ties to the smallest position).
// Every wave of the workgroup calls it (the barrier per column is the workgroup's); waves >= LU2_PW only synchronise.
// The bookkeeping of positions is branch-free (selects); the pivot index / first zero pivot are noted by thread 0 (every lane
// knows them).
template <typename T, int PB, bool TWO>
__device__ __forceinline__ void lu2_panel_columns(V4<T> (&ra)[PB / 4], V4<T> (&rb)[PB / 4], int& posa, int& posb, bool& donea,
                                                  bool& doneb, const int pb, const int k0, const Panel2Lds<T>& S) {
    typedef V4<T> vec;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool pw = w < LU2_PW;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        if (j < pb) {
            const int par = j & 1;
            T* cand = S.rowP + par * (LU2_PW * PB);
            T* wv = S.wval + par * LU2_PW;
            int* wi = S.widx + par * LU2_PW;
            int* wt = S.wtid + par * LU2_PW;
            T* wr = S.wrcp + par * LU2_PW;
            if (pw) {
                const T aa = ra[j >> 2].v[j & 3];
                const T ka = donea ? T(-1) : tabs(aa);
                bool pickb = false;
                T key = ka, apick = aa;
                int cpos = posa;
                if constexpr (TWO) {
                    const T ab = rb[j >> 2].v[j & 3];
                    const T kb = doneb ? T(-1) : tabs(ab);
                    pickb = kb > ka || (kb == ka && posb < posa);
                    key = pickb ? kb : ka;
                    cpos = pickb ? posb : posa;
                    apick = pickb ? ab : aa;
                }
                T myrcp = T(1) / apick;
                asm volatile("" : "+v"(myrcp));
                T bw; int lb;
                wave_argmax(key, bw, lb);
                const unsigned long long tied = __ballot(key == bw);
                if (__popcll(tied) > 1) {
                    int cp = (key == bw) ? cpos : 0x7fffffff;
                    cp = row16_min_i32(cp);
                    cp = min(min(__builtin_amdgcn_readlane(cp, 0), __builtin_amdgcn_readlane(cp, 16)),
                             min(__builtin_amdgcn_readlane(cp, 32), __builtin_amdgcn_readlane(cp, 48)));
                    lb = __ffsll((unsigned long long)__ballot(key == bw && cpos == cp)) - 1;
                }
                if (lane == lb) {
                    wv[w] = bw;
                    wr[w] = myrcp;
                    wi[w] = cpos;
                    wt[w] = pickb ? tid + 256 : tid;
                }
                // (two separate branches: as an if / else the compiler stores (pickb ? rb : ra)[q] through a selected POINTER and
                //  both rows live in scratch memory for the whole kernel)
                if (lane == lb && !pickb) {
#pragma unroll
                    for (int q = 0; q < PB / 4; ++q) *(vec*)(cand + w * PB + 4 * q) = ra[q];
                }
                if constexpr (TWO) {
                    asm volatile("" ::: "memory");
                    if (lane == lb && pickb) {
#pragma unroll
                        for (int q = 0; q < PB / 4; ++q) *(vec*)(cand + w * PB + 4 * q) = rb[q];
                    }
                }
            }
            __syncthreads();
            if (pw) {
                const T cv = wv[lane & 3];
                const int ci = wi[lane & 3];
                const int ct = wt[lane & 3];
                T best = hwmax(cv, dpp<0xB1>(cv));
                best = hwmax(best, dpp<0x4E>(best));
                const unsigned long long tied = __ballot(cv == best) & 0xFull;
                int ww = __ffsll(tied) - 1;
                if (__popcll(tied) > 1) {
                    int cp = (cv == best) ? ci : 0x7fffffff;
                    cp = min(cp, dpp_i32<0xB1>(cp));
                    cp = min(cp, dpp_i32<0x4E>(cp));
                    ww = __ffsll((unsigned long long)__ballot(cv == best && ci == cp) & 0xFull) - 1;
                }
                const int pivpos = __builtin_amdgcn_readlane(ci, ww);
                const int bi = __builtin_amdgcn_readlane(ct, ww);
                const T* rowPc = cand + ww * PB;
                vec pr[PB / 4];
#pragma unroll
                for (int q = j >> 2; q < PB / 4; ++q) pr[q] = *(const vec*)(rowPc + 4 * q);
                const T rinv = wr[ww];
                const bool nz = best > T(0);
                if (tid == 0) {
                    S.pidx[j] = pivpos;
                    if (!nz && S.cnt[1] == 0) S.cnt[1] = k0 + j + 1;
                }
                {
                    const bool isp = tid == bi;
                    posa = isp ? j : (posa == j ? pivpos : posa);
                    donea = donea || isp;
                    const bool upd = !donea && nz;
                    const T aj = ra[j >> 2].v[j & 3];
                    const T l = upd ? aj * rinv : T(0);
                    ra[j >> 2].v[j & 3] = upd ? l : aj;
#pragma unroll
                    for (int c = j + 1; c < PB; ++c) ra[c >> 2].v[c & 3] -= l * pr[c >> 2].v[c & 3];
                }
                if constexpr (TWO) {
                    const bool isp = tid + 256 == bi;
                    posb = isp ? j : (posb == j ? pivpos : posb);
                    doneb = doneb || isp;
                    const bool upd = !doneb && nz;
                    const T aj = rb[j >> 2].v[j & 3];
                    const T l = upd ? aj * rinv : T(0);
                    rb[j >> 2].v[j & 3] = upd ? l : aj;
#pragma unroll
                    for (int c = j + 1; c < PB; ++c) rb[c >> 2].v[c & 3] -= l * pr[c >> 2].v[c & 3];
                }
            }
        }
    }
}

// rows of a panel (positions k0 + t and k0 + t + 256, columns k0 .. k0 + pb) -> registers
template <typename T, int PB>
__device__ __forceinline__ void lu2_load_rows(const T* __restrict__ A, const int ld, V4<T> (&ra)[PB / 4], V4<T> (&rb)[PB / 4],
                                              const int k0, const int pb, const int M) {
    typedef V4<T> vec;
    const int tid = threadIdx.x;
    const bool acta = tid < 256 && tid < M, actb = tid < 256 && tid + 256 < M;
    if (pb == PB) {
#pragma unroll
        for (int q = 0; q < PB / 4; ++q) {
            if (acta) ra[q] = *(const vec*)(A + (size_t)(k0 + tid) * ld + k0 + 4 * q);
            else { ra[q].v[0] = ra[q].v[1] = ra[q].v[2] = ra[q].v[3] = T(0); }
            if (actb) rb[q] = *(const vec*)(A + (size_t)(k0 + tid + 256) * ld + k0 + 4 * q);
            else { rb[q].v[0] = rb[q].v[1] = rb[q].v[2] = rb[q].v[3] = T(0); }
        }
    } else {
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            ra[c >> 2].v[c & 3] = (acta && c < pb) ? A[(size_t)(k0 + tid) * ld + k0 + c] : T(0);
            rb[c >> 2].v[c & 3] = (actb && c < pb) ? A[(size_t)(k0 + tid + 256) * ld + k0 + c] : T(0);
        }
    }
}

template <typename T, int PB>
__device__ __forceinline__ void lu2_put_row(T* __restrict__ A, const int ld, const V4<T> (&row)[PB / 4], const int pos, const int rel,
                                            const int k0, const int pb, int* __restrict__ msg, int* __restrict__ cnt,
                                            int* __restrict__ xdst, int* __restrict__ xsrc, const bool xlocal) {
    T* dst = A + (size_t)(k0 + pos) * ld + k0;
    if (pb == PB && xlocal) {
        // both workgroups on ONE XCD: its L2 is their point of coherence -- plain stores (the lines stay in that L2, the reader's
        // sc1 loads find them there; a write-through store drops the line and the reader goes to memory)
#pragma unroll
        for (int q = 0; q < PB / 4; ++q) *(V4<T>*)(dst + 4 * q) = row[q];
    } else if (pb == PB) {
#pragma unroll
        for (int q = 0; q < PB / 4; ++q) st_vec_sc1(dst + 4 * q, row[q]);
    } else {
#pragma unroll
        for (int c = 0; c < PB; ++c)
            if (c < pb) st_sc1(dst + c, row[c >> 2].v[c & 3]);
    }
    if (pos < pb) st_sc1(msg + 2 + pos, rel);
    if (pos >= pb && pos != rel) {
        const int q = atomicAdd(cnt, 1);
        xdst[q] = pos;
        xsrc[q] = rel;
    }
}

// The factored panel kk leaves the registers: rows to their final positions in the matrix (write-through), the message, the
// pivots; then the flag.  cnt[0] == 0 on entry and on exit.  This CU's L1 may still hold the panel's columns as they were
// BEFORE the factorisation (plain loads), and a write-through store does not refresh it: the L1 is dropped afterwards.
template <typename T, int PB>
__device__ __forceinline__ void lu2_publish(T* __restrict__ A, const int ld, int* __restrict__ ipiv, const V4<T> (&ra)[PB / 4],
                                            const V4<T> (&rb)[PB / 4], const int posa, const int posb, const int kk, const int k0,
                                            const int pb, const int M, int* __restrict__ msg_base, int* __restrict__ cnt,
                                            const int* __restrict__ pidx, int* __restrict__ xdst, int* __restrict__ xsrc,
                                            unsigned long long* __restrict__ flag, const unsigned int epoch, const bool xlocal) {
    const int tid = threadIdx.x;
    int* msg = msg_base + (kk & 1) * LU2_MSG;
    if (tid < 256) {
        if (tid < M) lu2_put_row<T, PB>(A, ld, ra, posa, tid, k0, pb, msg, cnt, xdst, xsrc, xlocal);
        if (tid + 256 < M) lu2_put_row<T, PB>(A, ld, rb, posb, tid + 256, k0, pb, msg, cnt, xdst, xsrc, xlocal);
    }
    __syncthreads();
    if (tid < pb) ipiv[k0 + tid] = k0 + pidx[tid] + 1;
    const int ne = cnt[0];
    if (tid < ne) { st_sc1(msg + 34 + tid, xdst[tid]); st_sc1(msg + 66 + tid, xsrc[tid]); }
    if (tid == 0) { st_sc1(msg, ne); st_sc1(msg + 1, cnt[1]); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        cnt[0] = 0;
        __hip_atomic_store(flag, ((unsigned long long)epoch << 32) | (unsigned long long)(kk + 1), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (!xlocal) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!xlocal) __syncthreads();
}

// dbg (optional, workgroup 0 of the matrix): cycles of wave 0 -- [0] waiting for the partner, [1] receive (message, L panel),
// [2] the next panel's columns (interchanges, U12, update in registers), [3] panel factorisation, [4] publish,
// [5] interchanges + U12 (after the inverse / loads / barrier: [8] inverse of L11, [9] tile loads, [10] barrier), [6] trailing
// update, [7] total, [11] both workgroups on one XCD
template <typename T, int PB>
__device__ __forceinline__ void wg_lu_factor2(T* __restrict__ A, const int N, const int ld, int* __restrict__ ipiv,
                                              int* __restrict__ info, char* __restrict__ smem, const int me,
                                              unsigned long long* __restrict__ scr, const unsigned int epoch,
                                              const bool xlocal_ok, unsigned long long* __restrict__ dbg) {
    typedef V4<T> vec;
    const int Mpad = round_up(N, 64);
    const Lu2Lds<T, PB> L(Mpad);
    T* LT = (T*)(smem + L.lt);
    T* UP = (T*)(smem + L.up);
    T* LIT = (T*)(smem + L.lit);
    int* pidx = (int*)(smem + L.pidx);
    int* src = (int*)(smem + L.src);
    int* xdst = (int*)(smem + L.xdst);
    int* xsrc = (int*)(smem + L.xsrc);
    int* pxdst = (int*)(smem + L.pxdst);
    int* pxsrc = (int*)(smem + L.pxsrc);
    int* cnt = (int*)(smem + L.cnt);
    const Panel2Lds<T> S{(T*)(smem + L.rowp), (T*)(smem + L.wval), (T*)(smem + L.wrcp), (int*)(smem + L.widx),
                         (int*)(smem + L.wtid), pidx, cnt};
    const int tid = threadIdx.x;
    const int nblk = (N + PB - 1) / PB;
    int* const msg_base = (int*)(scr + 8);
    unsigned long long dbt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dt0 = 0, dt_all = 0;
    const bool dbg_on = dbg != nullptr;
    if (dbg_on) dt_all = clock64();
#define LU2_STAMP(i) do { if (dbg_on) { const unsigned long long t_ = clock64(); dbt[i] += t_ - dt0; dt0 = t_; } } while (0)

    if (tid < 8) cnt[tid] = 0;
    if (tid < 2 * LU2_PW) S.wval[tid] = T(-2);
    // Which XCD is the partner on?  Each workgroup announces its own (one write-through word) and reads the other's.
    if (tid == 0) {
        const unsigned int xme = (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;       // HW_REG_XCC_ID
        const unsigned long long ann = ((unsigned long long)epoch << 8) | 0x80ull;
        __hip_atomic_store(scr + 2 + me, ann | xme, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long g = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (((g = __hip_atomic_load(scr + 2 + (1 - me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~0xFull) != ann) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ULL) { cnt[2] = 1; g = ~0ull; break; }
        }
        cnt[4] = (xlocal_ok && (unsigned int)(g & 0xFull) == xme) ? 1 : 0;
    }
    __syncthreads();
    const bool xlocal = cnt[4] != 0;

    // ---- prologue: workgroup 0 factors panel 0 ----
    if (me == 0) {
        if (dbg_on) dt0 = clock64();
        const int pb = N < PB ? N : PB;
        vec ra[PB / 4], rb[PB / 4];
        lu2_load_rows<T, PB>(A, ld, ra, rb, 0, pb, N);
        int posa = tid, posb = tid + 256;
        bool donea = !(tid < 256 && tid < N), doneb = !(tid < 256 && tid + 256 < N);
        if (N > 256) lu2_panel_columns<T, PB, true>(ra, rb, posa, posb, donea, doneb, pb, 0, S);
        else lu2_panel_columns<T, PB, false>(ra, rb, posa, posb, donea, doneb, pb, 0, S);
        LU2_STAMP(3);
        lu2_publish<T, PB>(A, ld, ipiv, ra, rb, posa, posb, 0, 0, pb, N, msg_base, cnt, pidx, pxdst, pxsrc, scr + me, epoch, xlocal);
        LU2_STAMP(4);
    }

    for (int k = 0; k < nblk; ++k) {
        const int k0 = k * PB;
        const int pb = (N - k0 < PB) ? (N - k0) : PB;
        const int M = N - k0, M2 = M - pb;
        const bool mine = (k & 1) == me;
        if (dbg_on) dt0 = clock64();
        // ---- receive panel k ----
        if (!mine && tid == 0) {
            const unsigned long long want = ((unsigned long long)epoch << 32) | (unsigned long long)(k + 1);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned long long got = __hip_atomic_load(scr + (1 - me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((got >> 32) == (unsigned long long)epoch && got >= want) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ULL) { cnt[2] = 1; break; }      // 1 s: give up, flagged below
            }
        }
        __syncthreads();
        LU2_STAMP(0);
        {
            const int* msg = msg_base + (k & 1) * LU2_MSG;
            if (tid < pb) src[tid] = ld_sc1(msg + 2 + tid);
            if (tid < PB) { xdst[tid] = ld_sc1(msg + 34 + tid); xsrc[tid] = ld_sc1(msg + 66 + tid); }
            if (tid == 0) { cnt[3] = ld_sc1(msg); if (!mine && cnt[1] == 0) cnt[1] = ld_sc1(msg + 1); }
            if (tid < M) {
                const T* rowp = A + (size_t)(k0 + tid) * ld + k0;
                T r[PB];
                if (pb == PB && mine && xlocal) {      // (my own plain stores: this CU's L1 / the XCD's L2 have them)
#pragma unroll
                    for (int q = 0; q < PB / 4; ++q) {
                        const vec v4 = *(const vec*)(rowp + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[4 * q + e] = v4.v[e];
                    }
                } else if (pb == PB) {
                    u32x4 raw[8];
                    ld_row128_sc1(rowp, raw);
                    row128_unpack<T, PB>(raw, r);
                } else {
#pragma unroll
                    for (int c = 0; c < PB; ++c) r[c] = (c < pb) ? ld_sc1(rowp + c) : T(0);
                }
                if (tid >= pb) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) LT[c * Mpad + (tid - pb)] = r[c];
                }
                // (wave 0: its lanes 0 .. pb-1 hold the rows of the top block)
                if constexpr (sizeof(T) == 4) {
                    if (tid < pb && pb == PB) {
#pragma unroll
                        for (int c = 0; c < PB; ++c) LIT[c * PB + tid] = r[c];       // L11 itself, [k * PB + i] = L[i][k]
                    }
                } else {
                    if (tid < 64 && pb == PB && k0 + pb < N) lu2_invert_l11<T, PB>(r, LIT);
                }
            }
        }
        __syncthreads();
        const int ne = __builtin_amdgcn_readfirstlane(cnt[3]);
        // (any row interchanged at all?  lane j looks at src[j]: one LDS read and a ballot per wave, the same answer in every wave)
        const int lane_ = tid & 63;
        const bool anyswap = ne > 0 || __ballot(lane_ < pb && src[lane_ < pb ? lane_ : 0] != lane_) != 0ull;
        LU2_STAMP(1);

        const int k1 = k0 + pb;
        const bool next_mine = (k + 1 < nblk) && (((k + 1) & 1) == me);
        // ---- interchanges + U12 on ALL my columns, one tile (block) per wave and round; the inverse of L11 by wave 0 beside the
        //      other waves' loads ----
        {
            const int w = tid >> 6;
            // the idx-th of my blocks that panel k touches: right of it, or left of it when anything was interchanged
            auto nth_block = [&](int idx) -> int {
                for (int cb = me; cb < nblk; cb += 2) {
                    if (cb == k || (cb < k && !anyswap)) continue;
                    if (idx-- == 0) return cb;
                }
                return -1;
            };
            Lu2Tile<T, PB> tl;
            int idx = LU2_NW - 1 - w;
            tl.cb = __builtin_amdgcn_readfirstlane(nth_block(idx));
            tl.right = tl.cb > k;
            LU2_STAMP(8);
            lu2_tile_load<T, PB>(tl, A, ld, N, k0, pb, ne, src, xdst, xsrc);
            LU2_STAMP(9);
            __syncthreads();
            LU2_STAMP(10);
            lu2_tile_finish<T, PB>(tl, A, ld, N, k0, pb, src, LIT, UP, Mpad);
            while (tl.cb >= 0) {
                idx += LU2_NW;
                tl.cb = __builtin_amdgcn_readfirstlane(nth_block(idx));
                tl.right = tl.cb > k;
                lu2_tile_load<T, PB>(tl, A, ld, N, k0, pb, ne, src, xdst, xsrc);
                lu2_tile_finish<T, PB>(tl, A, ld, N, k0, pb, src, LIT, UP, Mpad);
            }
        }
        __syncthreads();
        LU2_STAMP(5);
        if (next_mine) {
            // ---- block k+1: update in registers, factor, publish ----
            const int pb1 = (N - k1 < PB) ? (N - k1) : PB;
            const int M1 = N - k1;
            // panel k's update of block k+1 alone (tile column 0 of the trailing matrix, all eight waves on the matrix cores:
            // in registers -- 2 x 32 x 32 terms per thread on the four panel waves -- it cost 22 k cycles of the chain), then the
            // block's rows into the panel waves' registers
            lu2_trailing<T, PB>(A + (size_t)k1 * ld + k1, ld, M2, LT, UP, Mpad, 0, 0, /*only_tj=*/0);
            __syncthreads();
            vec ra[PB / 4], rb[PB / 4];
            lu2_load_rows<T, PB>(A, ld, ra, rb, k1, pb1, M1);
            LU2_STAMP(2);
            int posa = tid, posb = tid + 256;
            bool donea = !(tid < 256 && tid < M1), doneb = !(tid < 256 && tid + 256 < M1);
            if (M1 > 256) lu2_panel_columns<T, PB, true>(ra, rb, posa, posb, donea, doneb, pb1, k1, S);
            else lu2_panel_columns<T, PB, false>(ra, rb, posa, posb, donea, doneb, pb1, k1, S);
            LU2_STAMP(3);
            lu2_publish<T, PB>(A, ld, ipiv, ra, rb, posa, posb, k + 1, k1, pb1, M1, msg_base, cnt, pidx, pxdst, pxsrc, scr + me, epoch, xlocal);
            LU2_STAMP(4);
        }
        // ---- trailing update of my tile columns: tile column tj = block k + 1 + tj ----
        if (M2 > 0)
            lu2_trailing<T, PB>(A + (size_t)k1 * ld + k1, ld, M2, LT, UP, Mpad, /*tj_par=*/(k + 1 + me) & 1,
                                /*tj_first=*/next_mine ? 1 : 0);
        __syncthreads();
        LU2_STAMP(6);
    }
    // One word, two writers: a time-out (-7) must survive whatever the other workgroup stores afterwards -- atomicMin for it, a
    // compare-exchange from 0 for the index of a zero pivot.  (Workgroup 0 cleared the word when it started; a
    // workgroup 1 that had given up BEFORE that start leaves workgroup 0 without its panels: it times out itself.)
    if (tid == 0 && me == 0) {
        if (cnt[2]) atomicMin(info, -7);           // the partner never arrived (should never happen)
        else if (cnt[1] != 0) atomicCAS(info, 0, cnt[1]);
    }
    if (tid == 0 && me == 1 && cnt[2]) atomicMin(info, -7);
    if (dbg_on && tid == 0 && me == 0) {
        dbt[7] = clock64() - dt_all;
        dbt[11] = xlocal ? 1 : 0;
        for (int q = 0; q < 16; ++q) dbg[q] = dbt[q];
    }
#undef LU2_STAMP
}

template <typename T> __host__ __device__ constexpr int lu2_panel_width() { return sizeof(T) == 4 ? 32 : 16; }

}  // namespace lqp
