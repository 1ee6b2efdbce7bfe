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
constexpr int LU2_SCR_WORDS = 8 + LU2_MSG;   // 64-bit words per problem: [0], [1] panels published by workgroup 0 / 1
                                             // {launch epoch : 32 | count : 32}; [8..): two messages of LU2_MSG ints

template <typename T, int PB> struct Lu2Lds {
    int lt, up, l11, rowp, wval, wrcp, widx, wtid, pidx, src, xdst, xsrc, pxdst, pxsrc, cnt, total;
    __host__ __device__ explicit Lu2Lds(int Mpad) {
        int o = 0;
        lt = o;   o += PB * Mpad * (int)sizeof(T);
        up = o;   o += PB * Mpad * (int)sizeof(T);
        l11 = o;  o += round_up(PB * (PB + 1) * (int)sizeof(T), 32);
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

// f64 trailing update on the matrix cores: v_mfma_f64_16x16x4_f64, D = C - L21 U12 over one 16x16 tile.
//   A operand (16x4 slice of L21): lane l holds L21[i = l & 15][k = l >> 4]
//   B operand (4x16 slice of U12): lane l holds U12[k = l >> 4][j = l & 15]
//   C/D: lane l holds rows 4 (l >> 4) + q, q = 0..3, of column l & 15
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Trailing update of the tile columns this workgroup owns.  A22: the M2 x M2 trailing matrix (row stride ld); LT / UP:
// L21^T and U12 in LDS (row kk at kk * Mpad); tile columns tj with ((tj + tj_par) & 1) == 0 and tj >= tj_first are taken.
// f32: 32x32 tiles (PB == 32 columns = one block per tile column); f64: 16x16 tiles (PB == 16).
template <typename T, int PB>
__device__ __forceinline__ void lu2_trailing(T* __restrict__ A22, const int ld, const int M2, const T* __restrict__ LT,
                                             const T* __restrict__ UP, const int Mpad, const int tj_par, const int tj_first) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if constexpr (sizeof(T) == 4) {
        static_assert(sizeof(T) == 8 || PB == 32, "one 32-column tile per block");
        const int li = lane & 31, lh = lane >> 5;
        const int nt = (M2 + 31) >> 5;
        // my tile columns: tj = tj0, tj0 + 2, ...
        int tj0 = tj_par & 1;                 // smallest tj with (tj + tj_par) even
        while (tj0 < tj_first) tj0 += 2;
        const int ncol = tj0 < nt ? (nt - tj0 + 1) >> 1 : 0;
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
        const int ncol = tj0 < nt ? (nt - tj0 + 1) >> 1 : 0;
        const int ntiles = nt * ncol;
        for (int t = __builtin_amdgcn_readfirstlane(w); t < ntiles; t += LU2_NW) {
            const int ti = t / ncol, tj = tj0 + 2 * (t - ti * ncol);
            const int i0 = ti << 4, j0 = tj << 4;
            double* base = (double*)A22 + (size_t)(i0 + 4 * lh) * ld + j0 + li;
            const bool colok = j0 + li < M2;
            const int rlim = M2 - i0 - 4 * lh;
            f64x4 cur;
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = (colok && q < rlim) ? base[(size_t)q * ld] : 0.0;
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
                if (colok && q < rlim) base[(size_t)q * ld] = cur[q];
        }
    }
}

// the interchanges of a panel applied to column `col` (rows k0 ..), and -- `right` -- U12 = L11^-1 (P A)12 of that column
// into UP[j * Mpad + ucol] and back to the matrix (what wg_lu_factor does per thread; same operations in the same order)
template <typename T, int PB>
__device__ __forceinline__ void lu2_swap_u12_column(T* __restrict__ A, const int ld, const int k0, const int pb, const int col,
                                                    const bool right, const int ucol, const int ne, const int* __restrict__ src,
                                                    const int* __restrict__ xdst, const int* __restrict__ xsrc,
                                                    const T* __restrict__ L11, T* __restrict__ UP, const int Mpad) {
    T* Ac = A + (size_t)k0 * ld + col;
    T top[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j)
        top[j] = (j < pb) ? Ac[(size_t)__builtin_amdgcn_readfirstlane(src[j]) * ld] : T(0);
    for (int q0 = 0; q0 < ne; q0 += 8) {
        T ext[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            ext[q] = (q0 + q < ne) ? Ac[(size_t)__builtin_amdgcn_readfirstlane(xsrc[q0 + q]) * ld] : T(0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q0 + q < ne) Ac[(size_t)__builtin_amdgcn_readfirstlane(xdst[q0 + q]) * ld] = ext[q];
    }
    if (right) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
#pragma unroll
            for (int i = j + 1; i < PB; ++i) top[i] -= L11[i * (PB + 1) + j] * top[j];
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) UP[j * Mpad + ucol] = top[j];
    }
#pragma unroll
    for (int j = 0; j < PB; ++j)
        if (j < pb && (right || __builtin_amdgcn_readfirstlane(src[j]) != j)) Ac[(size_t)j * ld] = top[j];
}

template <typename T> struct Panel2Lds {
    T* rowP; T* wval; T* wrcp; int* widx; int* wtid; int* pidx; int* cnt;
};

// The PB columns of one panel on LU2_PW waves, two rows per thread: row a = relative row t, row b = t + 256.
// Same per-element arithmetic and the same pivot rule as lu_panel_columns (max |.|, ties to the smallest position).
// Every wave of the workgroup calls it (the barrier per column is the workgroup's); waves >= LU2_PW only synchronise.
template <typename T, int PB>
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
                const T aa = ra[j >> 2].v[j & 3], ab = rb[j >> 2].v[j & 3];
                const T ka = donea ? T(-1) : tabs(aa), kb = doneb ? T(-1) : tabs(ab);
                const bool pickb = kb > ka || (kb == ka && posb < posa);
                const T key = pickb ? kb : ka;
                const int cpos = pickb ? posb : posa;
                T myrcp = T(1) / (pickb ? ab : aa);
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
                asm volatile("" ::: "memory");
                if (lane == lb && pickb) {
#pragma unroll
                    for (int q = 0; q < PB / 4; ++q) *(vec*)(cand + w * PB + 4 * q) = rb[q];
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
                // row a
                if (tid == bi) {
                    S.pidx[j] = pivpos;
                    if (!nz && S.cnt[1] == 0) S.cnt[1] = k0 + j + 1;
                    posa = j; donea = true;
                } else if (posa == j) {
                    posa = pivpos;
                }
                if (tid + 256 == bi) {
                    S.pidx[j] = pivpos;
                    if (!nz && S.cnt[1] == 0) S.cnt[1] = k0 + j + 1;
                    posb = j; doneb = true;
                } else if (posb == j) {
                    posb = pivpos;
                }
                {
                    const bool upd = !donea && nz;
                    const T aj = ra[j >> 2].v[j & 3];
                    const T l = upd ? aj * rinv : T(0);
                    ra[j >> 2].v[j & 3] = upd ? l : aj;
#pragma unroll
                    for (int c = j + 1; c < PB; ++c) ra[c >> 2].v[c & 3] -= l * pr[c >> 2].v[c & 3];
                }
                {
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
                                            int* __restrict__ xdst, int* __restrict__ xsrc) {
    T* dst = A + (size_t)(k0 + pos) * ld + k0;
    if (pb == PB) {
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
                                            unsigned long long* __restrict__ flag, const unsigned int epoch) {
    const int tid = threadIdx.x;
    int* msg = msg_base + (kk & 1) * LU2_MSG;
    if (tid < 256) {
        if (tid < M) lu2_put_row<T, PB>(A, ld, ra, posa, tid, k0, pb, msg, cnt, xdst, xsrc);
        if (tid + 256 < M) lu2_put_row<T, PB>(A, ld, rb, posb, tid + 256, k0, pb, msg, cnt, xdst, xsrc);
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
        asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// dbg (optional, workgroup 0 of the matrix): cycles of wave 0 -- [0] waiting for the partner, [1] receive (message, L panel),
// [2] the next panel's columns (interchanges, U12, update in registers), [3] panel factorisation, [4] publish,
// [5] interchanges + U12 of the other columns, [6] trailing update, [7] total
template <typename T, int PB>
__device__ __forceinline__ void wg_lu_factor2(T* __restrict__ A, const int N, const int ld, int* __restrict__ ipiv,
                                              int* __restrict__ info, char* __restrict__ smem, const int me,
                                              unsigned long long* __restrict__ scr, const unsigned int epoch,
                                              unsigned long long* __restrict__ dbg) {
    typedef V4<T> vec;
    const int Mpad = round_up(N, 64);
    const Lu2Lds<T, PB> L(Mpad);
    T* LT = (T*)(smem + L.lt);
    T* UP = (T*)(smem + L.up);
    T* L11 = (T*)(smem + L.l11);
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
    unsigned long long dbt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dt0 = 0, dt_all = 0;
    const bool dbg_on = dbg != nullptr;
    if (dbg_on) dt_all = clock64();
#define LU2_STAMP(i) do { if (dbg_on) { const unsigned long long t_ = clock64(); dbt[i] += t_ - dt0; dt0 = t_; } } while (0)

    if (tid < 4) cnt[tid] = 0;
    if (tid < 2 * LU2_PW) S.wval[tid] = T(-2);
    __syncthreads();

    // ---- prologue: workgroup 0 factors panel 0 ----
    if (me == 0) {
        if (dbg_on) dt0 = clock64();
        const int pb = N < PB ? N : PB;
        vec ra[PB / 4], rb[PB / 4];
        lu2_load_rows<T, PB>(A, ld, ra, rb, 0, pb, N);
        int posa = tid, posb = tid + 256;
        bool donea = !(tid < 256 && tid < N), doneb = !(tid < 256 && tid + 256 < N);
        lu2_panel_columns<T, PB>(ra, rb, posa, posb, donea, doneb, pb, 0, S);
        LU2_STAMP(3);
        lu2_publish<T, PB>(A, ld, ipiv, ra, rb, posa, posb, 0, 0, pb, N, msg_base, cnt, pidx, pxdst, pxsrc, scr + me, epoch);
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
                vec v[PB / 4];
                if (pb == PB) {
#pragma unroll
                    for (int q = 0; q < PB / 4; ++q) v[q] = ld_vec_sc1(rowp + 4 * q);
                } else {
#pragma unroll
                    for (int c = 0; c < PB; ++c) v[c >> 2].v[c & 3] = (c < pb) ? ld_sc1(rowp + c) : T(0);
                }
                if (tid < pb) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) L11[tid * (PB + 1) + c] = v[c >> 2].v[c & 3];
                } else {
#pragma unroll
                    for (int c = 0; c < PB; ++c) LT[c * Mpad + (tid - pb)] = v[c >> 2].v[c & 3];
                }
            }
        }
        __syncthreads();
        const int ne = __builtin_amdgcn_readfirstlane(cnt[3]);
        bool anyswap = ne > 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j < pb && __builtin_amdgcn_readfirstlane(src[j]) != j) anyswap = true;
        LU2_STAMP(1);

        const int k1 = k0 + pb;
        const bool next_mine = (k + 1 < nblk) && (((k + 1) & 1) == me);
        if (next_mine) {
            // ---- the columns of block k+1 first: interchanges + U12, update in registers, factor, publish ----
            const int pb1 = (N - k1 < PB) ? (N - k1) : PB;
            const int M1 = N - k1;
            if (tid < pb1)
                lu2_swap_u12_column<T, PB>(A, ld, k0, pb, k1 + tid, true, tid, ne, src, xdst, xsrc, L11, UP, Mpad);
            __syncthreads();
            vec ra[PB / 4], rb[PB / 4];
            lu2_load_rows<T, PB>(A, ld, ra, rb, k1, pb1, M1);
            if (tid < 256) {
                const bool acta = tid < M1, actb = tid + 256 < M1;
#pragma unroll 4
                for (int kk = 0; kk < PB; ++kk) {
                    const T la = acta ? LT[kk * Mpad + tid] : T(0);
                    const T lb = actb ? LT[kk * Mpad + tid + 256] : T(0);
#pragma unroll
                    for (int q = 0; q < PB / 4; ++q) {
                        const vec uq = *(const vec*)(UP + kk * Mpad + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ra[q].v[e] -= la * uq.v[e];
                            rb[q].v[e] -= lb * uq.v[e];
                        }
                    }
                }
                if (pb1 < PB) {       // (columns beyond the panel: U12 was not computed for them)
#pragma unroll
                    for (int c = 0; c < PB; ++c)
                        if (c >= pb1) { ra[c >> 2].v[c & 3] = T(0); rb[c >> 2].v[c & 3] = T(0); }
                }
            }
            LU2_STAMP(2);
            int posa = tid, posb = tid + 256;
            bool donea = !(tid < 256 && tid < M1), doneb = !(tid < 256 && tid + 256 < M1);
            lu2_panel_columns<T, PB>(ra, rb, posa, posb, donea, doneb, pb1, k1, S);
            LU2_STAMP(3);
            lu2_publish<T, PB>(A, ld, ipiv, ra, rb, posa, posb, k + 1, k1, pb1, M1, msg_base, cnt, pidx, pxdst, pxsrc, scr + me, epoch);
            LU2_STAMP(4);
        }
        // ---- my other columns: interchanges (left and right of the panel) + U12 ----
        {
            const int nmine = ((nblk - me + 1) >> 1) * PB;        // index space over my blocks me, me + 2, ...
            for (int idx = tid; idx < nmine; idx += LU2_NT) {
                const int cb = 2 * (idx / PB) + me;
                const int col = cb * PB + (idx % PB);
                if (cb == k || (next_mine && cb == k + 1) || col >= N) continue;
                const bool right = cb > k;
                if (right || anyswap)
                    lu2_swap_u12_column<T, PB>(A, ld, k0, pb, col, right, col - k1, ne, src, xdst, xsrc, L11, UP, Mpad);
            }
        }
        __syncthreads();
        LU2_STAMP(5);
        // ---- trailing update of my tile columns: tile column tj = block k + 1 + tj ----
        if (M2 > 0)
            lu2_trailing<T, PB>(A + (size_t)k1 * ld + k1, ld, M2, LT, UP, Mpad, /*tj_par=*/(k + 1 + me) & 1,
                                /*tj_first=*/next_mine ? 1 : 0);
        __syncthreads();
        LU2_STAMP(6);
    }
    if (tid == 0 && me == 0) {
        if (cnt[2]) *info = -7;                    // the partner never arrived (should never happen)
        else if (cnt[1] != 0) *info = cnt[1];
    }
    if (tid == 0 && me == 1 && cnt[2]) *info = -7;
    if (dbg_on && tid == 0 && me == 0) {
        dbt[7] = clock64() - dt_all;
        for (int q = 0; q < 8; ++q) dbg[q] = dbt[q];
    }
#undef LU2_STAMP
}

template <typename T> __host__ __device__ constexpr int lu2_panel_width() { return sizeof(T) == 4 ? 32 : 16; }

}  // namespace lqp
