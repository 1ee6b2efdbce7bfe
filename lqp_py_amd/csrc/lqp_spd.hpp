// Symmetric "inverse" path of the x-update (f32, n <= 512, few equality rows).
//
// The reference solves M [x; nu] = [w; b], M = [[K, A^T], [A, 0]], K = Qs + rho I, with a cached pivoted
// LU every iteration (lqp_py/solve_box_qp_admm_torch.py:205-215, 258-268).  K is symmetric positive
// definite whenever the QP is convex and rho > 0, and only x is needed inside the loop, so
//
//     x = H w + c,    H = K^-1 - G S^-1 G^T,   G = K^-1 A^T,   S = A G,   c = G S^-1 b,
//     nu = S^-1 (G^T w - b)                                  (needed once, after the last x-update)
//
// H is symmetric: its lower 64x64 blocks are stored once and every block serves BOTH products
// (y_i += B x_j and y_j += B^T x_i).  Per iteration the loop then moves n^2/2 elements instead of the
// N^2 of the two triangular solves, has no dependent chain (a GEMV), and its head stays on chip.
//
//   wg_spd_sweep      -K^-1 in place on the block-packed lower triangle: block symmetric sweep, each pivot
//                     block through its Cholesky factor (W = L^-1), all rank-64 updates on MFMA
//   wg_sym_gemv       y = Hs w from the packed lower blocks (register / LDS resident head + prefetch ring)
//   k_eq_correct      G, S^-1, T = G S^-1, c, s0 = S^-1 b and the rank-m correction Hs += T G^T
//
// Stored matrix Hs = -H (the sweep ends at -K^-1; the sign is folded into the loop: x = c - Hs w).
// A non-positive pivot (K not positive definite in f32) is reported through info; the host then repeats
// the solve on the LU path.
#pragma once
#include "lqp_common.hpp"
#include "lqp_lu.hpp"
#include "lqp_trsv.hpp"
#include "lqp_f16x2.hpp"

namespace lqp {

constexpr int SPD_LS = 68;          // LDS row stride (floats) of the 64-row operand panels: 16-B aligned, conflict-free b128
constexpr int SPD_MAXK = 8;         // n <= 512: the (K-1)-block panel + W + W^T fill the 160 KB of LDS
constexpr int SPD_MAXM = 16;        // equality rows handled by the rank-m correction

__host__ __device__ constexpr int sym_blocks(int K) { return K * (K + 1) / 2; }
// stream index of lower block (i, j), i >= j: column-major over the lower triangle
__host__ __device__ constexpr int sym_idx(int i, int j, int K) { return j * K - j * (j - 1) / 2 + (i - j); }
#ifndef LQP_PIV_WAVES
#define LQP_PIV_WAVES 16        // waves that eliminate a pivot block in a 1024-thread workgroup (4, 8 or 16)
#endif
#ifndef LQP_PIV_WAVES_RS
#define LQP_PIV_WAVES_RS 8      // ... in the 512-thread resident sweep (4 or 8)
#endif
#ifndef LQP_PIV_NB
#define LQP_PIV_NB 4
#endif
constexpr int PIV_NB = LQP_PIV_NB;                            // pivot columns per LDS exchange of the pivot-block elimination
constexpr int PIV_LDS = 2 * PIV_NB * 64 + 64;        // floats: coefficients [2][PIV_NB][64] | scales [64]
__host__ __device__ inline int spd_lds_bytes(int K) {
    return ((K > 1 ? K - 1 : 1) + 2) * 64 * SPD_LS * 4 + PIV_LDS * 4 + 16;
}

// ---- packed lower blocks <- K = src (+ rho on the diagonal), identity on the padding ----
// dsc != nullptr: src is the UNSCALED matrix and the value taken is (dsc[row] * src[row][col]) * dsc[col] -- what the
// scaling pass would have stored (same operations, same order): the scaled copy is then never written.
__device__ __forceinline__ void sym_scale4(V4<float>& v, const float* __restrict__ dsc, const int row, const int col, const int n) {
    if (dsc && row < n) {
        const float di = dsc[row];
        if (col + 3 < n && ((((uintptr_t)(dsc + col)) & 15) == 0)) {
            const V4<float> dj = *(const V4<float>*)(dsc + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = (di * v.v[e]) * dj.v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (col + e < n) v.v[e] = (di * v.v[e]) * dsc[col + e];
        }
    }
}
__device__ __forceinline__ void wg_sym_init(float* __restrict__ Hs, const float* __restrict__ src, const int ld,
                                            const int n, const int K, const float rho_add,
                                            const float* __restrict__ dsc = nullptr) {
    const int tid = threadIdx.x, r = tid >> 4, c4 = (tid & 15) * 4;
    const bool vec_ok = (ld % 4 == 0) && ((((uintptr_t)src) & 15) == 0);
    for (int j = 0; j < K; ++j)
        for (int i = j; i < K; ++i) {
            const int gr = i * 64 + r, gc = j * 64 + c4;
            V4<float> v;
            if (gr < n && gc + 3 < n && vec_ok) {
                v = *(const V4<float>*)(src + (size_t)gr * ld + gc);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v.v[e] = (gr < n && gc + e < n) ? src[(size_t)gr * ld + gc + e] : (gr == gc + e ? 1.f : 0.f);
            }
            sym_scale4(v, dsc, gr, gc, n);
            if (gr < n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gr == gc + e) v.v[e] += rho_add;
            }
            *(V4<float>*)(Hs + (size_t)sym_idx(i, j, K) * LQP_BLK + tid * 4) = v;
        }
}

// max |A_ij - A_ji| over the matrix, compared with its rounding level: returns > 0 when src is NOT symmetric
// (difference above 1e-5 of the largest entry), 0 otherwise.  Every block (i, j) is read next to its mirror
// (j, i), both as coalesced row segments; the mirror goes through a padded LDS tile (two tiles alternate, one
// barrier per block) and is compared transposed.  (Comparing src[c][r] straight from global memory cost 95 us per
// launch at n = 500: 16-B pieces of 64 different rows per load instruction.)
// smem: 2 * 64 * SPD_LS + 2 * LQP_NW floats.
__device__ __forceinline__ float wg_sym_asymmetry(const float* __restrict__ src, const int ld, const int n, const int K,
                                                  float* __restrict__ smem_f, const float* __restrict__ dsc = nullptr) {
    const int tid = threadIdx.x, r = tid >> 4, c4 = (tid & 15) * 4;
    float* tile = smem_f;                                   // [2][64][SPD_LS]
    float* red = smem_f + 2 * 64 * SPD_LS;
    const bool vec_ok = (ld % 4 == 0) && ((((uintptr_t)src) & 15) == 0);
    // 4 elements of row `row`, columns col .. col+3 (zero outside the matrix)
    auto load4 = [&](const int row, const int col) -> V4<float> {
        V4<float> v;
        if (row < n && col + 3 < n && vec_ok) {
            v = *(const V4<float>*)(src + (size_t)row * ld + col);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = (row < n && col + e < n) ? src[(size_t)row * ld + col + e] : 0.f;
        }
        sym_scale4(v, dsc, row, col, n);
        return v;
    };
    float dmax = 0.f, vmax = 0.f;
    const int nblk = sym_blocks(K);
    int i = 0, j = 0;
    V4<float> a = load4(r, c4), bm = a;                      // block (0,0) is its own mirror
    for (int t = 0; t < nblk; ++t) {
        float* T = tile + (t & 1) * 64 * SPD_LS;
        *(V4<float>*)(T + r * SPD_LS + c4) = bm;
        const V4<float> ac = a;
        // the next block's two row segments are requested before this one is compared
        int ni = i + 1, nj = j;
        if (ni == K) { ++nj; ni = nj; }
        if (t + 1 < nblk) {
            a = load4(ni * 64 + r, nj * 64 + c4);
            bm = load4(nj * 64 + r, ni * 64 + c4);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bt = T[(c4 + e) * SPD_LS + r];      // mirror[c][r] = src[64 j + c][64 i + r]
            dmax = tmax(dmax, tabs(ac.v[e] - bt));
            vmax = tmax(vmax, tmax(tabs(ac.v[e]), tabs(bt)));
        }
        i = ni; j = nj;
    }
    __syncthreads();
    dmax = wg_max(dmax, red);
    vmax = wg_max(vmax, red + LQP_NW);
    return dmax > 1e-5f * vmax ? dmax : 0.f;
}

// wg_sym_init and (when `check`) wg_sym_asymmetry in ONE pass over src, for the blocks part, part + NP, ... of the
// stream: workgroup `part` of NP.  The symmetry verdict then is per workgroup (difference above 1e-5 of the largest
// entry among ITS blocks and their mirrors; consecutive blocks alternate between the workgroups, so each of them
// sees diagonal blocks).  smem as wg_sym_asymmetry.
template <int NP>
__device__ __forceinline__ float wg_sym_check_init(float* __restrict__ Hs, const float* __restrict__ src, const int ld,
                                                   const int n, const int K, const float rho_add,
                                                   float* __restrict__ smem_f, const bool check, const int part,
                                                   const float* __restrict__ dsc = nullptr,
                                                   float* __restrict__ fro_out = nullptr) {
    const int tid = threadIdx.x, r = tid >> 4, c4 = (tid & 15) * 4;
    float* tile = smem_f;                                   // [2][64][SPD_LS]
    float* red = smem_f + 2 * 64 * SPD_LS;
    const bool vec_ok = (ld % 4 == 0) && ((((uintptr_t)src) & 15) == 0);
    auto load4 = [&](const int row, const int col) -> V4<float> {      // zero outside the matrix
        V4<float> v;
        if (row < n && col + 3 < n && vec_ok) {
            v = *(const V4<float>*)(src + (size_t)row * ld + col);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = (row < n && col + e < n) ? src[(size_t)row * ld + col + e] : 0.f;
        }
        sym_scale4(v, dsc, row, col, n);
        return v;
    };
    auto block_of = [&](const int t, int& i, int& j) {
        int rem = t;
        j = 0;
        while (rem >= K - j) { rem -= K - j; ++j; }
        i = j + rem;
    };
    float dmax = 0.f, vmax = 0.f, fro2 = 0.f;
    const int nblk = sym_blocks(K);
    int i = 0, j = 0;
    V4<float> a, bm;
    if (part < nblk) {
        block_of(part, i, j);
        a = load4(i * 64 + r, j * 64 + c4);
        if (check) bm = load4(j * 64 + r, i * 64 + c4);
    }
    int cnt = 0;
    for (int t = part; t < nblk; t += NP, ++cnt) {
        float* T = tile + (cnt & 1) * 64 * SPD_LS;
        if (check) *(V4<float>*)(T + r * SPD_LS + c4) = bm;
        const V4<float> ac = a;
        const int ci = i, cj = j;
        if (fro_out) {                                      // (needs `check`: the mirrors are only loaded then)
#pragma unroll
            for (int e = 0; e < 4; ++e) fro2 += ac.v[e] * ac.v[e] + (ci != cj ? bm.v[e] * bm.v[e] : 0.f);
        }
        if (t + NP < nblk) {                                // the next block is requested before this one is used
            block_of(t + NP, i, j);
            a = load4(i * 64 + r, j * 64 + c4);
            if (check) bm = load4(j * 64 + r, i * 64 + c4);
        }
        V4<float> v = ac;
        const int gr = ci * 64 + r, gc = cj * 64 + c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (gr == gc + e) v.v[e] += gr < n ? rho_add : 1.f;     // identity on the padding
        *(V4<float>*)(Hs + (size_t)sym_idx(ci, cj, K) * LQP_BLK + tid * 4) = v;
        if (check) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bt = T[(c4 + e) * SPD_LS + r];
                dmax = tmax(dmax, tabs(ac.v[e] - bt));
                vmax = tmax(vmax, tmax(tabs(ac.v[e]), tabs(bt)));
            }
        }
    }
    if (!check) return 0.f;
    __syncthreads();
    if (fro_out) {
        fro2 = wg_sum(fro2, red);
        if (tid == 0) *fro_out = fro2;
    }
    dmax = wg_max(dmax, red);
    vmax = wg_max(vmax, red + LQP_NW);
    return dmax > 1e-5f * vmax ? dmax : 0.f;
}

// ONE pass over an UNSCALED src for everything that needs all of it before the scaling is known: the lower blocks as they
// are (identity on the padding, nothing on the diagonal), the symmetry verdict of wg_sym_check_init, and the column maxima
// of |src| (what the auto-scaling starts from, reference :163) -- of this workgroup's blocks AND their mirrors, so that the
// NP partial vectors together cover every entry.  The readers of the blocks scale them as they load them
// (wg_spd_sweep_resident_v2, RsLateRho::dsc).  cm_out: 64 K floats (global).
// smem: 2 * 64 * SPD_LS + 2 * LQP_NW + 64 K floats.
template <int NP>
__device__ __forceinline__ float wg_sym_prep(float* __restrict__ Hs, const float* __restrict__ src, const int ld,
                                             const int n, const int K, float* __restrict__ smem_f, const int part,
                                             float* __restrict__ cm_out) {
    const int tid = threadIdx.x, lane = tid & 63, r = tid >> 4, c4 = (tid & 15) * 4;
    float* tile = smem_f;                                   // [2][64][SPD_LS]
    float* red = smem_f + 2 * 64 * SPD_LS;
    unsigned int* cm = (unsigned int*)(red + 2 * LQP_NW);   // column maxima as bit patterns (non-negative floats order like integers)
    for (int i = tid; i < 64 * K; i += (int)blockDim.x) cm[i] = 0u;
    const bool vec_ok = (ld % 4 == 0) && ((((uintptr_t)src) & 15) == 0);
    auto load4 = [&](const int row, const int col) -> V4<float> {      // zero outside the matrix
        V4<float> v;
        if (row < n && col + 3 < n && vec_ok) {
            v = *(const V4<float>*)(src + (size_t)row * ld + col);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = (row < n && col + e < n) ? src[(size_t)row * ld + col + e] : 0.f;
        }
        return v;
    };
    auto block_of = [&](const int t, int& i, int& j) {
        int rem = t;
        j = 0;
        while (rem >= K - j) { rem -= K - j; ++j; }
        i = j + rem;
    };
    // max over the four rows a wave holds of one column quad, then one LDS atomic per column from lanes 0..15
    auto col_max = [&](const V4<float>& v, const int col0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = tabs(v.v[e]);
            a = tmax(a, xor16(a));
            a = tmax(a, xor32(a));
            if (lane < 16) atomicMax(cm + col0 + c4 + e, __float_as_uint(a));
        }
    };
    float dmax = 0.f, vmax = 0.f;
    const int nblk = sym_blocks(K);
    // two blocks (and their mirrors) are requested ahead of the one in use: with one, the pass ran at 3.8 TB/s
    int i = 0, j = 0, i1 = 0, j1 = 0;
    V4<float> a, bm, a1, bm1;
    if (part < nblk) {
        block_of(part, i, j);
        a = load4(i * 64 + r, j * 64 + c4);
        bm = load4(j * 64 + r, i * 64 + c4);
    }
    if (part + NP < nblk) {
        block_of(part + NP, i1, j1);
        a1 = load4(i1 * 64 + r, j1 * 64 + c4);
        bm1 = load4(j1 * 64 + r, i1 * 64 + c4);
    }
    __syncthreads();                                        // (cm is zero)
    int cnt = 0;
    for (int t = part; t < nblk; t += NP, ++cnt) {
        // the mirror's LDS tile: row stride 64 with the 16-byte chunks of row R at positions chunk ^ (R >> 2) -- the 16-byte
        // row stores stay conflict-free and the transposed reads below hit 32 different banks per wave (with padded rows a
        // stride that keeps 16-byte alignment puts the 16 rows a wave reads on two banks: 8-way conflicts)
        float* T = tile + (cnt & 1) * 64 * SPD_LS;
        *(V4<float>*)(T + r * 64 + 4 * ((c4 >> 2) ^ ((r >> 2) & 15))) = bm;
        const V4<float> ac = a, bc = bm;
        const int ci = i, cj = j;
        a = a1; bm = bm1; i = i1; j = j1;
        if (t + 2 * NP < nblk) {
            block_of(t + 2 * NP, i1, j1);
            a1 = load4(i1 * 64 + r, j1 * 64 + c4);
            bm1 = load4(j1 * 64 + r, i1 * 64 + c4);
        }
        V4<float> v = ac;
        const int gr = ci * 64 + r, gc = cj * 64 + c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (gr == gc + e && gr >= n) v.v[e] = 1.f;      // identity on the padding
        *(V4<float>*)(Hs + (size_t)sym_idx(ci, cj, K) * LQP_BLK + tid * 4) = v;
        col_max(ac, cj * 64);
        if (ci != cj) col_max(bc, ci * 64);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bt = T[(c4 + e) * 64 + ((r & 3) | (4 * ((r >> 2) ^ (((c4 + e) >> 2) & 15))))];
            dmax = tmax(dmax, tabs(ac.v[e] - bt));
            vmax = tmax(vmax, tmax(tabs(ac.v[e]), tabs(bt)));
        }
    }
    __syncthreads();
    for (int c = tid; c < 64 * K; c += (int)blockDim.x) cm_out[c] = __uint_as_float(cm[c]);
    dmax = wg_max(dmax, red);
    vmax = wg_max(vmax, red + LQP_NW);
    return dmax > 1e-5f * vmax ? dmax : 0.f;
}

// one 32x32 output quadrant: acc = X[x0 .. x0+31][0..63] * Z[z0 .. z0+31][0..63]^T, both operands in LDS with
// row stride SPD_LS.  Lane l feeds row l&31 and the k range 32*(l>>5) .. +31 (any pairing of k values is a
// valid MFMA schedule as long as A and B agree).
// HALF: only k in [32, 64) (UP) or [0, 32) contributes (a triangular operand is zero on the other half): 16 MFMAs.
template <int HALF = 0, bool UP = false>
__device__ __forceinline__ f32x16 spd_quadrant(const float* __restrict__ X, const float* __restrict__ Z) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    constexpr int KL = HALF ? 16 : 32;                   // k values per lane half
    const int kb = (HALF && UP ? 32 : 0) + KL * lh;
    const float* xa = X + li * SPD_LS + kb;
    const float* zb = Z + li * SPD_LS + kb;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int t = 0; t < KL / 4; ++t) {
        const V4<float> a = *(const V4<float>*)(xa + 4 * t);
        const V4<float> b = *(const V4<float>*)(zb + 4 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[e], b.v[e], acc, 0, 0, 0);
    }
    return acc;
}
// the same product with the lane parts of the operand addresses made by the caller: xa = X + (l&31) * SPD_LS + kb,
// zb = Z + (l&31) * SPD_LS + kb, kb = 32 (l>>5) for the full k range, 16 (l>>5) [+ 32: upper half] with HALF
template <int HALF = 0>
__device__ __forceinline__ f32x16 spd_quadrant_lp(const float* __restrict__ xa, const float* __restrict__ zb) {
    constexpr int KL = HALF ? 16 : 32;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int t = 0; t < KL / 4; ++t) {
        const V4<float> a = *(const V4<float>*)(xa + 4 * t);
        const V4<float> b = *(const V4<float>*)(zb + 4 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[e], b.v[e], acc, 0, 0, 0);
    }
    return acc;
}
// accumulator register q of lane l is element (row = (q&3) + 8 (q>>2) + 4 (l>>5), col = l&31) of the quadrant
__device__ __forceinline__ int quad_row(int q, int lh) { return (q & 3) + 8 * (q >> 2) + 4 * lh; }

// ---- pivot block: in-place forward elimination of [A | I] -> W = L^-1 (A = L L^T), W and W^T left in LDS ----
// src: the 64x64 block in global memory (row-major); kbase: 64 * (index of the pivot block), for the error code.
// Contains workgroup barriers: every thread of the workgroup must call it.
//
// Column c switches role from "A" to "augmented" at step c, so one 64x64 array is enough.  NWP waves work (lane = row,
// wave = a group of CW = 64 / NWP columns): the pivot row reaches a wave through v_readlane from its own lane c, the
// elimination coefficients through LDS.  Columns are eliminated in PANELS of PIV_NB: the wave that holds the panel's
// columns (all 64 rows of them, in registers) runs its PIV_NB column steps alone -- pivot from its own lane c by
// v_readlane, no LDS -- and publishes the PIV_NB coefficient columns once; after ONE barrier every wave applies the
// PIV_NB rank-1 updates to its own columns.  Every element sees the same operations in the same order whatever NWP
// and PIV_NB are: identical bits.  Measured per block (1024-thread workgroup, one per CU): column-by-column with a
// barrier per column and 4 waves 48k cycles (16 waves: 57k); panels of 4 with 4 waves 36k; the per-pivot work of a
// wave is ~2.6 instructions per column it holds, and one wave alone on its SIMD issues a VALU instruction every 4
// cycles where two or more issue one every 2: more waves with fewer columns each run the same update sooner.
// Synchronisation of a GROUP of waves of one workgroup through an LDS counter (the other waves of the workgroup do
// something else meanwhile, so s_barrier cannot be used): every wave of the group adds 1 and waits until the counter
// has reached `target` (= group size x number of syncs so far).  All waves of a workgroup are resident, so the wait
// always ends.
template <bool TIGHT = false>
__device__ __forceinline__ void lds_group_sync(int* cnt, const int target) {
    if constexpr (TIGHT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS traffic only (the pivot chain)
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        if constexpr (!TIGHT) __builtin_amdgcn_s_sleep(1);
    if constexpr (TIGHT) asm volatile("" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#ifndef LQP_WAIT_SLEEP
#define LQP_WAIT_SLEEP 1       // s_sleep argument of the waves that wait for another group of their workgroup (x 64 cycles)
#endif
__device__ __forceinline__ void lds_wait_ge(int* word, const int target) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
        __builtin_amdgcn_s_sleep(LQP_WAIT_SLEEP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the same wait with a way out: after ~0.5 s the timeout word is raised and the wait ends (the results are flagged, the
// workgroup still reaches its barriers) -- for schedules in which one group of waves waits for ANOTHER group's progress
__device__ __forceinline__ void lds_wait_ge_bounded(int* word, const int target, int* __restrict__ timeout_word) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();             // 100 MHz
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target) {
        __builtin_amdgcn_s_sleep(LQP_WAIT_SLEEP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ULL) {
            __hip_atomic_store(timeout_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// GSYNC = false: called by all waves of the workgroup (waves >= NWP only take part in the barriers).
// GSYNC = true : called ONLY by waves 0..NWP-1 while the other waves of the workgroup do something else (s_barrier is
//                not available then): they synchronise among themselves through the LDS counter `gaux`, `gseq` is the
//                group's running target.  The caller guarantees that nobody still reads W / W^T and that the source
//                block is complete.  (Measured alternative, slower: coefficients carrying a tag that the readers
//                poll, with acknowledgements for the double buffer -- 230 k cycles for the 5.3 blocks of a backward
//                factorisation against 195 k with the counter.  The same idea for all 16 waves without any barrier,
//                one buffer per panel, so that the other waves' update phases run behind the owner chain instead of in
//                it: 14.4 - 15.4 us per block against 11.1 us -- polling waves cost more than the barrier they replace.)
#ifndef LQP_PIV_PRIO
#define LQP_PIV_PRIO 1
#endif
#ifndef LQP_PIV_EXP
#define LQP_PIV_EXP 0      // timing experiments only: 1 = no panel sync, 2 = no update phase, 4 = no column steps
#endif
typedef float piv_f2 __attribute__((ext_vector_type(2)));
template <int NWP, bool GSYNC = false>
__device__ __forceinline__ void wg_pivot_block(const float* __restrict__ src_blk, float* __restrict__ W,
                                               float* __restrict__ WT, float* __restrict__ pcol,
                                               int* __restrict__ flag, const int kbase,
                                               int* __restrict__ gaux = nullptr, int* __restrict__ gseq = nullptr) {
    static_assert(NWP == 4 || NWP == 8 || NWP == 16, "4, 8 or 16 pivot waves");
    constexpr int CW = 64 / NWP;                 // columns per wave
    static_assert(CW % PIV_NB == 0 && PIV_NB % 2 == 0, "a panel lives in one wave; columns are updated in pairs");
    const int tid = threadIdx.x, w = tid >> 6;
    const int prw = tid & 63, pq = w;
    const bool pwork = pq < NWP;
    float* const coefs = pcol;                   // [2][PIV_NB][64]: elimination coefficients of the current panel
    float* const svals = pcol + 2 * PIV_NB * 64; // [64]: 1 / sqrt(pivot)
    float xq[CW];
    if (pwork) {
        const float* src = src_blk + prw * 64 + pq * CW;
#pragma unroll
        for (int t = 0; t < CW / 4; ++t) {
            const V4<float> v4 = *(const V4<float>*)(src + 4 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[4 * t + e] = v4.v[e];
        }
    }
    if constexpr (!GSYNC) __syncthreads();     // previous step's LDS reads are over
    if (pwork) {
#pragma unroll 1
        for (int qc = 0; qc < NWP; ++qc) {
#pragma unroll
            for (int pp = 0; pp < CW / PIV_NB; ++pp) {
                const int c0 = qc * CW + pp * PIV_NB;
                // (double buffer by panel parity: a wave is at most one barrier ahead of the slowest reader)
                float* cf = coefs + (((qc * (CW / PIV_NB)) + pp) & 1) * PIV_NB * 64;
                float cv[PIV_NB];
                if (pq == qc && !(LQP_PIV_EXP & 4)) {
                    // the panel's column steps: one wave, everything in registers; this chain is the critical path of the
                    // whole block, so nothing but the arithmetic sits in it (error flag and scales leave after the panel)
                    float sreg[PIV_NB];
                    int badc = 0;
#pragma unroll
                    for (int t = 0; t < PIV_NB; ++t) {
                        const int ec = pp * PIV_NB + t, c = c0 + t;
                        const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xq[ec]), c));
                        // (a non-positive pivot is only RECORDED here: the matrix is then not positive definite, the
                        //  caller falls back to LU or poisons its results; keeping the test out of the arithmetic saves
                        //  two VALU <-> SGPR round trips per column on this dependent chain)
                        badc = (!(d > 0.f) && badc == 0) ? c + 1 : badc;
                        const float s = __builtin_amdgcn_rsqf(d);
                        const bool below = prw > c, on = prw == c;
                        // rows below the pivot: x -= (x_rc / d) * (pivot row); the pivot row keeps its raw values and
                        // is scaled by s when W is written (it is final: nobody reads it again)
                        const float coef = below ? xq[ec] * s * s : 0.f;
                        float pr[PIV_NB];
#pragma unroll
                        for (int t2 = 0; t2 < PIV_NB; ++t2)
                            pr[t2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xq[pp * PIV_NB + t2]), c));
                        __builtin_amdgcn_sched_barrier(0);
                        {
                            const piv_f2 nc = piv_f2{-coef, -coef};
#pragma unroll
                            for (int t2 = 0; t2 < PIV_NB; t2 += 2) {
                                piv_f2 x = piv_f2{xq[pp * PIV_NB + t2], xq[pp * PIV_NB + t2 + 1]};
                                x = __builtin_elementwise_fma(nc, piv_f2{pr[t2], pr[t2 + 1]}, x);
                                xq[pp * PIV_NB + t2] = x[0]; xq[pp * PIV_NB + t2 + 1] = x[1];
                            }
                        }
                        xq[ec] = below ? -coef : (on ? 1.f : xq[ec]);
                        cf[t * 64 + prw] = coef;
                        sreg[t] = s;
                    }
                    float sv = sreg[0];
#pragma unroll
                    for (int t = 1; t < PIV_NB; ++t) sv = prw == t ? sreg[t] : sv;
                    if (prw < PIV_NB) svals[c0 + prw] = sv;
                    if (badc != 0 && prw == 0 && flag[0] == 0) flag[0] = kbase + badc;
                    if constexpr (!GSYNC && LQP_PIV_PRIO) __builtin_amdgcn_s_setprio(0);
                }
                if constexpr (LQP_PIV_EXP & 1) {}
                else if constexpr (GSYNC) lds_group_sync<true>(gaux, *gseq += NWP);
                else wg_barrier_lds();
                // the wave that owns the NEXT panel goes first on its SIMD: its update + column steps are the critical
                // path, the other waves' updates fill the gaps
                if constexpr (!GSYNC && LQP_PIV_PRIO)
                    if (pq == ((pp + 1 < CW / PIV_NB) ? qc : qc + 1)) __builtin_amdgcn_s_setprio(3);
                if constexpr (!(LQP_PIV_EXP & 2)) {
                    // all coefficients of the panel in one LDS round trip (one at a time, each read's latency sits in
                    // the dependent chain of the columns)
#pragma unroll
                    for (int t = 0; t < PIV_NB; ++t) cv[t] = cf[t * 64 + prw];
                    // the pivot row comes from this wave's own lane c: all v_readlane of a column first, into
                    // different SGPRs (interleaved with their FMAs the compiler funnels them through ONE SGPR, and
                    // the readlane -> fma -> readlane chain then costs ~35 cycles per element).  The owner's own panel
                    // columns are final: it skips them (wave-uniform branch, no selects in the chain).
                    if (pq == qc) {
                        if constexpr (CW > PIV_NB) {
#pragma unroll
                            for (int t = 0; t < PIV_NB; ++t) {
                                const int c = c0 + t;
                                float pr[CW];
#pragma unroll
                                for (int e = 0; e < CW; ++e)
                                    if (e / PIV_NB != pp)
                                        pr[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xq[e]), c));
                                __builtin_amdgcn_sched_barrier(0);
                                {
                                    const piv_f2 nc = piv_f2{-cv[t], -cv[t]};
#pragma unroll
                                    for (int e = 0; e < CW; e += 2) {
                                        if (e / PIV_NB != pp) {
                                            piv_f2 x = piv_f2{xq[e], xq[e + 1]};
                                            x = __builtin_elementwise_fma(nc, piv_f2{pr[e], pr[e + 1]}, x);
                                            xq[e] = x[0]; xq[e + 1] = x[1];
                                        }
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < PIV_NB; ++t) {
                            const int c = c0 + t;
                            float pr[CW];
#pragma unroll
                            for (int e = 0; e < CW; ++e)
                                pr[e] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xq[e]), c));
                            __builtin_amdgcn_sched_barrier(0);
                            // two columns per v_pk_fma_f32 (the pivot-row pair straight from the SGPR pair the two
                            // v_readlane wrote; same rounding as the scalar FMA): the update phase is issue bound
                            {
                                const piv_f2 nc = piv_f2{-cv[t], -cv[t]};
#pragma unroll
                                for (int e = 0; e < CW; e += 2) {
                                    piv_f2 x = piv_f2{xq[e], xq[e + 1]};
                                    x = __builtin_elementwise_fma(nc, piv_f2{pr[e], pr[e + 1]}, x);
                                    xq[e] = x[0]; xq[e + 1] = x[1];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
        const float srow = svals[prw];             // scale of this lane's row (written before the last barrier)
        // W (lower, zero above the diagonal) and W^T
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            const int q = pq * CW + e;
            xq[e] = (q > prw) ? 0.f : xq[e] * srow;
            WT[q * SPD_LS + prw] = xq[e];
        }
#pragma unroll
        for (int t = 0; t < CW / 4; ++t) {
            V4<float> v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4.v[e] = xq[4 * t + e];
            *(V4<float>*)(W + prw * SPD_LS + pq * CW + 4 * t) = v4;
        }
    } else if constexpr (!GSYNC) {
#pragma unroll 1
        for (int c = 0; c < 64 / PIV_NB; ++c) wg_barrier_lds();
    }
}

// ---- the same pivot block on the matrix cores ---------------------------------------------------------------
// wg_pivot_block spends half of its time in 4096 (v_readlane, FMA) pairs -- the rank-1 updates of the 64 columns,
// pivot-row entries broadcast one by one -- and a sixth in LDS write -> barrier -> read round trips, 16 waves busy
// with work that is a rank-4 update per panel.  Here a panel's update IS one matrix instruction pair per 32x32
// quadrant (v_mfma_f32_32x32x2_f32: exact f32, an fma chain over k), the working array lives in MFMA accumulators,
// and four waves work, each on the quadrants it owns:
//   wave 0  chain, panels 0..7.  S = the Schur complement, quadrants (0,0), (0,1).  Per panel of 4 columns: rows
//           c0..c0+3 of S come out of the accumulators as lane vectors (register q of lane half h holds one row of a
//           quadrant across 32 lanes: ONE v_permlane32_swap joins the two quadrants of a row); by symmetry they are the
//           panel's columns; the four column steps run on them as before (pivot by v_readlane, rsq, coefficient column,
//           the later panel columns updated); the coefficient columns (lane = row) and the transformed pivot rows
//           (lane = column) are exactly the A and B operands of the rank-4 update once their halves are paired
//           (v_permlane32_swap again): no LDS, no barrier on the chain.  Both also go to LDS queues (the W^T area and
//           the upper half of the W area, free until the end).
//   wave 3  quadrant (1,1) of S: follows the queues through panels 0..7, then IS the chain for panels 8..15.
//   wave 1, wave 2  the augmented part (W under construction, unit lower triangular until the final scaling): column
//           block 0 = quadrants (0,0), (1,0) | column block 1 = quadrant (1,1).  They follow the coefficient queue: the
//           coefficient columns are the A operand, the pivot rows of the panel come from their OWN accumulators
//           (transformed by the panel's 4x4 unit triangle first: 6 FMAs), so nothing but the queue is shared.
// The other waves of the workgroup leave after the staging barrier.  W, W^T as wg_pivot_block leaves them (W = L^-1
// row-scaled by 1/sqrt(pivot), zero above the diagonal).  Rounding differs from wg_pivot_block in the last bits (the
// pivot rows are taken as the symmetric counterpart of the columns); every caller uses one of the two, never both.
// Measured: see DESIGN.md section 5 (tools/microbench/piv_bench.hip).
#ifndef LQP_PIV_MFMA
#define LQP_PIV_MFMA 1
#endif
#ifdef LQP_PIV_STAMPS          // timing builds of tools/microbench only: cycle stamps of the working waves of workgroup 0
__device__ unsigned long long* g_piv_stamps = nullptr;
// (each stamp reloads the pointer from memory: ~400 cycles -- a run with stamps is ~25 % slower, read differences with that in mind)
#define PIV_STAMP(slot) do { if (g_piv_stamps && blockIdx.x == 0 && lane == 0) g_piv_stamps[(slot)] = clock64(); } while (0)
#else
#define PIV_STAMP(slot) do { } while (0)
#endif
__device__ __forceinline__ float piv_readlane(const float v, const int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// lo = [a(lanes 0-31) | b(lanes 0-31)], hi = [a(lanes 32-63) | b(lanes 32-63)]
// (v_permlane32_swap through inline assembly: with __builtin_amdgcn_permlane32_swap hipcc 7.2 handed the FIRST result to
//  users of the second one as soon as both were live across other code -- both MFMAs of a panel then got the same B
//  operand.  The s_nops cover the wait states the swap needs behind a VALU write of its operands and before its results are read; the assembler
//  block is invisible to the compiler's hazard recogniser.)
__device__ __forceinline__ void piv_pair(float a, float b, float& lo, float& hi) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
}
// the same IN PLACE (a := lo, b := hi) for two / four independent pairs behind ONE pair of wait-state nops: the chain
// wave is bound by its instruction count (one wave issues an instruction every ~5.7 cycles, an s_nop included --
// tools/microbench/chain_probe.hip), and operands that die in the swap need no copies
__device__ __forceinline__ void piv_swap2(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void piv_swap4(float& a, float& b, float& c, float& d, float& e, float& f, float& g, float& h) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
        "v_permlane32_swap_b32 %6, %7\n\ts_nop 1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}
// 16 bytes another workgroup of this launch has stored (and signalled): two 8-byte agent-scope loads -- they bypass this CU's
// L1, which may still hold the line from an earlier step, so the reader needs no L1 invalidate (acquire fence) at all
__device__ __forceinline__ V4<float> ld16_handoff(const float* __restrict__ p) {
    const unsigned long long a = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    V4<float> v;
    v.v[0] = __uint_as_float((unsigned int)a); v.v[1] = __uint_as_float((unsigned int)(a >> 32));
    v.v[2] = __uint_as_float((unsigned int)b); v.v[3] = __uint_as_float((unsigned int)(b >> 32));
    return v;
}
// GSYNC = false: called by ALL waves of the workgroup; W, W^T and pcol must be free (a barrier since their last use).
// GSYNC = true : called by waves 0..3 while the rest of the workgroup does something else; src_blk is a 64x64 tile in
//                LDS (row stride 64), gwords two LDS ints zeroed at kernel start, gcall the number of earlier calls in
//                this kernel.  The caller synchronises afterwards.
// ROLES = false: this instantiation is only ever executed by waves 4.. of the workgroup (they stage the tile, join the
//                barrier and leave): the role code is not even compiled in -- a caller that has already branched on the
//                wave number keeps the roles' registers out of its other branch this way.
// IN_W (with GSYNC): the tile already sits in the W area (row stride SPD_LS), src_blk is not read.
template <bool GSYNC = false, bool ROLES = true, bool IN_W = false, bool WMAX = false>
__device__ __forceinline__ void wg_pivot_block_mfma(const float* __restrict__ src_blk, float* __restrict__ W,
                                                    float* __restrict__ WT, float* __restrict__ pcol,
                                                    int* __restrict__ flag, const int kbase,
                                                    int* __restrict__ gwords = nullptr, const int gcall = 0,
                                                    const float diag_add = 0.f,         // (!GSYNC: added to the tile's diagonal as it is staged)
                                                    const bool handoff = false,         // (!GSYNC: src_blk was stored by another workgroup of this launch)
                                                    float* __restrict__ wmax_out = nullptr) {      // [4]: waves 1, 2 leave max |W| of what they store (LDS)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    float* const svals = pcol;                                        // [64]: 1 / sqrt(pivot)
    int* const words = GSYNC ? gwords : (int*)(pcol + 64);            // [0] panels published, [1] consumers done
    float* const queue = WT;                                          // [16][4][64] coefficient columns
    float* const xqueue = W;                                          // [8][4][64] pivot rows of panels 0..7 (over rows 0..31 of the staged tile)
    const int rbase = GSYNC ? 16 * gcall : 0, dbase = GSYNC ? 2 * gcall : 0;
    if constexpr (!GSYNC) {
        // the tile into the W area (row stride SPD_LS): one coalesced pass by the whole workgroup
        for (int i = tid * 4; i < LQP_BLK; i += (int)blockDim.x * 4) {
            V4<float> v = handoff ? ld16_handoff(src_blk + i) : *(const V4<float>*)(src_blk + i);
            if (diag_add != 0.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] += ((i >> 6) == (i & 63) + e) ? diag_add : 0.f;
            }
            *(V4<float>*)(W + (i >> 6) * SPD_LS + (i & 63)) = v;
        }
        if (tid == 0) { words[0] = 0; words[1] = 0; }
        __syncthreads();
    }
    if constexpr (!ROLES) return;
    if (w > 3) return;
    PIV_STAMP(w * 32);
    const float* const Tl = (GSYNC && !IN_W) ? src_blk : W;
    constexpr int ld = (GSYNC && !IN_W) ? 64 : SPD_LS;
    // Lane-dependent address parts, made opaque once per call: everything below is base + compile-time offset (the
    // offset field of the LDS instructions).  Left to itself the compiler forms each of the ~200 addresses (and the 16
    // values of the identity) as a loop invariant of the CALLER's pivot-step loop, keeps them in registers for the whole
    // kernel and spills the caller's tiles: 245 spilled registers in the resident sweep.
    int lane_o = lane, li_o = li, lh_o = lh;
    asm volatile("" : "+v"(lane_o), "+v"(li_o), "+v"(lh_o));
    float* const qcoef = queue + lane_o;                              // coefficient column t of panel P: qcoef[(4 P + t) * 64]
    float* const qrow = xqueue + lane_o;
    const float* const tile_o = Tl + (4 * lh_o) * ld + li_o;          // element (quad_row(q, lh), li) at + ((q&3) + 8 (q>>2)) * ld
    float* const w_o = W + (4 * lh_o) * SPD_LS + li_o;                // W[8 a + 4 lh + e][li] at + (8 a + e) * SPD_LS
    float* const wt_o = WT + li_o * SPD_LS + 4 * lh_o;                // W^T[li][8 a + 4 lh + e] at + 8 a + e
    float* const wtz_o = WT + (4 * lh_o) * SPD_LS + li_o;             // W^T[8 a + 4 lh + e][li] (the zero quadrant)
    const float* const dg_o = xqueue + (4 * lh_o) * 65;               // pivot of row r: xqueue[r * 64 + r]
    constexpr auto qoff = [](const int q) { return (q & 3) + 8 * (q >> 2); };
    // queue entry (panel P, column t): 64 floats.  Panels 8..15 start at row 32 of their area (68-float rows), so that rows
    // 0..31 of W and W^T carry nothing of the lower half: quadrant (0,0) can be stored while the lower half is eliminated.
    constexpr auto qidx = [](const int P, const int t) { return (P * 4 + t) * 64 + (P >= 8 ? 32 * SPD_LS - 2048 : 0); };
#ifndef LQP_PIV_ABL
#define LQP_PIV_ABL 0      // timing experiments of the chain only (results wrong): 1 no matrix instructions, 2 no look-ahead reads, 4 no queue stores, 8 no swaps
#endif
    auto mfma = [](const float a, const float b, const f32x16 c) -> f32x16 {
#if LQP_PIV_ABL & 1
        f32x16 r = c; r[0] += a * b; return r;
#else
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
    };
    auto wait_published = [&](const int target) {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    // The panel's column steps on its four columns x[] (lane = row).  The dependent chain of a column is
    //   v_readlane (pivot) -> v_rcp -> v_mul (coefficient column) -> v_fma (next column) -> v_readlane (next pivot),
    // ~33 cycles per link; but ONE wave issues an instruction only every ~5.7 cycles whatever it is (s_nop included:
    // tools/microbench/chain_probe.hip), and a panel is ~100 instructions: the chain wave is bound by its instruction
    // COUNT.  So it carries nothing it can leave to others -- the mask that zeroes the coefficients of the rows above the
    // pivot (the columns' own entries there are dead, only the augmented part needs it), the scales 1 / sqrt(pivot) and
    // the sign check of the pivots are the consumer waves' (they find the pivots on the diagonal of the pivot-row queue) --
    // and the coefficient columns are formed, queued and used NEGATED (ncu = -column / pivot: the negation rides on the
    // multiply as a source modifier, every user wants the negative: fma(ncu, row entry, x), A operand of the matrix
    // instructions), without the mask.
    // column step t of a panel: ncu[t] and the updates of the panel's later columns
    auto column_step = [&](const int c0, const int t, float (&x)[4], float (&ncu)[4]) {
        const int c = c0 + t;
        ncu[t] = x[t] * -__builtin_amdgcn_rcpf(piv_readlane(x[t], c));
#pragma unroll
        for (int t2 = t + 1; t2 < 4; ++t2) x[t2] = __builtin_fmaf(ncu[t], piv_readlane(x[t2], c), x[t2]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // The matrix instructions and the chain.  A v_mfma_f32_32x32x2 holds the pipe for 64 cycles; issued by ONE wave,
    // a second one right behind it (or a read of its result) stalls that wave until the pipe is free -- every instruction of
    // the chain waits with it (measured: the four of a panel cost their full 4 x 64 cycles on top of the ~530 of the rest).
    // So (1) the four are issued >= 11 instructions apart, each where its operands have just become final, and nobody reads
    // an accumulator less than that after it was written; (2) the rows of the NEXT panel leave the accumulators BEFORE this
    // panel's instructions touch them (they hold the updates through the previous panel) and take this panel's rank-4
    // update on the vector unit: y_t += sum_k ncoef_k[r0 + t] * (pivot row k), k = 0..3, the coefficients read back from
    // the queue (one broadcast ds_read_b128 per column) -- same products, same order as the matrix instructions' chain.
    // The accumulators' own copies of those rows get the update too (ncu is not masked) but nobody reads them again.
    auto coef4 = [&](const int P, const int k, const int r0) -> V4<float> {
#if LQP_PIV_ABL & 2
        return V4<float>{{0.25f, 0.5f, 0.25f, 0.5f}};
#else
        return *(const V4<float>*)(queue + qidx(P, k) + r0);      // (uniform address)
#endif
    };
    // y_t += sum_k ncoef_k[t] * x_k, two rows per instruction (v_pk_fma_f32: the chain is bound by its instruction count);
    // per row the same four fmas in the same order
    auto rank4 = [](float (&y)[4], const V4<float>& k0, const V4<float>& k1, const V4<float>& k2, const V4<float>& k3,
                    const float (&x)[4]) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 y01 = f2{y[0], y[1]}, y23 = f2{y[2], y[3]};
        const V4<float>* kk[4] = {&k0, &k1, &k2, &k3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f2 xk = f2{x[k], x[k]};
            y01 = __builtin_elementwise_fma(f2{kk[k]->v[0], kk[k]->v[1]}, xk, y01);
            y23 = __builtin_elementwise_fma(f2{kk[k]->v[2], kk[k]->v[3]}, xk, y23);
        }
        y[0] = y01[0]; y[1] = y01[1]; y[2] = y23[0]; y[3] = y23[1];
    };
    int pub = rbase;                            // (a running count, opaque: sixteen constants in sixteen registers otherwise)
    asm volatile("" : "+v"(pub));
    auto publish = [&](const int count) {      // (the queue writes of that panel have long landed where this is called)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pub += 1;
        (void)count;
        if (lane == 0) __hip_atomic_store(words, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (w == 0) {
        // ================= chain, upper half: quadrants (0,0), (0,1) of the Schur complement =================
        f32x16 S00, S01;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            S00[q] = tile_o[qoff(q) * ld];
            S01[q] = tile_o[qoff(q) * ld + 32];
        }
        if constexpr (!GSYNC || IN_W) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the pivot-row queue overwrites these rows)
        // rows c0 .. c0+3 of S as lane vectors (lane = column; by symmetry = the panel's columns, lane = row)
        float x[4];
        {
            float h0 = S01[0], h1 = S01[1], h2 = S01[2], h3 = S01[3];
            x[0] = S00[0]; x[1] = S00[1]; x[2] = S00[2]; x[3] = S00[3];
            piv_swap4(x[0], h0, x[1], h1, x[2], h2, x[3], h3);
        }
        float pend_a = 0.f, pend_b = 0.f;       // operands of the previous panel's last matrix instruction (on S01)
        PIV_STAMP(1);
#pragma unroll
        for (int P = 0; P < 8; ++P) {
            const int c0 = 4 * P, n0 = c0 + 4, q0 = 4 * (n0 / 8), LH = (n0 / 4) & 1;
            __builtin_amdgcn_sched_barrier(0);                       // (one scheduling region per panel)
            float ncu[4], y[4];
            if (P > 0) {
                publish(P);
                if (P < 7) S01 = mfma(pend_a, pend_b, S01);         // (second half of panel P-1, column half 1; nobody reads it after panel 6)
            }
            column_step(c0, 0, x, ncu);
            column_step(c0, 1, x, ncu);
            if (P < 7) {
                // the next panel's rows as the accumulators hold them: through panel P-1
                float z[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { y[t] = S00[q0 + t]; z[t] = S01[q0 + t]; }
                piv_swap4(y[0], z[0], y[1], z[1], y[2], z[2], y[3], z[3]);
                if (LH) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t] = z[t];
                }
            }
            // columns 0, 1 are final: queued, then (the coefficient columns in place: they die here) turned into the operands
            // of their half of the rank-4 update -- A = [ncu0 | ncu1] over rows 0..31, B = the pivot rows' two column halves
            qcoef[qidx(P, 0)] = ncu[0]; qcoef[qidx(P, 1)] = ncu[1];
            qrow[qidx(P, 0)] = x[0]; qrow[qidx(P, 1)] = x[1];
            float b01_lo = x[0], b01_hi = x[1];
            V4<float> k0, k1;
            if (P < 7) {
                k0 = coef4(P, 0, n0); k1 = coef4(P, 1, n0);
                piv_swap2(ncu[0], ncu[1], b01_lo, b01_hi);
                S00 = mfma(ncu[0], b01_lo, S00);
            }
            __builtin_amdgcn_sched_barrier(0);
            column_step(c0, 2, x, ncu);
            column_step(c0, 3, x, ncu);
            qcoef[qidx(P, 2)] = ncu[2]; qcoef[qidx(P, 3)] = ncu[3];
            qrow[qidx(P, 2)] = x[2]; qrow[qidx(P, 3)] = x[3];
            if (P < 7) {
                S01 = mfma(ncu[0], b01_hi, S01);
                const V4<float> k2 = coef4(P, 2, n0), k3 = coef4(P, 3, n0);
                rank4(y, k0, k1, k2, k3, x);
                // the other half of the rank-4 update; its second instruction waits for the start of the next panel
                piv_swap2(ncu[2], ncu[3], x[2], x[3]);
                S00 = mfma(ncu[2], x[2], S00);
                pend_a = ncu[2]; pend_b = x[3];
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = y[t];
            }
            PIV_STAMP(2 + P);
        }
        publish(8);
        PIV_STAMP(20);
        return;
    }
    if (w == 3) {
        // ================= quadrant (1,1) of the Schur complement: follows panels 0..7, then the chain =================
        f32x16 S11;
#pragma unroll
        for (int q = 0; q < 16; ++q) S11[q] = tile_o[(32 + qoff(q)) * ld + 32];
#pragma unroll
        for (int P = 0; P < 8; ++P) {
            __builtin_amdgcn_sched_barrier(0);
            wait_published(rbase + P + 1);
            float cf[4], xr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { cf[t] = qcoef[qidx(P, t)]; xr[t] = qrow[qidx(P, t)]; }
            // A = [ncf0 | ncf1] over rows 32..63, B = the pivot rows' columns 32..63: the `hi` results
            piv_swap4(cf[0], cf[1], cf[2], cf[3], xr[0], xr[1], xr[2], xr[3]);
            S11 = mfma(cf[1], xr[1], S11);
            S11 = mfma(cf[3], xr[3], S11);
        }
        pub += 8;
        // rows 32 .. 35 (lanes 0-31: columns the elimination has left; the consumers' mask zeroes what they produce)
        float x[4];
        {
            float h0 = S11[0], h1 = S11[1], h2 = S11[2], h3 = S11[3];
            x[0] = S11[0]; x[1] = S11[1]; x[2] = S11[2]; x[3] = S11[3];
            piv_swap4(x[0], h0, x[1], h1, x[2], h2, x[3], h3);
        }
        PIV_STAMP(96 + 1);
#pragma unroll
        for (int P = 8; P < 16; ++P) {
            const int c0 = 4 * P, n0 = c0 + 4, rr = n0 - 32, q0 = 4 * (rr / 8), LH = (rr / 4) & 1;
            __builtin_amdgcn_sched_barrier(0);
            float ncu[4], y[4];
            if (P > 8) publish(P);
            column_step(c0, 0, x, ncu);
            column_step(c0, 1, x, ncu);
            if (P < 15) {
                // the next panel's rows as the accumulator holds them (through panel P-1)
                if (LH) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) y[t] = S11[q0 + t];
                } else {
                    float z[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { y[t] = S11[q0 + t]; z[t] = S11[q0 + t]; }
                    piv_swap4(y[0], z[0], y[1], z[1], y[2], z[2], y[3], z[3]);
                }
            }
            qcoef[qidx(P, 0)] = ncu[0]; qcoef[qidx(P, 1)] = ncu[1];
            qrow[qidx(P, 0)] = x[0]; qrow[qidx(P, 1)] = x[1];
            V4<float> k0, k1;
            float b01_lo = x[0], b01_hi = x[1];
            if (P < 15) {
                k0 = coef4(P, 0, n0); k1 = coef4(P, 1, n0);
                piv_swap2(ncu[0], ncu[1], b01_lo, b01_hi);
                S11 = mfma(ncu[1], b01_hi, S11);
            }
            __builtin_amdgcn_sched_barrier(0);
            column_step(c0, 2, x, ncu);
            column_step(c0, 3, x, ncu);
            qcoef[qidx(P, 2)] = ncu[2]; qcoef[qidx(P, 3)] = ncu[3];
            qrow[qidx(P, 2)] = x[2]; qrow[qidx(P, 3)] = x[3];
            if (P < 15) {
                const V4<float> k2 = coef4(P, 2, n0), k3 = coef4(P, 3, n0);
                rank4(y, k0, k1, k2, k3, x);
                piv_swap2(ncu[2], ncu[3], x[2], x[3]);
                S11 = mfma(ncu[3], x[3], S11);
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = y[t];
            }
            PIV_STAMP(96 + 2 + (P - 8));
        }
        publish(16);
        PIV_STAMP(96 + 20);
        return;
    }
    // ================= waves 1, 2: the augmented part =================
    // (one instantiation per role: which quadrant a panel's rows come from and which ones it updates are then
    //  compile-time facts of straight-line code)
    auto consumer = [&](auto colblock0_tag) {
        constexpr bool CB0 = decltype(colblock0_tag)::value;
        f32x16 Wa, Wb;                   // wave 1: quadrants (0,0), (1,0); wave 2: (1,1), -
#pragma unroll
        for (int q = 0; q < 16; ++q) { Wa[q] = qoff(q) + 4 * lh_o == li_o ? 1.f : 0.f; Wb[q] = 0.f; }
        // row scales 1 / sqrt(pivot): pivot r sits on the diagonal of the pivot-row queue (entry r of row r).  They are taken
        // as the panels arrive -- after every second one the eight rows 8 g .. 8 g + 7 are through, every lane reads the four
        // of its half -- so that only the last group is left behind the chain; a pivot that is not positive (the matrix is
        // not positive definite) is recorded like in wg_pivot_block.
        float sc[CB0 ? 32 : 16];
        int bad = 0;
#pragma unroll
        for (int P = CB0 ? 0 : 8; P < 16; ++P) {                 // (column block 1 is untouched by the upper half)
            const int c0 = 4 * P, I = c0 / 32, rr = c0 % 32, q0 = 4 * (rr / 8), LH = (rr / 4) & 1;
            __builtin_amdgcn_sched_barrier(0);
            wait_published(rbase + P + 1);
            float cf[4];
            {
                int lane_p = lane_o;               // (opaque per panel: see the lane bases above)
                asm volatile("" : "+v"(lane_p));
#pragma unroll
                for (int t = 0; t < 4; ++t) {      // the rows above a pivot take no part in its elimination
                    const float c = qcoef[qidx(P, t)];
                    cf[t] = lane_p > c0 + t ? c : 0.f;
                }
            }
            // the panel's 4x4 unit lower triangle: (negated) coefficient of row c0+k at column step j
            const float t10 = piv_readlane(cf[0], c0 + 1), t20 = piv_readlane(cf[0], c0 + 2), t30 = piv_readlane(cf[0], c0 + 3);
            const float t21 = piv_readlane(cf[1], c0 + 2), t31 = piv_readlane(cf[1], c0 + 3), t32 = piv_readlane(cf[2], c0 + 3);
            // the panel's rows of this column block, as the column steps leave them (half LH of registers q0 .. q0+3)
            float r0, r1, r2, r3;
            if (CB0 && I == 1) { r0 = Wb[q0]; r1 = Wb[q0 + 1]; r2 = Wb[q0 + 2]; r3 = Wb[q0 + 3]; }
            else { r0 = Wa[q0]; r1 = Wa[q0 + 1]; r2 = Wa[q0 + 2]; r3 = Wa[q0 + 3]; }
            r1 = __builtin_fmaf(t10, r0, r1);
            r2 = __builtin_fmaf(t21, r1, __builtin_fmaf(t20, r0, r2));
            r3 = __builtin_fmaf(t32, r2, __builtin_fmaf(t31, r1, __builtin_fmaf(t30, r0, r3)));
            // (in place: r0 := [r0.lo | r1.lo], r1 := [r0.hi | r1.hi], ...; the coefficients arrive negated)
            piv_swap4(r0, r1, r2, r3, cf[0], cf[1], cf[2], cf[3]);
            const float b_01 = LH ? r1 : r0, b_23 = LH ? r3 : r2;
            const float a0_01 = cf[0], a1_01 = cf[1], a0_23 = cf[2], a1_23 = cf[3];
            if (CB0) {
                if (I == 0) {
                    Wa = mfma(a0_01, b_01, Wa);
                    Wb = mfma(a1_01, b_01, Wb);
                    Wa = mfma(a0_23, b_23, Wa);
                    Wb = mfma(a1_23, b_23, Wb);
                } else {
                    Wb = mfma(a1_01, b_01, Wb);
                    Wb = mfma(a1_23, b_23, Wb);
                }
            } else {
                Wa = mfma(a1_01, b_01, Wa);
                Wa = mfma(a1_23, b_23, Wa);
            }
            if (P & 1) {
                const int g = P / 2;                     // rows 8 g + e + 4 lh, e = 0..3
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rowc = 8 * g + e;
                    const float d = dg_o[rowc * 65 + (rowc >= 32 ? 32 * SPD_LS - 2048 : 0)];
                    bad = (!(d > 0.f) && bad == 0) ? rowc + 4 * lh_o + 1 : bad;
                    sc[(CB0 ? 4 * g : 4 * (g - 4)) + e] = __builtin_amdgcn_rsqf(d);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_published(rbase + 16);
        {
            const unsigned long long mb = __ballot(bad != 0);
            if (mb != 0ull && lane == 0 && flag[0] == 0) flag[0] = kbase + __builtin_amdgcn_readlane(bad, (int)__builtin_ctzll(mb));
        }
        // the queues sit in the W and W^T areas: both consumers must be through with them before anybody writes there
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(words + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < dbase + 2)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
#ifdef LQP_PIV_DEBUG_STOP
        return;
#endif
        // W = (row scale) x (unit lower triangle), zero above the diagonal; W^T
        [[maybe_unused]] float wmx = 0.f;
        auto store_quadrant = [&](const f32x16& v, const int I, const int J, const float* scq) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                V4<float> v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (above the diagonal the unit triangle is exactly zero: 0 - coef * 0 at every step, no select needed)
                    v4.v[e] = v[4 * a + e] * scq[4 * a + e];
                    if constexpr (WMAX) wmx = tmax(wmx, tabs(v4.v[e]));
                    w_o[(32 * I + 8 * a + e) * SPD_LS + 32 * J] = v4.v[e];
                }
                *(V4<float>*)(wt_o + (32 * J) * SPD_LS + 32 * I + 8 * a) = v4;      // (four consecutive entries of a W^T row)
            }
        };
        PIV_STAMP(w * 32 + 1);
        if constexpr (CB0) {
            store_quadrant(Wa, 0, 0, sc);
            store_quadrant(Wb, 1, 0, sc + 16);
            PIV_STAMP(w * 32 + 2);
        } else {
            store_quadrant(Wa, 1, 1, sc);
#pragma unroll
            for (int q = 0; q < 16; ++q) {                       // quadrant (0,1) of W = quadrant (1,0) of W^T = 0
                w_o[qoff(q) * SPD_LS + 32] = 0.f;
                wtz_o[(32 + qoff(q)) * SPD_LS] = 0.f;
            }
        }
        if constexpr (WMAX) {
            wmx = wave_max(wmx);
            if (lane == 0) wmax_out[w] = wmx;
        }
    };
    if (w == 1) consumer(std::true_type());
    else consumer(std::false_type());
}

// the pivot block of every factorisation below: on the matrix cores unless built with -DLQP_PIV_MFMA=0
template <int NWP, bool MFMA = (LQP_PIV_MFMA != 0)>
__device__ __forceinline__ void wg_pivot(const float* __restrict__ src_blk, float* __restrict__ W, float* __restrict__ WT,
                                         float* __restrict__ pcol, int* __restrict__ flag, const int kbase) {
    if constexpr (MFMA) wg_pivot_block_mfma<false>(src_blk, W, WT, pcol, flag, kbase);
    else wg_pivot_block<NWP>(src_blk, W, WT, pcol, flag, kbase);
}

// ---- block symmetric sweep: Hs (lower blocks of an SPD matrix) -> -inverse, in place ----
// info: 0, or 1 + index of the first non-positive pivot.
// NP == 1: the whole sweep, in place.  NP > 1: pivot steps [k0, k1) only, OUT of place (every block of the matrix
// is rewritten by every step: read from Hs, written to Hdst), with the tile tasks of a step shared between NP
// workgroups (`part` of NP; each of them factorises the pivot block and stages the panel for itself).  One launch
// per step then lets NP workgroups work on one matrix with no synchronisation inside a kernel.
template <int NP = 1>
__device__ __forceinline__ void wg_spd_sweep(float* __restrict__ Hs, const int K, int* __restrict__ info, char* smem,
                                             unsigned long long* __restrict__ dbg = nullptr, float* Hdst = nullptr,
                                             const int k0 = 0, const int k1 = 0, const int part = 0,
                                             float* __restrict__ Wg = nullptr, const int pivot_tasks = 48) {
    static_assert(NP == 1 || NP == 2, "one or two workgroups per matrix");
    float* const Hd = NP == 1 ? Hs : Hdst;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = tid >> 4, cq = tid & 15, li = lane & 31, lh = lane >> 5;
    const int nslot = K > 1 ? K - 1 : 1;
    float* Y = (float*)smem;                       // nslot x 64 x LS: the panel of the current pivot block
    float* W = Y + (size_t)nslot * 64 * SPD_LS;    // L^-1 of the pivot block (lower)
    float* WT = W + 64 * SPD_LS;                   // its transpose
    float* pcol = WT + 64 * SPD_LS;                // [2][64] pivot column, by column parity
    int* flag = (int*)(pcol + PIV_LDS);
    if (tid == 0) flag[0] = 0;
    unsigned long long tp = 0, ty = 0, tu = 0, t0 = 0, tb = dbg ? clock64() : 0;   // debug cycle counters

    for (int k = (NP == 1 ? 0 : k0); k < (NP == 1 ? K : k1); ++k) {
        if (dbg) t0 = clock64();
        // ---- pivot block -> W, W^T | panel blocks A_ik (i != k) -> LDS: slot s holds P_i = A_ik, i.e. block (k, i)
        //      transposed when i < k ----
        if constexpr (NP == 1 && LQP_PIV_MFMA != 0) {
            // the matrix-core pivot block works on waves 0..3; the other twelve stage the panel meanwhile (nothing is held
            // in registers across the pivot block: its accumulators need them)
            wg_pivot<LQP_PIV_WAVES>(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK, W, WT, pcol, flag, k * 64);
            if (w >= 4) {
                for (int v0 = tid - 256; v0 < 1024 * (K - 1); v0 += LQP_NT - 256) {
                    const int s = v0 >> 10, v = v0 & 1023, rr = v >> 4, cc = v & 15;
                    const int i = s < k ? s : s + 1;
                    const int blk = i > k ? sym_idx(i, k, K) : sym_idx(k, i, K);
                    const V4<float> pv = *(const V4<float>*)(Hs + (size_t)blk * LQP_BLK + v * 4);
                    float* Ys = Y + (size_t)s * 64 * SPD_LS;
                    if (s >= k) {
                        *(V4<float>*)(Ys + rr * SPD_LS + cc * 4) = pv;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) Ys[(cc * 4 + e) * SPD_LS + rr] = pv.v[e];
                    }
                }
            }
        } else {
        // (panel blocks into registers; they land while the pivot block is factorised)
        V4<float> preg[SPD_MAXK - 1];
#pragma unroll
        for (int s = 0; s < SPD_MAXK - 1; ++s) {
            if (s < K - 1) {
                const int i = s < k ? s : s + 1;
                const int blk = i > k ? sym_idx(i, k, K) : sym_idx(k, i, K);
                preg[s] = *(const V4<float>*)(Hs + (size_t)blk * LQP_BLK + tid * 4);
            }
        }
        if (NP == 1 || k == 0) {
            wg_pivot<LQP_PIV_WAVES>(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK, W, WT, pcol, flag, k * 64);
        } else {      // W, W^T of this pivot block were prepared by the previous launch (lookahead below)
            for (int i = tid * 4; i < 2 * 64 * SPD_LS; i += LQP_NT * 4) *(V4<float>*)(W + i) = *(const V4<float>*)(Wg + i);
        }
#pragma unroll
        for (int s = 0; s < SPD_MAXK - 1; ++s) {
            if (s < K - 1) {
                float* Ys = Y + (size_t)s * 64 * SPD_LS;
                if (s >= k) {
                    *(V4<float>*)(Ys + r * SPD_LS + cq * 4) = preg[s];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ys[(cq * 4 + e) * SPD_LS + r] = preg[s].v[e];
                }
            }
        }
        }
        __syncthreads();
        if (dbg) { const unsigned long long t = clock64(); tp += t - t0; t0 = t; }
        // ---- Y_i = P_i W^T (in place: all products first, then the writes) ----
        {
            const int ntask = (K - 1) * 4;
            f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                if (task < ntask) {
                    const int s = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                    const float* Xp = Y + ((size_t)s * 64 + 32 * qi) * SPD_LS;
                    acc[u] = qj == 0 ? spd_quadrant<1, false>(Xp, W) : spd_quadrant(Xp, W + 32 * SPD_LS);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                if (task < ntask) {
                    const int s = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                    float* dst = Y + ((size_t)s * 64 + 32 * qi) * SPD_LS + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) dst[quad_row(q, lh) * SPD_LS] = acc[u][q];
                }
            }
            __syncthreads();
        }
        if (dbg) { const unsigned long long t = clock64(); ty += t - t0; t0 = t; }
        // ---- all writes of this step (every task only READS LDS) ----
        //   [0, nupd):            A_ij -= Y_i Y_j^T            (i >= j, both != k)
        //   [nupd, nupd + npan):  A_ik  = Y_i W  (stored as block (i,k), or as its transpose W^T Y_i^T in (k,i))
        //   last 4:               A_kk  = -(W^T W)
        {
            const int npair = (K - 1) * K / 2;
            const int nupd = npair * 4, nrest = (K - 1) * 4 + 4;
            // update tiles: the C quadrant of the NEXT task is requested before the MFMA chain of this one, so the
            // waves of a SIMD do not all sit in their load phase (then all in their MFMA phase) together
            const int plook = (NP > 1 && k + 1 < K) ? k * (k + 1) / 2 + k : -1;      // pair (slot k, slot k) = tile (k+1, k+1)
            auto upd_addr = [&](const int task, int& si, int& sj, bool& skip, bool& mirror) -> size_t {
                const int qi = (task >> 1) & 1, qj = task & 1, p = task >> 2;
                si = 0;
                while ((si + 1) * (si + 2) / 2 <= p) ++si;
                sj = p - si * (si + 1) / 2;
                skip = si == sj && qi == 0 && qj == 1;          // diagonal tile: mirrored from its (1,0) quadrant
                if (NP > 1 && p == plook) skip = true;          // the next pivot tile is the lookahead's
                mirror = si == sj && qi == 1 && qj == 0;
                const int i = si < k ? si : si + 1, j = sj < k ? sj : sj + 1;
                return (size_t)sym_idx(i, j, K) * LQP_BLK;
            };
            auto upd_load = [&](const float* T0, const int task, f32x16& c) {
                const float* C = T0 + (32 * ((task >> 1) & 1)) * 64 + 32 * (task & 1) + li;
#pragma unroll
                for (int q = 0; q < 16; ++q) c[q] = C[quad_row(q, lh) * 64];
            };
            // NP == 2: workgroup 0 takes the head of the update list and the panel / pivot tasks (they need W, W^T of
            // this step), workgroup 1 the tail -- after it has updated the NEXT pivot tile, factorised it and left
            // its W, W^T in Wg for the next launch (that costs about as much as `pivot_tasks` tile tasks)
            int ulo = 0, uhi = nupd;
            bool do_rest = true;
            if (NP > 1) {
                // workgroup 1's share of the update list: half of what is left of (all tasks - the pivot's worth)
                int n1 = (nupd + nrest - (k + 1 < K ? pivot_tasks : 0)) / 2;
                n1 = n1 < 0 ? 0 : (n1 > nupd ? nupd : n1);
                const int cut = nupd - n1;
                if (part == 0) uhi = cut; else { ulo = cut; do_rest = false; }
                if (part == 1 && k + 1 < K) {
                    float* Tn = Hd + (size_t)sym_idx(k + 1, k + 1, K) * LQP_BLK;
                    if (w < 4 && w != 1) {                      // quadrants (0,0), (1,0) + mirror, (1,1); slot of k+1 is k
                        const int qi = w >> 1, qj = w & 1;
                        const float* Cs = Hs + (size_t)sym_idx(k + 1, k + 1, K) * LQP_BLK + (32 * qi) * 64 + 32 * qj + li;
                        f32x16 cur;
#pragma unroll
                        for (int q = 0; q < 16; ++q) cur[q] = Cs[quad_row(q, lh) * 64];
                        const f32x16 acc = spd_quadrant(Y + ((size_t)k * 64 + 32 * qi) * SPD_LS, Y + ((size_t)k * 64 + 32 * qj) * SPD_LS);
                        cur -= acc;
                        float* C = Tn + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                        for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                        if (w == 2) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) Tn[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                        }
                    }
                    __threadfence_block();
                    __syncthreads();
                    wg_pivot<LQP_PIV_WAVES>(Tn, W, WT, pcol, flag, (k + 1) * 64);
                    __syncthreads();
                    for (int i = tid * 4; i < 2 * 64 * SPD_LS; i += LQP_NT * 4) *(V4<float>*)(Wg + i) = *(const V4<float>*)(W + i);
                }
            }
            int task = ulo + __builtin_amdgcn_readfirstlane(w);
            int si = 0, sj = 0; bool skip = false, mirror = false;
            size_t T0 = 0;
            f32x16 nxt;
            if (task < uhi) { T0 = upd_addr(task, si, sj, skip, mirror); if (!skip) upd_load(Hs + T0, task, nxt); }
            while (task < uhi) {
                const int qi = (task >> 1) & 1, qj = task & 1;
                f32x16 cur = nxt;
                float* Tc = Hd + T0;
                const int csi = si, csj = sj;
                const bool cskip = skip, cmirror = mirror;
                const int ntask = task + LQP_NW;
                if (ntask < uhi) { T0 = upd_addr(ntask, si, sj, skip, mirror); if (!skip) upd_load(Hs + T0, ntask, nxt); }
                if (!cskip) {
                    const f32x16 acc = spd_quadrant(Y + ((size_t)csi * 64 + 32 * qi) * SPD_LS,
                                                    Y + ((size_t)csj * 64 + 32 * qj) * SPD_LS);
                    cur -= acc;
                    float* C = Tc + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                    if (cmirror) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) Tc[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                    }
                }
                task = ntask;
            }
            // the new panel column and the pivot block (stores only)
            for (int t2 = do_rest ? __builtin_amdgcn_readfirstlane(w) : nrest; t2 < nrest; t2 += LQP_NW) {
                const int qi = (t2 >> 1) & 1, qj = t2 & 1;
                if (t2 < nrest - 4) {
                    const int s = t2 >> 2;
                    const int i = s < k ? s : s + 1;
                    const float* Ys = Y + (size_t)s * 64 * SPD_LS;
                    f32x16 acc;
                    float* C;
                    if (i > k) {          // Y W: column c >= 32 (qj == 1) only sees k >= 32
                        acc = qj == 1 ? spd_quadrant<1, true>(Ys + (32 * qi) * SPD_LS, WT + 32 * SPD_LS)
                                      : spd_quadrant(Ys + (32 * qi) * SPD_LS, WT);
                        C = Hd + (size_t)sym_idx(i, k, K) * LQP_BLK;
                    } else {              // W^T Y^T: row r >= 32 (qi == 1) only sees k >= 32
                        acc = qi == 1 ? spd_quadrant<1, true>(WT + 32 * SPD_LS, Ys + (32 * qj) * SPD_LS)
                                      : spd_quadrant(WT, Ys + (32 * qj) * SPD_LS);
                        C = Hd + (size_t)sym_idx(k, i, K) * LQP_BLK;
                    }
                    C += (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = acc[q];
                } else {
                    const f32x16 acc = (qi | qj) ? spd_quadrant<1, true>(WT + (32 * qi) * SPD_LS, WT + (32 * qj) * SPD_LS)
                                                 : spd_quadrant(WT, WT);
                    float* C = Hd + (size_t)sym_idx(k, k, K) * LQP_BLK + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = -acc[q];
                }
            }
        }
        __syncthreads();
        if (dbg) { const unsigned long long t = clock64(); tu += t - t0; }
    }
    if (dbg && tid == 0) { dbg[4] = tp; dbg[5] = ty; dbg[6] = tu; dbg[7] = clock64() - tb; }
    if (tid == 0 && flag[0] != 0 && *info == 0) *info = flag[0];
}

// ---- the same sweep for SPD_MAXK < K <= SPD_BIGK (512 < n <= 1024): the panel of a pivot step (K - 1 blocks) no
// longer fits the LDS next to W and W^T.  Y_i = P_i W^T is computed in chunks of SPD_MAXK - 1 blocks and parked in `Yg`
// (global scratch of the matrix, (K - 1) x 4096 floats: it stays in L2); the tile updates A_ij -= Y_i Y_j^T then walk
// over pairs of GROUPS of BIG_G panel blocks -- both groups staged in LDS (over W and W^T, no longer needed in that
// phase), every tile of the pair updated, next pair.
// In place, one workgroup per matrix; LDS as wg_spd_sweep at K = SPD_MAXK.
constexpr int SPD_BIGK = 16;
constexpr int BIG_G = 4;            // two groups of 4 panel blocks: 8 of the 9 block slots (W, W^T are dead by then)
// NP == 2 (few problems: two workgroups per matrix, launches as the barrier between them): pivot steps [k0, k1) only,
// `phases` bit 0 = pivot block + Y phase (both workgroups factorise the pivot block for themselves and take half of the
// panel each), bit 1 = update phase (the group pairs dealt out alternately); one launch per phase.  -(W^T W) then waits in
// the extra block Yg[K - 1] until the update launch copies it over the pivot tile (the partner may still be reading that).
template <int NP = 1>
__device__ __forceinline__ void wg_spd_sweep_big(float* __restrict__ Hs, const int K, int* __restrict__ info, char* smem,
                                                 float* __restrict__ Yg, const int k0 = 0, const int k1 = 0,
                                                 const int phases = 3, const int part = 0) {
    static_assert(NP == 1 || NP == 2, "one or two workgroups per matrix");
    constexpr int CH = SPD_MAXK - 1;                       // panel blocks per chunk of the Y phase
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = tid >> 4, cq = tid & 15, li = lane & 31, lh = lane >> 5;
    float* Y = (float*)smem;
    float* W = Y + (size_t)CH * 64 * SPD_LS;
    float* WT = W + 64 * SPD_LS;
    float* pcol = WT + 64 * SPD_LS;
    int* flag = (int*)(pcol + PIV_LDS);
    if (tid == 0) flag[0] = 0;
    // (two pivot steps per pass over the tiles, NP == 2: phases bit 4 -- this step's Y goes to the SECOND set of panel blocks, Yg + K
    //  blocks --, bit 2 the look-ahead on block column k + 1, bit 3 the fused update of steps k and k + 1: see below)
    float* const Ygw = Yg + ((NP == 2 && (phases & 16)) ? (size_t)K * LQP_BLK : 0);
    for (int k = (NP == 1 ? 0 : k0); k < (NP == 1 ? K : k1); ++k) {
        if (phases & 1) {
            wg_pivot<LQP_PIV_WAVES>(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK, W, WT, pcol, flag, k * 64);
            __syncthreads();
            // this workgroup's part of the panel: slots [s_lo, s_hi)
            const int s_half = (K - 1 + 1) / 2;
            const int s_lo = NP == 1 ? 0 : (part == 0 ? 0 : s_half), s_hi = NP == 1 ? K - 1 : (part == 0 ? s_half : K - 1);
            // ---- Y phase: chunk of the panel -> LDS, Y_i = P_i W^T in place, Y_i -> Yg, A_ik = Y_i W -> its block ----
            for (int c0 = s_lo; c0 < s_hi; c0 += CH) {
                const int cn = (s_hi - c0) < CH ? (s_hi - c0) : CH;
                for (int u = 0; u < cn; ++u) {
                    const int sl = c0 + u, i = sl < k ? sl : sl + 1;
                    const int blk = i > k ? sym_idx(i, k, K) : sym_idx(k, i, K);
                    const V4<float> pv = *(const V4<float>*)(Hs + (size_t)blk * LQP_BLK + tid * 4);
                    float* Ys = Y + (size_t)u * 64 * SPD_LS;
                    if (i > k) {
                        *(V4<float>*)(Ys + r * SPD_LS + cq * 4) = pv;
                    } else {                                    // block (k, i), i < k: P_i is its transpose
#pragma unroll
                        for (int e = 0; e < 4; ++e) Ys[(cq * 4 + e) * SPD_LS + r] = pv.v[e];
                    }
                }
                __syncthreads();
                {
                    const int ntask = cn * 4;                   // <= 28: at most two quadrants per wave
                    f32x16 acc[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                        if (task < ntask) {
                            const int sl = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                            const float* Xp = Y + ((size_t)sl * 64 + 32 * qi) * SPD_LS;
                            acc[u] = qj == 0 ? spd_quadrant<1, false>(Xp, W) : spd_quadrant(Xp, W + 32 * SPD_LS);
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                        if (task < ntask) {
                            const int sl = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                            float* dst = Y + ((size_t)sl * 64 + 32 * qi) * SPD_LS + 32 * qj + li;
#pragma unroll
                            for (int q = 0; q < 16; ++q) dst[quad_row(q, lh) * SPD_LS] = acc[u][q];
                        }
                    }
                    __syncthreads();
                }
                if (NP == 2 && (phases & 32)) {
                    // the panel blocks go out as TWO-HALF operands (lqp_f16x2.hpp: cells [8 hi | 8 mid], one power-of-two scale per
                    // block and 32-row half -- its largest entry: a wave holds four rows of one half): what the fused update and
                    // the look-ahead multiply on the float16 pipe.  This phase's own products stay float32.
                    unsigned int* ymx = (unsigned int*)pcol;                    // [cn][2] (the pivot block is done with pcol)
                    if (tid < 2 * CH) ymx[tid] = 0u;
                    __syncthreads();
                    V4<float> yv[CH];
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        if (u < cn) {
                            yv[u] = *(const V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4);
                            float mx = fmaxf(fmaxf(fabsf(yv[u].v[0]), fabsf(yv[u].v[1])), fmaxf(fabsf(yv[u].v[2]), fabsf(yv[u].v[3])));
                            mx = wave_max(mx);
                            if (lane == 0) atomicMax(ymx + 2 * u + (w >> 3), __float_as_uint(mx));
                        }
                    }
                    __syncthreads();
                    float* ysc = Yg + (size_t)2 * K * LQP_BLK + ((phases & 16) ? 2 * K : 0);      // scales: [set][slot][half]
                    const int cs = cq >> 2, ch = cq & 1, piece = (cq >> 1) & 1;                     // this thread's run of cell (cs, ch)
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        if (u < cn) {
                            float sc, inv;
                            f2_scale_of(__uint_as_float(ymx[2 * u + (w >> 3)]), sc, inv);
                            _Float16 hi[4], mid[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = yv[u].v[e] * sc;
                                hi[e] = (_Float16)a;
                                mid[e] = (_Float16)(a - (float)hi[e]);
                            }
                            char* cell = (char*)(Ygw + (size_t)(c0 + u) * LQP_BLK) + (size_t)r * 256 + 64 * cs + 32 * ch;
                            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                            *(h16x4*)(cell + 8 * piece) = h16x4{hi[0], hi[1], hi[2], hi[3]};
                            *(h16x4*)(cell + 16 + 8 * piece) = h16x4{mid[0], mid[1], mid[2], mid[3]};
                            if ((tid & 511) == 0) ysc[2 * (c0 + u) + (w >> 3)] = sc;
                        }
                    }
                } else
                for (int u = 0; u < cn; ++u)
                    *(V4<float>*)(Ygw + (size_t)(c0 + u) * LQP_BLK + tid * 4) = *(const V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4);
                for (int t2 = __builtin_amdgcn_readfirstlane(w); t2 < cn * 4; t2 += LQP_NW) {
                    const int u = t2 >> 2, qi = (t2 >> 1) & 1, qj = t2 & 1;
                    const int sl = c0 + u, i = sl < k ? sl : sl + 1;
                    const float* Ys = Y + (size_t)u * 64 * SPD_LS;
                    f32x16 acc;
                    float* C;
                    if (i > k) {              // Y W: column c >= 32 (qj == 1) only sees k >= 32
                        acc = qj == 1 ? spd_quadrant<1, true>(Ys + (32 * qi) * SPD_LS, WT + 32 * SPD_LS)
                                      : spd_quadrant(Ys + (32 * qi) * SPD_LS, WT);
                        C = Hs + (size_t)sym_idx(i, k, K) * LQP_BLK;
                    } else {                  // W^T Y^T: row r >= 32 (qi == 1) only sees k >= 32
                        acc = qi == 1 ? spd_quadrant<1, true>(WT + 32 * SPD_LS, Ys + (32 * qj) * SPD_LS)
                                      : spd_quadrant(WT, Ys + (32 * qj) * SPD_LS);
                        C = Hs + (size_t)sym_idx(k, i, K) * LQP_BLK;
                    }
                    C += (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = acc[q];
                }
                __syncthreads();
            }
            if (w < 4 && part == 0) {                           // A_kk = -(W^T W)
                const int qi = (w >> 1) & 1, qj = w & 1;
                const f32x16 acc = (qi | qj) ? spd_quadrant<1, true>(WT + (32 * qi) * SPD_LS, WT + (32 * qj) * SPD_LS)
                                             : spd_quadrant(WT, WT);
                float* C = (NP == 1 ? Hs + (size_t)sym_idx(k, k, K) * LQP_BLK : Ygw + (size_t)(K - 1) * LQP_BLK) + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = -acc[q];
            }
            __threadfence_block();
            __syncthreads();
        }
        if (phases & 2) {
            if (NP > 1 && part == 0)
                *(V4<float>*)(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK + tid * 4) = *(const V4<float>*)(Yg + (size_t)(K - 1) * LQP_BLK + tid * 4);
            // ---- update phase: A_ij -= Y_i Y_j^T for all i >= j, both != k, by pairs of panel groups ----
            const int ng = (K - 1 + BIG_G - 1) / BIG_G;
            int pair_no = 0;
            for (int ga = 0; ga < ng; ++ga) {
                for (int gb = 0; gb <= ga; ++gb, ++pair_no) {
                    if (NP > 1 && (pair_no & 1) != part) continue;
                    const int a0 = BIG_G * ga, b0 = BIG_G * gb;
                    const int an = (K - 1 - a0) < BIG_G ? (K - 1 - a0) : BIG_G, bn = (K - 1 - b0) < BIG_G ? (K - 1 - b0) : BIG_G;
                    const int boff = gb == ga ? 0 : BIG_G;
                    for (int u = 0; u < an; ++u)
                        *(V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4) = *(const V4<float>*)(Yg + (size_t)(a0 + u) * LQP_BLK + tid * 4);
                    if (gb != ga)
                        for (int u = 0; u < bn; ++u)
                            *(V4<float>*)(Y + ((size_t)(BIG_G + u) * 64 + r) * SPD_LS + cq * 4) =
                                *(const V4<float>*)(Yg + (size_t)(b0 + u) * LQP_BLK + tid * 4);
                    __syncthreads();
                    const int npair = gb == ga ? an * (an + 1) / 2 : an * bn;
                    // (the C quadrant of a wave's NEXT task is requested before the MFMA chain of this one)
                    auto decode = [&](const int task, int& ua, int& ub, bool& skip, bool& mirror) -> float* {
                        const int qi = (task >> 1) & 1, qj = task & 1, p = task >> 2;
                        if (gb == ga) {
                            ua = 0;
                            while ((ua + 1) * (ua + 2) / 2 <= p) ++ua;
                            ub = p - ua * (ua + 1) / 2;
                        } else {
                            ua = p / bn;
                            ub = p - ua * bn;
                        }
                        const int si = a0 + ua, sj = b0 + ub;                  // si >= sj
                        skip = si == sj && qi == 0 && qj == 1;                  // diagonal tile: mirrored from its (1,0) quadrant
                        mirror = si == sj && qi == 1 && qj == 0;
                        const int i = si < k ? si : si + 1, j = sj < k ? sj : sj + 1;
                        return Hs + (size_t)sym_idx(i, j, K) * LQP_BLK;
                    };
                    auto load_c = [&](const float* T0, const int task, f32x16& c) {
                        const float* C = T0 + (32 * ((task >> 1) & 1)) * 64 + 32 * (task & 1) + li;
#pragma unroll
                        for (int q = 0; q < 16; ++q) c[q] = C[quad_row(q, lh) * 64];
                    };
                    const int ntask = npair * 4;
                    int task = __builtin_amdgcn_readfirstlane(w);
                    int ua = 0, ub = 0; bool skip = false, mirror = false;
                    float* T0 = nullptr;
                    f32x16 nxt;
                    if (task < ntask) { T0 = decode(task, ua, ub, skip, mirror); if (!skip) load_c(T0, task, nxt); }
                    while (task < ntask) {
                        const int qi = (task >> 1) & 1, qj = task & 1;
                        f32x16 cur = nxt;
                        float* Tc = T0;
                        const int cua = ua, cub = ub;
                        const bool cskip = skip, cmirror = mirror;
                        const int nt = task + LQP_NW;
                        if (nt < ntask) { T0 = decode(nt, ua, ub, skip, mirror); if (!skip) load_c(T0, nt, nxt); }
                        if (!cskip) {
                            const f32x16 acc = spd_quadrant(Y + ((size_t)cua * 64 + 32 * qi) * SPD_LS,
                                                            Y + ((size_t)(boff + cub) * 64 + 32 * qj) * SPD_LS);
                            cur -= acc;
                            float* C = Tc + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                            for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                            if (cmirror) {
#pragma unroll
                                for (int q = 0; q < 16; ++q) Tc[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                            }
                        }
                        task = nt;
                    }
                    __syncthreads();
                }
            }
            __threadfence_block();
        }
        if constexpr (NP == 2) {
            // ---- two pivot steps per pass over the tiles (k + 1 < K).  Step k's update touches every tile, step k + 1's again: 2 x
            //      (K^2 / 2) tile round trips through HBM per pair of steps.  Fused: launch A = phase 1 of step k (as ever); launch
            //      B (bit 2) = step k's update on block COLUMN k + 1 only (K - 1 tiles: what step k + 1's pivot block and panel
            //      read); launch C = phase 1 of step k + 1 with its Y' in the second set (bit 4); launch D (bit 3) = ONE read-
            //      modify-write of every other tile with both products, Y_i Y_j^T then Y'_i Y'_j^T -- subtracted one after the
            //      other, so every tile sees the operations of the two separate passes in their order: the same bits.  Rows are
            //      walked in the index that skips k + 1 (= the slot index of the second set; the first set has the same index for
            //      every row but k, whose tiles -- Y W since launch A, the pivot tile -W^T W -- take the second product only). ----
            const int kn = k + 1;
            if (phases & 4) {
                if (part == 0)
                    *(V4<float>*)(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK + tid * 4) = *(const V4<float>*)(Yg + (size_t)(K - 1) * LQP_BLK + tid * 4);
                // Y_{k+1} (first-set slot k) in LDS slot 0; the other rows of the first set in chunks of seven behind it, dealt
                // alternately to the two workgroups; the diagonal tile (k + 1, k + 1) with part 0's first chunk
                *(V4<float>*)(Y + (size_t)r * SPD_LS + cq * 4) = *(const V4<float>*)(Yg + (size_t)k * LQP_BLK + tid * 4);
                int mine[SPD_BIGK], nm = 0;
                for (int sl = 0, t = 0; sl < K - 1; ++sl) {
                    if (sl == k) continue;
                    if ((t++ & 1) == part) mine[nm++] = sl;
                }
                bool diag_done = part != 0;
                for (int c0 = 0; c0 < nm || !diag_done; c0 += 6) {
                    const int cn = (nm - c0) < 6 ? (nm - c0 > 0 ? nm - c0 : 0) : 6;
                    for (int u = 0; u < cn; ++u)
                        *(V4<float>*)(Y + ((size_t)(1 + u) * 64 + r) * SPD_LS + cq * 4) = *(const V4<float>*)(Yg + (size_t)mine[c0 + u] * LQP_BLK + tid * 4);
                    __syncthreads();
                    const bool with_diag = !diag_done;
                    const int ntask = (cn + (with_diag ? 1 : 0)) * 4;
                    for (int task = __builtin_amdgcn_readfirstlane(w); task < ntask; task += LQP_NW) {
                        const int u = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                        const bool is_diag = u == cn;                    // (the extra task group: tile (k + 1, k + 1))
                        if (is_diag && qi == 0 && qj == 1) continue;   // mirrored from its (1, 0) quadrant
                        const int sl = is_diag ? k : mine[c0 + u];
                        const int i = sl < k ? sl : sl + 1;              // block row of the other operand (i != k)
                        // tile (max, min): X = the panel block of the larger row, Z = that of the smaller one (as the update phase)
                        const float* Xb = is_diag ? Y : (i > kn ? Y + (size_t)(1 + u) * 64 * SPD_LS : Y);
                        const float* Zb = is_diag ? Y : (i > kn ? Y : Y + (size_t)(1 + u) * 64 * SPD_LS);
                        float* T0 = Hs + (size_t)(is_diag ? sym_idx(kn, kn, K) : (i > kn ? sym_idx(i, kn, K) : sym_idx(kn, i, K))) * LQP_BLK;
                        float* C = T0 + (32 * qi) * 64 + 32 * qj + li;
                        f32x16 cur;
#pragma unroll
                        for (int q = 0; q < 16; ++q) cur[q] = C[quad_row(q, lh) * 64];
                        if (phases & 32) {
                            // (two-half operands: the blocks in LDS are cell images, a scale per block and 32-row half)
                            const float* ysc = Yg + (size_t)2 * K * LQP_BLK;
                            const int slx = (is_diag || i < kn) ? k : sl, slz = (is_diag || i > kn) ? k : sl;
                            const float un = -1.f / (ysc[2 * slx + qi] * ysc[2 * slz + qj]);
                            const f32x16 acc = f2_quadrant<0, 4>((const char*)(Xb + (size_t)(32 * qi + li) * SPD_LS) + 32 * lh,
                                                                 (const char*)(Zb + (size_t)(32 * qj + li) * SPD_LS) + 32 * lh);
#pragma unroll
                            for (int q = 0; q < 16; ++q) cur[q] = fmaf(acc[q], un, cur[q]);
                        } else {
                            const f32x16 acc = spd_quadrant(Xb + (size_t)(32 * qi) * SPD_LS, Zb + (size_t)(32 * qj) * SPD_LS);
                            cur -= acc;
                        }
#pragma unroll
                        for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                        if (is_diag && qi == 1 && qj == 0) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) T0[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                        }
                    }
                    diag_done = true;
                    __syncthreads();
                }
                __threadfence_block();
            }
            if (phases & 8) {
                const float* YgB = Yg + (size_t)K * LQP_BLK;
                if (part == 0)
                    *(V4<float>*)(Hs + (size_t)sym_idx(kn, kn, K) * LQP_BLK + tid * 4) = *(const V4<float>*)(YgB + (size_t)(K - 1) * LQP_BLK + tid * 4);
                constexpr int G2 = 2;                                     // rows per group: (second | first set) x (a | b) = 8 LDS slots
                const int nrow = K - 1, ng = (nrow + G2 - 1) / G2;
                // (group a stays in LDS for all of this workgroup's pairs of one ga: staged afresh for every pair, the panel blocks moved
                //  as many bytes as the tiles -- 288 block loads against 272 tile round trips per pair of steps at K = 16; 168 like this)
                for (int ga = 0; ga < ng; ++ga) {
                    const int a0 = G2 * ga;
                    const int an = (nrow - a0) < G2 ? (nrow - a0) : G2;
                    for (int u = 0; u < an; ++u) {
                        *(V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4) = *(const V4<float>*)(YgB + (size_t)(a0 + u) * LQP_BLK + tid * 4);
                        if (a0 + u != k)
                            *(V4<float>*)(Y + ((size_t)(G2 + u) * 64 + r) * SPD_LS + cq * 4) = *(const V4<float>*)(Yg + (size_t)(a0 + u) * LQP_BLK + tid * 4);
                    }
                    for (int gb = 0; gb <= ga; ++gb) {
                        if (((ga * (ga + 1) / 2 + gb) & 1) != part) continue;      // (the pairs dealt alternately in their running number)
                        const int b0 = G2 * gb;
                        const int bn = (nrow - b0) < G2 ? (nrow - b0) : G2;
                        const int boff = gb == ga ? 0 : 2 * G2;            // LDS slots: a second set 0.., a first set G2.., b: + 2 G2
                        if (gb != ga)
                            for (int u = 0; u < bn; ++u) {
                                *(V4<float>*)(Y + ((size_t)(2 * G2 + u) * 64 + r) * SPD_LS + cq * 4) = *(const V4<float>*)(YgB + (size_t)(b0 + u) * LQP_BLK + tid * 4);
                                if (b0 + u != k)
                                    *(V4<float>*)(Y + ((size_t)(3 * G2 + u) * 64 + r) * SPD_LS + cq * 4) = *(const V4<float>*)(Yg + (size_t)(b0 + u) * LQP_BLK + tid * 4);
                            }
                        __syncthreads();
                        const int npair = gb == ga ? an * (an + 1) / 2 : an * bn;
                        auto decode = [&](const int task, int& ua, int& ub, bool& skip, bool& mirror, bool& first) -> float* {
                            const int qi = (task >> 1) & 1, qj = task & 1, p = task >> 2;
                            if (gb == ga) {
                                ua = 0;
                                while ((ua + 1) * (ua + 2) / 2 <= p) ++ua;
                                ub = p - ua * (ua + 1) / 2;
                            } else {
                                ua = p / bn;
                                ub = p - ua * bn;
                            }
                            const int si = a0 + ua, sj = b0 + ub;                  // si >= sj
                            skip = si == sj && qi == 0 && qj == 1;                  // diagonal tile: mirrored from its (1,0) quadrant
                            mirror = si == sj && qi == 1 && qj == 0;
                            first = si != k && sj != k;                              // the first set's product applies
                            const int i = si < kn ? si : si + 1, j = sj < kn ? sj : sj + 1;
                            return Hs + (size_t)sym_idx(i, j, K) * LQP_BLK;
                        };
                        auto load_c = [&](const float* T0, const int task, f32x16& c) {
                            const float* C = T0 + (32 * ((task >> 1) & 1)) * 64 + 32 * (task & 1) + li;
#pragma unroll
                            for (int q = 0; q < 16; ++q) c[q] = C[quad_row(q, lh) * 64];
                        };
                        const int ntask = npair * 4;
                        int task = __builtin_amdgcn_readfirstlane(w);
                        int ua = 0, ub = 0; bool skip = false, mirror = false, first = false;
                        float* T0 = nullptr;
                        f32x16 nxt;
                        if (task < ntask) { T0 = decode(task, ua, ub, skip, mirror, first); if (!skip) load_c(T0, task, nxt); }
                        while (task < ntask) {
                            const int qi = (task >> 1) & 1, qj = task & 1;
                            f32x16 cur = nxt;
                            float* Tc = T0;
                            const int cua = ua, cub = ub;
                            const bool cskip = skip, cmirror = mirror, cfirst = first;
                            const int nt = task + LQP_NW;
                            if (nt < ntask) { T0 = decode(nt, ua, ub, skip, mirror, first); if (!skip) load_c(T0, nt, nxt); }
                            if (!cskip && (phases & 32)) {
                                const float* ysc = Yg + (size_t)2 * K * LQP_BLK;          // [set][slot][half]
                                const int sa = a0 + cua, sb = b0 + cub;
                                if (cfirst) {
                                    const float un = -1.f / (ysc[2 * sa + qi] * ysc[2 * sb + qj]);
                                    const f32x16 acc1 = f2_quadrant<0, 4>((const char*)(Y + ((size_t)(G2 + cua) * 64 + 32 * qi + li) * SPD_LS) + 32 * lh,
                                                                          (const char*)(Y + ((size_t)(boff + G2 + cub) * 64 + 32 * qj + li) * SPD_LS) + 32 * lh);
#pragma unroll
                                    for (int q = 0; q < 16; ++q) cur[q] = fmaf(acc1[q], un, cur[q]);
                                }
                                const float un2 = -1.f / (ysc[2 * K + 2 * sa + qi] * ysc[2 * K + 2 * sb + qj]);
                                const f32x16 acc2 = f2_quadrant<0, 4>((const char*)(Y + ((size_t)cua * 64 + 32 * qi + li) * SPD_LS) + 32 * lh,
                                                                      (const char*)(Y + ((size_t)(boff + cub) * 64 + 32 * qj + li) * SPD_LS) + 32 * lh);
#pragma unroll
                                for (int q = 0; q < 16; ++q) cur[q] = fmaf(acc2[q], un2, cur[q]);
                            } else if (!cskip) {
                                if (cfirst) {
                                    const f32x16 acc1 = spd_quadrant(Y + ((size_t)(G2 + cua) * 64 + 32 * qi) * SPD_LS,
                                                                     Y + ((size_t)(boff + G2 + cub) * 64 + 32 * qj) * SPD_LS);
                                    cur -= acc1;
                                }
                                const f32x16 acc2 = spd_quadrant(Y + ((size_t)cua * 64 + 32 * qi) * SPD_LS,
                                                                 Y + ((size_t)(boff + cub) * 64 + 32 * qj) * SPD_LS);
                                cur -= acc2;
                            }
                            if (!cskip) {
                                float* C = Tc + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                                for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                                if (cmirror) {
#pragma unroll
                                    for (int q = 0; q < 16; ++q) Tc[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                                }
                            }
                            task = nt;
                        }
                        __syncthreads();
                    }
                    __syncthreads();                                        // (group a is restaged: nobody may still read it)
                }
                __threadfence_block();
            }
        }
    }
    if (tid == 0 && flag[0] != 0 && *info == 0) *info = flag[0];
}

// ---------------------------------------------------------------------------
// y = Hs w from the packed lower blocks.  Thread t owns elements [4t, 4t+4) of every block (row t>>4, columns
// 4 (t&15) ..), as in the triangular stream.  Blocks arrive column by column (j, then i = j .. K-1):
//   product 1  y_i[r] += sum_c B[r][c] w_j[c]   row sum: 16 adjacent lanes (DPP), written to the block's own slot
//              ylds[s][r] (s = stream index of the block)
//   product 2  y_j[c] += sum_r B[r][c] w_i[r]   kept in 4 registers over the whole block column, reduced over the
//              4 rows of the wave at the end of the column and written to this wave's own slice part[w][.]
// so no barrier is needed inside the product.  The caller combines (sym_combine) after a barrier.  Blocks [0, LQP_RREG) live in registers, the next `rl` in LDS, the rest stream through the ring
// (which wraps into the next call: virtual length padded to a multiple of LQP_PF).
// ---------------------------------------------------------------------------
template <int NT> struct SymWalk {
    int i, j, s;
    Frag<float, NT> wj;    // w_j slice of this thread's columns
    float acc2[LQP_BLK / NT];
};

// thread t of an NT-thread workgroup owns EPT = 4096 / NT consecutive elements of a block: row t / LPR,
// columns EPT * (t % LPR) ..  (NT = 1024: 4 elements, 16 lanes per row; NT = 512: 8 elements, 8 lanes per row)
// (j0, s0: the walk starts at block column j0, whose first block has stream index s0 -- a workgroup that holds a RANGE of whole
//  block columns, wg_sym_gemv below)
template <int NT>
__device__ __forceinline__ void sym_begin(SymWalk<NT>& wk, const float* __restrict__ v, const int j0 = 0, const int s0 = 0) {
    constexpr int EPT = LQP_BLK / NT, LPR = LQP_NB / EPT;
    wk.i = j0; wk.j = j0; wk.s = s0;
    const float* p = v + j0 * 64 + (threadIdx.x % LPR) * EPT;
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) wk.wj.q[q] = *(const V4<float>*)(p + 4 * q);
#pragma unroll
    for (int e = 0; e < EPT; ++e) wk.acc2[e] = 0.f;
}

template <int NT>
__device__ __forceinline__ void sym_block(SymWalk<NT>& wk, const Frag<float, NT>& blk, const int K, const int Np,
                                          const float* __restrict__ v, float* __restrict__ ylds,
                                          float* __restrict__ part) {
    constexpr int EPT = LQP_BLK / NT, LPR = LQP_NB / EPT;
    const int tid = threadIdx.x, r = tid / LPR, cq = tid % LPR, lane = tid & 63, w = tid >> 6;
    wk.i = __builtin_amdgcn_readfirstlane(wk.i);     // (the walk is uniform: keep it in SGPRs, scalar branches)
    wk.j = __builtin_amdgcn_readfirstlane(wk.j);
    wk.s = __builtin_amdgcn_readfirstlane(wk.s);
    float wi = v[wk.i * 64 + r];                     // requested first: its LDS latency hides under the row sum
    asm volatile("" : "+v"(wi));                      // (keeps the read from being sunk into the i != j branch)
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) d += dot4(blk.q[q], wk.wj.q[q]);
    const float s1 = rowgroup_sum<NT>(d);
    ylds[wk.s * 64 + r] = s1;                        // one slot per block, write-only; all 16 lanes of the row group hold
                                                     // s1 and store it (same address): no predicated branch in the block
    wk.s = __builtin_amdgcn_readfirstlane(wk.s + 1);     // (the walk is uniform: keep it in SGPRs, scalar branches)
    if (wk.i != wk.j) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) wk.acc2[e] += blk.q[e >> 2].v[e & 3] * wi;
    }
    wk.i = __builtin_amdgcn_readfirstlane(wk.i + 1);
    if (wk.i == K) {
        // end of block column j: fold the rows this wave holds, publish its 64 column sums
#if LQP_LANE_SWAP
        // (lanes l, l ^ LPR, ... hold the same columns of other rows: folded by lane swaps on pairs of elements, no LDS)
        col_fold<EPT>(wk.acc2);
        if (LPR == 16 || (lane & 8) == 0) {
            float* dst = part + (size_t)w * Np + wk.j * 64 + cq * EPT + col_fold_elem(lane);
#pragma unroll
            for (int k = 0; k < EPT / 4; ++k) dst[4 * k] = wk.acc2[k];
        }
#else
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            float a = wk.acc2[e];
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)         // lanes l, l ^ off hold the same columns of other rows
                a += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ off) << 2, __builtin_bit_cast(int, a)));
            wk.acc2[e] = a;
        }
        if (lane < LPR) {
            float* dst = part + (size_t)w * Np + wk.j * 64 + cq * EPT;
#pragma unroll
            for (int q = 0; q < EPT / 4; ++q) {
                V4<float> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.v[e] = wk.acc2[4 * q + e];
                *(V4<float>*)(dst + 4 * q) = o;
            }
        }
#endif
#pragma unroll
        for (int e = 0; e < EPT; ++e) wk.acc2[e] = 0.f;
        wk.j = __builtin_amdgcn_readfirstlane(wk.j + 1);
        wk.i = wk.j;
        if (wk.j < K) {
            const float* p = v + wk.j * 64 + cq * EPT;
#pragma unroll
            for (int q = 0; q < EPT / 4; ++q) wk.wj.q[q] = *(const V4<float>*)(p + 4 * q);
        }
    }
}

// register-resident head of the symmetric stream
typedef ResidentRegs<float, LQP_NT> SymResident;
template <int NT>
__device__ __forceinline__ void sym_resident_load(ResidentRegs<float, NT>& rr, float* __restrict__ lds_res,
                                                  const float* __restrict__ Hs, const int S, const int rl) {
    constexpr int NR = resident_regs<NT>();
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (i < S) rr.r[i] = frag_load<float, NT>(Hs + (size_t)i * LQP_BLK);
    for (int i = 0; i < rl; ++i)
        frag_store<float, NT>(lds_res + (size_t)i * LQP_BLK, frag_load<float, NT>(Hs + (size_t)(NR + i) * LQP_BLK));
}
// prime the ring with the first streamed blocks
template <int NT>
__device__ __forceinline__ void sym_prime(BlockStream<float, NT>& st, const float* __restrict__ Hs,
                                          const int first, const int S) {
#pragma unroll
    for (int i = 0; i < LQP_PF; ++i)                        // (slots past the end get block `first` again: never used)
        st.buf[i] = frag_load<float, NT>(Hs + (size_t)(first + i < S ? first + i : (first < S ? first : 0)) * LQP_BLK);
}

// y[e] for e = 64 i + r: the row sums of blocks (i, j <= i) plus the column partials of the NW waves
template <int NT = LQP_NT>
__device__ __forceinline__ float sym_combine(const int e, const int K, const int Np, const float* __restrict__ ylds,
                                             const float* __restrict__ part) {
    const int i = e >> 6, r = e & 63;
    float y = 0.f;
    for (int j = 0; j <= i; ++j) y += ylds[sym_idx(i, j, K) * 64 + r];
#pragma unroll
    for (int ww = 0; ww < NT / 64; ++ww) y += part[(size_t)ww * Np + e];
    return y;
}

// RES: resident head (registers + rl LDS blocks) and a ring that already holds the first streamed blocks and is
// refilled cyclically; !RES: everything streamed, ring primed here, not cyclic.
// Sn >= 0: a RANGE of the stream -- the Sn blocks of the whole block columns j0 .. (stream indices s0 .. s0 + Sn - 1; `Hs`, the
// resident head and the ring are the caller's, relative to block s0): the partial product of those blocks alone -- the row sums
// in the blocks' own ylds slots, the column sums of columns j0 .. in part[w][.]; what belongs to other columns is left alone
// (two workgroups per matrix above 512 rows: k_admm_loop_np2).
template <bool RES, int NT = LQP_NT>
__device__ __forceinline__ void wg_sym_gemv(BlockStream<float, NT>& st, const ResidentRegs<float, NT>& rr,
                                            const float* __restrict__ lds_res, const int rl,
                                            const float* __restrict__ Hs, const int K, const int Np,
                                            const float* __restrict__ v, float* __restrict__ ylds,
                                            float* __restrict__ part, const int j0 = 0, const int s0 = 0, const int Sn = -1) {
    constexpr int NR = resident_regs<NT>();
    const int S = Sn < 0 ? sym_blocks(K) : Sn;
    SymWalk<NT> wk;
    sym_begin<NT>(wk, v, j0, s0);
    int R0 = 0;
    if constexpr (RES) {
#pragma unroll
        for (int s = 0; s < NR; ++s)
            if (s < S) sym_block<NT>(wk, rr.r[s], K, Np, v, ylds, part);
        for (int s = 0; s < rl; ++s) {
            const Frag<float, NT> blk = frag_load<float, NT>(lds_res + (size_t)s * LQP_BLK);
            sym_block<NT>(wk, blk, K, Np, v, ylds, part);
        }
        R0 = (S < NR ? S : NR) + rl;
    } else {
        sym_prime<NT>(st, Hs, 0, S);
    }
    const int Sr = S - R0;                                  // streamed blocks
    const int Sv = round_up(Sr, LQP_PF);                    // virtual ring length
    for (int s0 = 0; s0 < Sv; s0 += LQP_PF) {
#pragma unroll
        for (int i = 0; i < LQP_PF; ++i) {
            const int s = s0 + i;
            if constexpr (RES) {
                // EVERY step takes its slot and refills it (padding steps re-load block R0 and drop it): with
                // conditional loads the compiler cannot count what is in flight and drains the ring
                // (s_waitcnt vmcnt(0)) before each block; like this it waits for the oldest load only.
                const Frag<float, NT> blk = st.buf[i];
                int nx = s + LQP_PF;
                if (nx >= Sv) nx -= Sv;                     // wraps into the next product
                st.buf[i] = frag_load<float, NT>(Hs + (size_t)(R0 + (nx < Sr ? nx : 0)) * LQP_BLK);
                if (s < Sr) sym_block<NT>(wk, blk, K, Np, v, ylds, part);
            } else if (s < Sr) {
                const Frag<float, NT> blk = st.buf[i];
                const int nx = s + LQP_PF;
                if (nx < Sr) st.buf[i] = frag_load<float, NT>(Hs + (size_t)nx * LQP_BLK);
                sym_block<NT>(wk, blk, K, Np, v, ylds, part);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same product shared by TWO workgroups (small batches: 2 B <= #CUs, so the second half of the chip is free).
// Block columns are dealt out in pairs (j, K-1-j) -- together K+1 blocks -- alternately to the two workgroups:
// K = 8: {0,2,5,7} and {1,3,4,6}, 18 blocks each.  A workgroup keeps ALL its blocks on chip for the whole
// launch (SPLIT_RR in registers, the rest in LDS: nothing is streamed inside the loop), computes the partial
// product of its blocks with the same walk as above (row sums to the block's own ylds slot, column sums to
// part[w][.]; slots / columns of the other workgroup stay zero) and the two partial vectors are exchanged through
// global memory by the caller (k_admm_loop_split).
// ---------------------------------------------------------------------------
// register-resident blocks per workgroup: 1024 threads (128 VGPRs each, 4 per block) keep 12 and the rest in LDS;
// 512 threads (256 VGPRs each, 8 per block) keep all 18
template <int NT> __host__ __device__ constexpr int split_rr() { return NT == 512 ? 18 : 12; }
constexpr int SPLIT_MINK = 3;       // (below that -- n <= 128 -- k_admm_loop_small holds the whole matrix in the registers of 256 threads)
// NP workgroups per matrix (2, or 4 when 4 B <= #CUs: batches up to 64): column pair p = min(j, K-1-j) belongs to workgroup p % NP
__host__ __device__ constexpr int split_owner(int j, int K, int NP = 2) { return (j < K - 1 - j ? j : K - 1 - j) & (NP - 1); }
__host__ __device__ constexpr int split_count(int K, int part, int NP = 2) {
    int c = 0;
    for (int j = 0; j < K; ++j) if (split_owner(j, K, NP) == part) c += K - j;
    return c;
}
template <int NT, int NP = 2> __host__ __device__ constexpr int split_lds_blocks(int K) {
    int mx = 0;
    for (int q = 0; q < NP; ++q) { const int a = split_count(K, q, NP); mx = a > mx ? a : mx; }
    const int c = mx - split_rr<NT>();
    return c > 0 ? c : 0;
}

#ifndef LQP_SPLIT_SWAP
#define LQP_SPLIT_SWAP 1
#endif
template <int NT> struct SplitResident {
    Frag<float, NT> r[split_rr<NT>()];
};

// compile-time map of workgroup PART's share of a K-block matrix: local block l <-> (i, j), columns ascending
template <int K, int PART, int NP = 2> struct SplitMap {
    static constexpr int count() { return split_count(K, PART, NP); }
    static constexpr int col_of(int l) {
        for (int j = 0; j < K; ++j) {
            if (split_owner(j, K, NP) != PART) continue;
            if (l < K - j) return j;
            l -= K - j;
        }
        return 0;
    }
    static constexpr int row_of(int l) {
        for (int j = 0; j < K; ++j) {
            if (split_owner(j, K, NP) != PART) continue;
            if (l < K - j) return j + l;
            l -= K - j;
        }
        return 0;
    }
    // local index of block (i, j), j owned
    static constexpr int local_of(int i, int j) {
        int l = 0;
        for (int c = 0; c < j; ++c) if (split_owner(c, K, NP) == PART) l += K - c;
        return l + (i - j);
    }
};

template <int K, int PART, int NT, int NP = 2>
__device__ __forceinline__ void split_resident_load(SplitResident<NT>& rr, float* __restrict__ lds_res, const float* __restrict__ Hs) {
    typedef SplitMap<K, PART, NP> M;
    constexpr int nloc = M::count(), RR = split_rr<NT>();
#pragma unroll
    for (int l = 0; l < RR; ++l)
        if (l < nloc) rr.r[l] = frag_load<float, NT>(Hs + (size_t)sym_idx(M::row_of(l), M::col_of(l), K) * LQP_BLK);
#pragma unroll
    for (int l = RR; l < nloc; ++l)
        frag_store<float, NT>(lds_res + (size_t)(l - RR) * LQP_BLK,
                              frag_load<float, NT>(Hs + (size_t)sym_idx(M::row_of(l), M::col_of(l), K) * LQP_BLK));
}

template <int K, int PART, int NT, int NP = 2>
__device__ __forceinline__ void split_resident_store(const SplitResident<NT>& rr, const float* __restrict__ lds_res, float* __restrict__ Hs) {
    typedef SplitMap<K, PART, NP> M;
    constexpr int nloc = M::count(), RR = split_rr<NT>();
#pragma unroll
    for (int l = 0; l < RR; ++l)
        if (l < nloc) frag_store<float, NT>(Hs + (size_t)sym_idx(M::row_of(l), M::col_of(l), K) * LQP_BLK, rr.r[l]);
#pragma unroll
    for (int l = RR; l < nloc; ++l)
        frag_store<float, NT>(Hs + (size_t)sym_idx(M::row_of(l), M::col_of(l), K) * LQP_BLK,
                              frag_load<float, NT>(lds_res + (size_t)(l - RR) * LQP_BLK));
}

// B_ij += sum_q T_q[64 i + r] G_q[64 j + c] on every block workgroup PART holds (registers and LDS): the rank-m
// equality correction H + T G^T.  Tl, Gl: [m][Np] in LDS.
template <int K, int PART, int NT, int NP = 2>
__device__ __forceinline__ void split_eq_update(SplitResident<NT>& rr, float* __restrict__ lds_res, const float* __restrict__ Gl,
                                                const float* __restrict__ Tl, const int m, const int Np) {
    typedef SplitMap<K, PART, NP> M;
    constexpr int nloc = M::count(), RR = split_rr<NT>(), EPT = LQP_BLK / NT, LPR = LQP_NB / EPT, NV = EPT / 4;
    const int tid = threadIdx.x, r = tid / LPR, c0 = (tid % LPR) * EPT;
#pragma unroll
    for (int l = 0; l < nloc; ++l) {
        const int bi = M::row_of(l), bj = M::col_of(l);
        Frag<float, NT> blk;
        if (l < RR) blk = rr.r[l < RR ? l : 0];
        else blk = frag_load<float, NT>(lds_res + (size_t)(l - RR) * LQP_BLK);
        for (int q = 0; q < m; ++q) {
            const float t = Tl[(size_t)q * Np + bi * 64 + r];
#pragma unroll
            for (int v4 = 0; v4 < NV; ++v4) {
                const V4<float> g = *(const V4<float>*)(Gl + (size_t)q * Np + bj * 64 + c0 + 4 * v4);
#pragma unroll
                for (int e = 0; e < 4; ++e) blk.q[v4].v[e] += t * g.v[e];
            }
        }
        if (l < RR) rr.r[l < RR ? l : 0] = blk;
        else frag_store<float, NT>(lds_res + (size_t)(l - RR) * LQP_BLK, blk);
    }
}

// Partial product of workgroup PART, fully static: every block from registers / LDS at compile-time positions.
// Thread t owns EPT = 4096 / NT consecutive elements of a block: row t / LPR, columns EPT (t % LPR) ..
//   row product   d_i += B_ij . w_j  accumulated per block ROW over this workgroup's columns, ONE reduction over the
//                 LPR lanes of a row per block row, result in yrow[64 i + r] (rows that get no contribution are
//                 never written: stay zero)
//   column product a_j += B_ij^T w_i accumulated over the block column, folded over the rows the wave holds and
//                 written to this wave's slice part[w][64 j ..]
// No walk state, no branches: the compiler interleaves the independent chains.
template <int K, int PART, int NT, int NP = 2>
__device__ __forceinline__ void wg_sym_gemv_split(const SplitResident<NT>& rr, const float* __restrict__ lds_res, const int Np,
                                                  const float* __restrict__ v, float* __restrict__ yrow,
                                                  float* __restrict__ part) {
    typedef SplitMap<K, PART, NP> M;
    constexpr int EPT = LQP_BLK / NT, LPR = LQP_NB / EPT, NV = EPT / 4, RR = split_rr<NT>();
    const int tid = threadIdx.x, r = tid / LPR, cq = tid % LPR, lane = tid & 63, w = tid >> 6;
    // two-wide arithmetic (v_pk_fma_f32: two FMAs per instruction and lane): the product is bound by VALU issue
    typedef float f2 __attribute__((ext_vector_type(2)));
    float wrow[K];
    f2 d[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { wrow[i] = v[i * 64 + r]; d[i] = f2{0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (split_owner(j, K, NP) != PART) continue;
        f2 wj[2 * NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const V4<float> t = *(const V4<float>*)(v + j * 64 + cq * EPT + 4 * q);
            wj[2 * q] = f2{t.v[0], t.v[1]};
            wj[2 * q + 1] = f2{t.v[2], t.v[3]};
        }
        f2 a2[EPT / 2];
#pragma unroll
        for (int e = 0; e < EPT / 2; ++e) a2[e] = f2{0.f, 0.f};
#pragma unroll
        for (int i = j; i < K; ++i) {
            const int l = M::local_of(i, j);
            Frag<float, NT> blk;
            if (l < RR) blk = rr.r[l < RR ? l : 0];
            else blk = frag_load<float, NT>(lds_res + (size_t)(l - RR) * LQP_BLK);
            const f2 wi = f2{wrow[i], wrow[i]};
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const f2 b01 = f2{blk.q[q].v[0], blk.q[q].v[1]}, b23 = f2{blk.q[q].v[2], blk.q[q].v[3]};
                d[i] = __builtin_elementwise_fma(b01, wj[2 * q], d[i]);
                d[i] = __builtin_elementwise_fma(b23, wj[2 * q + 1], d[i]);
                if (i != j) {
                    a2[2 * q] = __builtin_elementwise_fma(b01, wi, a2[2 * q]);
                    a2[2 * q + 1] = __builtin_elementwise_fma(b23, wi, a2[2 * q + 1]);
                }
            }
        }
#if LQP_SPLIT_SWAP
        // The fold over the rows a wave holds (lane bits 3 | 4 | 5, in this order: the same sums as ever) WITHOUT the LDS:
        // lane ^ 8 by DPP; lane ^ 16 and lane ^ 32 by v_permlane16_swap / v_permlane32_swap on PAIRS of elements -- the swap
        // hands each half of the lanes the partner's value of ONE of the two, so one add folds both and the number of
        // live values halves with every step (8 -> 4 -> 2): 8 + 4 + 2 adds and 6 swaps where 16 ds_bpermute round trips
        // (and their 24 adds) stood in the product's dependency chains.  Afterwards lane L holds, in t[k], element
        // 4 k + 2 (L >> 5) + ((L >> 4) & 1) of its column group (EPT = 8; EPT = 4: one value, element 2 (L >> 5) + ((L >> 4) & 1)).
        float a1[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) a1[e] = a2[e >> 1][e & 1];
        col_fold<EPT>(a1);
        if (LPR == 16 || (lane & 8) == 0) {
            float* dst = part + (size_t)w * Np + j * 64 + cq * EPT + col_fold_elem(lane);
#pragma unroll
            for (int k = 0; k < EPT / 4; ++k) dst[4 * k] = a1[k];
        }
#else
        float a1[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            float a = a2[e >> 1][e & 1];
            if constexpr (LPR == 8) a += dpp<0x128>(a);          // row_ror:8 = lane ^ 8 inside a 16-lane DPP row
            a += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __builtin_bit_cast(int, a)));
            a += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, a)));
            a1[e] = a;
        }
        if (lane < LPR) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                V4<float> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.v[e] = a1[4 * q + e];
                *(V4<float>*)(part + (size_t)w * Np + j * 64 + cq * EPT + 4 * q) = o;
            }
        }
#endif
    }
    constexpr int first_row = M::col_of(0);            // the lowest owned column: rows above it get nothing
#pragma unroll
    for (int i = first_row; i < K; ++i) yrow[i * 64 + r] = rowgroup_sum<NT>(d[i][0] + d[i][1]);
}

// y[e] of this workgroup's partial: the block-row sum plus the column partials of the waves
template <int NT>
__device__ __forceinline__ float split_combine(const int e, const int Np, const float* __restrict__ yrow,
                                               const float* __restrict__ part) {
    float y = yrow[e];
#pragma unroll
    for (int ww = 0; ww < NT / 64; ++ww) y += part[(size_t)ww * Np + e];
    return y;
}

// ---------------------------------------------------------------------------
// Register-resident sweep (small batches: 2 B <= #CUs): the same block symmetric sweep with the matrix held ON CHIP
// for all K pivot steps.  Two workgroups of 512 threads share a matrix; each keeps its half of the tiles (the
// column pairs of the two-workgroup loop, 18 tiles at K = 8) in registers in the MFMA accumulator layout -- tile =
// four 32x32 quadrants, quadrant q of the workgroup's list sits in slot q / 8 of wave q % 8: 9 slots x 16 = 144 of
// the 256 VGPRs a 512-thread workgroup has.  Per pivot step only the pivot tile and the 7 panel tiles travel:
//   publish   owners store them (write-through, sc1) to an exchange buffer (two parities: a workgroup is at most one
//             step ahead of its partner), every wave drains its stores, one lane raises the workgroup's step flag;
//   acquire   one lane polls the partner's flag, ONE agent-scope acquire drops the CU's stale L1 lines, barrier;
//   stage     pivot tile -> W = L^-1, W^T (wg_pivot_block, both workgroups for themselves), panel tiles -> LDS,
//             Y_i = P_i W^T in place;
//   update    every resident quadrant by its kind: A_ij -= Y_i Y_j^T | A_ik = Y_i W | A_ki = W^T Y_i^T | A_kk = -W^T W.
// The multi-launch form (k_spd_begin/step/end) moved all 36 tiles through L2/HBM in every step (1.24 GB per batch of
// 128 at n = 500, 13x the minimum); this one reads the matrix once and writes it once.
// ---------------------------------------------------------------------------
constexpr int RS_NT = 512, RS_NW = RS_NT / 64;
// xb: exchange buffer of this matrix, [2][K][4096] floats; fl: step flags of the two workgroups; epoch: added to the
// step numbers (a refactorisation in the same forward must not match the flags of the first one)
// rho not known when the blocks were built (FwdParams::rho_late): the two halves of ||Qs||_F^2 wait behind the exchange
// buffer (k_spd_begin left them there), rho = clamp(||Qs||_F / sqrt(n)) as the setup kernel would have computed it
// (reference :200-203) is added to the diagonal in registers and stored for the loop.
struct RsLateRho {
    int on, n;
    float rho_min, rho_max;
    float* rho_out;           // workgroup 0 only
    // the blocks are UNSCALED (k_spd_prep built them before the scaling was known):
    const float* dsc;         // the scaling D (n values): a tile entry is taken as (D_row * v) * D_col, like sym_scale4
    int fro_self;             // 1: rho = clamp(||Qs||_F / sqrt(n)) with the norm summed from the tiles themselves (the two
                              //    workgroups swap their halves with the step-0 flags); 0 (with dsc): rho_given is added
    float rho_given;
    int xcd_local;            // 1: workgroups that find themselves on ONE XCD exchange through its L2 (workgroup-scope stores)
    // the sweep makes the pass over the UNSCALED matrix itself (FwdParams::prep_fused == 3):
    const float* q;           // Q of this problem (n x n, row-major); nullptr: off
    float* cmx;               // global scratch of this problem: [NP][64 K + 2] words (column maxima | asymmetry | magnitude)
    unsigned long long* qdbg; // optional: 8 cycle stamps of the pass (tools/gpu_resident_phases.py), workgroup 0 of the matrix
    int dbg_tid;              // the thread whose phase stamps go to dbg (LQP_DBG_QPASS bits 8..: its wave)
};
// (the scaling vector from the column maxima: supplied by the kernel, which knows the problem's parameters)
//  scaling(red, d, work): every thread of the workgroup; red: n column maxima, d: n values out, work: 8 + RS_NW floats (LDS)
//  deferred(d): ONE wave, while the pivot block of step 0 is eliminated -- what the setup kernel left undone for want of d
struct RsNoScaling {
    __device__ __forceinline__ void scaling(float*, float*, float*) const {}
    __device__ __forceinline__ void deferred(const float*) const {}
};

// ---- the same resident sweep with the pivot block on the matrix cores --------------------------------------------
// wg_pivot_block_mfma keeps 32 accumulator registers per working wave next to the tiles; with 9 x 16 tile registers in
// every wave the compiler spilled 245 of them.  Here the tiles are dealt out unevenly: waves 0..3 (the pivot block's four
// working waves) hold NA tiles each, waves 4..7 NB = nloc - NA (8 and 10 of 18 at K = 8).  Wave w and wave w + 4 share a
// SIMD, so every SIMD still carries 18 quadrant updates per step: the update phase stays MFMA-bound.  While waves 0..3 run
// the pivot block, waves 4..7 stage the panel (it used to be staged by everybody afterwards).  Same exchange protocol,
// same arithmetic per tile as the first resident sweep (round 2; its code is in the repository's history).
template <int K, int NP = 2> __host__ __device__ constexpr int rs2_max() {
    int mx = 0;
    for (int q = 0; q < NP; ++q) { const int a = split_count(K, q, NP); mx = a > mx ? a : mx; }
    return mx;
}
template <int K, int NP = 2> __host__ __device__ constexpr int rs2_na() { return (rs2_max<K, NP>() * 4) / 9; }
template <int K, int NP = 2> __host__ __device__ constexpr int rs2_nb() { return rs2_max<K, NP>() - rs2_na<K, NP>(); }
// (i, j) of local tile l of workgroup `part` of NP (columns ascending, as SplitMap)
__device__ __forceinline__ void rs2_tile_of(int l, const int K, const int part, const int NP, int& ti, int& tj) {
    ti = 0; tj = 0;
    for (int j = 0; j < K; ++j) {
        if (split_owner(j, K, NP) != part) continue;
        if (l < K - j) { ti = j + l; tj = j; return; }
        l -= K - j;
    }
}
// NP = 4 (batches up to a quarter of the CUs, K >= 7): four workgroups share a matrix, one column pair each (9 tiles at
// K = 8: 4 + 5 per wave pair); every one of them still eliminates the pivot block and computes Y for itself -- what is
// divided is the tile updates.  Step flags: one 64-bit granule per workgroup, a workgroup waits for all the others.
// F16: the panel products (Y = P W^T and every tile update) on the float16 matrix pipe with two-half operands
// (lqp_f16x2.hpp): W, W^T and the Y panel live in LDS as split images of the size of their float32 rows, each 32-row
// block in a scale of its own.  The pivot block, the exchange and the staging are the float32 ones.
template <int K, int NP = 2, class DFn = RsNoScaling, bool F16 = false>
__device__ __forceinline__ void wg_spd_sweep_resident_v2(const float* Hsrc, float* Hdst,
                                                         float* __restrict__ xb, unsigned int* __restrict__ fl,
                                                         const unsigned int epoch, const int part, int* __restrict__ info,
                                                         int* __restrict__ status_timeout, char* smem,
                                                         const RsLateRho lr = RsLateRho{0, 0, 0.f, 0.f, nullptr, nullptr, 0, 0.f, 0, nullptr, nullptr, nullptr, 0},
                                                         unsigned long long* __restrict__ dbg = nullptr,
                                                         const DFn hooks = DFn{}) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qi = (w >> 1) & 1, qj = w & 1;                  // quadrant of the tile this wave holds in every slot
    const int nloc = split_count(K, part, NP);
    float* Y = (float*)smem;
    float* W = Y + (size_t)(K - 1) * 64 * SPD_LS;
    float* WT = W + 64 * SPD_LS;
    float* pcol = WT + 64 * SPD_LS;
    int* flag = (int*)(pcol + PIV_LDS);
    float* const dkeep = (float*)(flag + 20);               // [64 K] the scaling vector of the sweep's own pass over Q (rs_q_lds_bytes)
    if (tid == 0) { flag[0] = 0; flag[2] = 0; }
    // unscaled blocks (k_spd_prep): the scaling vector, 1 on the padding, in LDS while the tiles are loaded (the Y area
    // is not written before the staging of step 0, two barriers away)
    float* const Dl = Y;
    static_assert(64 * K <= RS_NT, "one element of the scaling vector per thread");
    // (requested first, staged behind the tile loads: its latency then hides under theirs)
    const float dmine = (lr.dsc && tid < lr.n) ? lr.dsc[tid] : 1.f;
    // step flags: one 64-bit granule per workgroup {step number, payload} -- at step 0 the payload is the workgroup's
    // half of ||Qs||_F^2 (fro_self), so the norm costs no hand-off of its own
    unsigned long long* const fl64 = (unsigned long long*)fl;
    // Which XCD are the workgroups of this matrix on?  Each announces its id now (write-through store) and reads the others'
    // after its tile loads.  On ONE XCD its L2 is their point of coherence: tiles and step granules are then stored with
    // workgroup scope -- they stay in that L2 instead of being written through to memory and fetched back from there
    // (131 MB per sweep of the batch) -- and read as before behind the acquire.  Never assumed: asked at every launch.
    const unsigned int xcd_me = (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
    // (with the pass over Q inside: announced BEHIND the column maxima, whose arrival it then signals as well)
    if (tid == 0 && !lr.q) __hip_atomic_store(fl64 + 4 + part, ((unsigned long long)(epoch + 1u) << 8) | xcd_me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // thread 0: wait for the announcements of the other workgroups of this matrix -> are all of them on this XCD?
    auto partners_on_my_xcd = [&](const bool wait_for_all) -> int {
        int same = 1;
        for (int q = 0; q < NP && (same || wait_for_all); ++q) {
            if (q == part) continue;
            unsigned long long g = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (((g = __hip_atomic_load(fl64 + 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 8) != (unsigned long long)(epoch + 1u)) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ULL) {      // (1 s: give up, results are flagged)
                    g = ~0ull;
                    if (wait_for_all) __hip_atomic_store(status_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            same = same && (unsigned int)(g & 0xFFull) == xcd_me;
        }
        return same;
    };

    auto body = [&](auto pivot_tag) {
        constexpr bool PIVOT = decltype(pivot_tag)::value;     // waves 0..3
        // (F16 with the tiles dealt evenly, 9 / 9 at K = 8: measured, no difference -- 0.280 ms either way)
        constexpr int NS = PIVOT ? rs2_na<K, NP>() : rs2_nb<K, NP>(), FIRST = PIVOT ? 0 : rs2_na<K, NP>();
        // ---- tiles into registers (accumulator layout: register q of lane l = element (quad_row(q, l>>5), l & 31)) ----
        f32x16 T[NS];
        int ti[NS], tj[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int l = FIRST + s;
            int a, b;
            rs2_tile_of(l < nloc ? l : 0, K, part, NP, a, b);
            ti[s] = __builtin_amdgcn_readfirstlane(l < nloc ? a : -1);
            tj[s] = __builtin_amdgcn_readfirstlane(l < nloc ? b : -1);
            if (ti[s] >= 0 && !lr.q) {
                const float* blk = Hsrc + (size_t)sym_idx(ti[s], tj[s], K) * LQP_BLK;
                if (ti[s] == tj[s] && ti[s] > 0 && qi == 0 && qj == 1) {
                    // upper-right quadrant of a diagonal tile := transpose of its lower-left one (as in the block sweep)
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = blk[(32 + li) * 64 + quad_row(q, lh)];
                } else {
                    const float* C = blk + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = C[quad_row(q, lh) * 64];
                }
            }
        }
        if (lr.q) {
            // ---- the pass over Q inside the sweep: this wave's quadrants straight into their registers, and with them the
            //      mirror quadrants (Q is read once per solve, by the workgroups that keep it) for the column maxima of |Q|
            //      (what the auto-scaling starts from, reference :163) and the symmetry verdict of wg_sym_prep.  The
            //      workgroups of the matrix swap their maxima through global scratch; every one of them then forms the
            //      scaling vector for itself (make_d: the same bits as k_fwd_setup).  LDS: the Y area, free until step 0. ----
            const int n = lr.n;
            const float* __restrict__ Qm = lr.q;
            const unsigned long long qt0 = (dbg || lr.qdbg) ? clock64() : 0ull;
            unsigned int* cm = (unsigned int*)(Y + 64 * K);       // [64 K] column maxima as bit patterns (non-negative floats order like integers)
            float* redm = Y + 2 * 64 * K;                          // [512] the combined maxima, then the sort buffer of make_d (n padded to a power of two)
            float* redw = redm + 512;                              // [2 RS_NW] asymmetry / magnitude per wave
            float* work = redw + 2 * RS_NW;                        // [8 + RS_NW] make_d
            float* trw = work + 8 + RS_NW + 8 + w * (32 * 33);     // this wave's 32 x 33 transposition tile
            for (int i = tid; i < 64 * K; i += RS_NT) cm[i] = 0u;
            // Quadrant (a, b) of tile (i, j) in the accumulator layout.  Requests only: the addresses are clamped to the matrix
            // and what lies outside is zeroed where the values are first used (mask_quadrant) -- a select behind every load
            // made the compiler wait for each quadrant before it asked for the next: 20 memory latencies, 112 us.
            auto load_quadrant = [&](f32x16& dst, const int i, const int j, const int a, const int b) {
                const int r0 = i * 64 + 32 * a, col = j * 64 + 32 * b + li;
                const float* __restrict__ base = Qm + (col < n ? col : n - 1);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = r0 + quad_row(q, lh);
                    dst[q] = base[(row < n ? row : n - 1) * n];
                }
            };
            auto mask_quadrant = [&](f32x16& dst, const int i, const int j, const int a, const int b) {
                if (i * 64 + 32 * a + 32 <= n && j * 64 + 32 * b + 32 <= n) return;       // (uniform: inside)
                const int r0 = i * 64 + 32 * a, col = j * 64 + 32 * b + li;
#pragma unroll
                for (int q = 0; q < 16; ++q) dst[q] = (r0 + quad_row(q, lh) < n && col < n) ? dst[q] : 0.f;
            };
            constexpr int MD = 2;                        // mirrors requested ahead (measured, group ms / spilled registers: 2: 0.358 / 31, 3: 0.360 / 63, 4: 0.378 / 125)
            f32x16 M[MD];
#pragma unroll
            for (int s = 0; s < MD && s < NS; ++s)
                if (ti[s] >= 0) load_quadrant(M[s], tj[s], ti[s], qj, qi);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // (the upper-right quadrant of a diagonal tile > 0 is held as the transpose of its lower-left one: it comes out
                //  of the mirror below)
                if (ti[s] >= 0 && !(ti[s] == tj[s] && ti[s] > 0 && qi == 0 && qj == 1)) load_quadrant(T[s], ti[s], tj[s], qi, qj);
            }
            wg_barrier_lds();                                       // (cm is zero; the requests stay in flight)
            float dmax = 0.f, vmax = 0.f;
            auto col_max = [&](const f32x16& v, const int col0) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) a = tmax(a, tabs(v[q]));
                a = tmax(a, xor32(a));
                if (lh == 0) atomicMax(cm + col0 + li, __float_as_uint(a));
            };
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (ti[s] < 0) continue;
                const bool held_as_mirror = ti[s] == tj[s] && ti[s] > 0 && qi == 0 && qj == 1;
                f32x16 Mc = M[s % MD];
                if (s + MD < NS && ti[s + MD] >= 0) load_quadrant(M[s % MD], tj[s + MD], ti[s + MD], qj, qi);
                mask_quadrant(Mc, tj[s], ti[s], qj, qi);
                col_max(Mc, ti[s] * 64 + 32 * qi);
                // transpose the mirror through LDS: Mt[q] = mirror(li, quad_row(q, lh))
                f32x16 Mt;
#pragma unroll
                for (int q = 0; q < 16; ++q) trw[quad_row(q, lh) * 33 + li] = Mc[q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16; ++q) Mt[q] = trw[li * 33 + quad_row(q, lh)];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (held_as_mirror) {
                    T[s] = Mt;
                } else {
                    mask_quadrant(T[s], ti[s], tj[s], qi, qj);
                    col_max(T[s], tj[s] * 64 + 32 * qj);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        dmax = tmax(dmax, tabs(T[s][q] - Mt[q]));
                        vmax = tmax(vmax, tmax(tabs(T[s][q]), tabs(Mt[q])));
                    }
                }
                // identity on the padding
                if (ti[s] == tj[s] && qi == qj) {
                    int dqp = (ti[s] * 64 + 32 * qi + li >= n) ? li - 4 * lh : -1;
                    asm volatile("" : "+v"(dqp));
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = (dqp == (q & 3) + 8 * (q >> 2)) ? 1.f : T[s][q];
                }
            }
            if (dbg && tid == 0) dbg[7] = clock64() - qt0;         // (tiles + mirrors)
            if (lr.qdbg && tid == 0) lr.qdbg[0] = clock64() - qt0;
            dmax = wave_max(dmax);
            vmax = wave_max(vmax);
            if (lane == 0) { redw[w] = dmax; redw[RS_NW + w] = vmax; }
            __syncthreads();                                        // (cm, redw complete)
            unsigned int* out = (unsigned int*)lr.cmx + (size_t)part * (64 * K + 2);
            for (int i = tid; i < 64 * K; i += RS_NT) __hip_atomic_store(out + i, cm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                float a = 0.f, v = 0.f;
                for (int ww = 0; ww < RS_NW; ++ww) { a = tmax(a, redw[ww]); v = tmax(v, redw[RS_NW + ww]); }
                redw[0] = a; redw[RS_NW] = v;
                __hip_atomic_store(out + 64 * K, __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(out + 64 * K + 1, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                        // (every store of this workgroup has been acknowledged)
            if (lr.qdbg && tid == 0) lr.qdbg[1] = clock64() - qt0;
            if (tid == 0) {
                __hip_atomic_store(fl64 + 4 + part, ((unsigned long long)(epoch + 1u) << 8) | xcd_me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int same = partners_on_my_xcd(true);
                flag[1] = same && lr.xcd_local;
            }
            __syncthreads();
            if (lr.qdbg && tid == 0) lr.qdbg[2] = clock64() - qt0;
            // (the partners' words: agent-scope loads, past this CU's L1)
            for (int i = tid; i < 64 * K; i += RS_NT) {
                unsigned int v = cm[i];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (q == part) continue;
                    const unsigned int o = __hip_atomic_load((const unsigned int*)lr.cmx + (size_t)q * (64 * K + 2) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = o > v ? o : v;
                }
                redm[i] = __uint_as_float(v);
            }
            if (tid == 0) {
                float a = redw[0], v = redw[RS_NW];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (q == part) continue;
                    const unsigned int* o = (const unsigned int*)lr.cmx + (size_t)q * (64 * K + 2) + 64 * K;
                    a = tmax(a, __uint_as_float(__hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                    v = tmax(v, __uint_as_float(__hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                }
                flag[2] = a > 1e-5f * v ? 1 : 0;                   // not symmetric (the rule of wg_sym_prep, over the whole matrix)
            }
            __syncthreads();
            if (lr.qdbg && tid == 0) lr.qdbg[3] = clock64() - qt0;
            hooks.scaling(redm, Dl, work);                          // Dl[0 .. n): the scaling vector (every thread returns behind a barrier)
            if (lr.qdbg && tid == 0) lr.qdbg[4] = clock64() - qt0;
            for (int i = n + tid; i < 64 * K; i += RS_NT) Dl[i] = 1.f;
            for (int i = tid; i < n; i += RS_NT) dkeep[i] = Dl[i];  // (the Y area goes to the panel of step 0)
            if (dbg && tid == 0) dbg[6] = clock64() - qt0;         // (... + exchange + scaling vector + deferred vectors)
            if (flag[2]) {
                if (tid == 0 && part == 0) *info = K * 64 + 2;      // the LU path takes it (status word: k_spd_end / the loop kernel)
                return;
            }
        }
        if (lr.dsc || lr.q) {
            // unscaled blocks (k_spd_prep): entry (r, c) is taken as (D_r * v) * D_c, what sym_scale4 computes
            if (lr.dsc) Dl[tid] = dmine;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (ti[s] < 0) continue;
                if (ti[s] == tj[s] && ti[s] > 0 && qi == 0 && qj == 1) {      // (as the lower-left entry it mirrors: row 32 + li, column quad_row)
                    const float dr = Dl[ti[s] * 64 + 32 + li];
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = (dr * T[s][q]) * Dl[tj[s] * 64 + quad_row(q, lh)];
                } else {
                    const float dc = Dl[tj[s] * 64 + 32 * qj + li];
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = (Dl[ti[s] * 64 + 32 * qi + quad_row(q, lh)] * T[s][q]) * dc;
                }
            }
        }
        if ((lr.dsc || lr.q) && lr.fro_self) {
            // this wave's share of ||Qs||_F^2: tiles below the diagonal count twice, a diagonal tile's four quadrants once
            // each (its upper-right one is held as the mirror of the lower-left one); the identity on the padding is left out
            // (lane-dependent compares against an opaque value, formed where they are used: as invariants of the step loop
            //  the sixteen diagonal masks would be held in scalar registers -- and spilled -- for the whole kernel)
            float fs = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (ti[s] < 0) continue;
                float t2 = 0.f;
                if (ti[s] == tj[s] && qi == qj) {
                    int dqp = (ti[s] * 64 + 32 * qi + li >= lr.n) ? li - 4 * lh : -1;      // register q is a padding-diagonal entry
                    asm volatile("" : "+v"(dqp));
#pragma unroll
                    for (int q = 0; q < 16; ++q) t2 += (dqp == (q & 3) + 8 * (q >> 2)) ? 0.f : T[s][q] * T[s][q];
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) t2 += T[s][q] * T[s][q];
                }
                fs += ti[s] == tj[s] ? t2 : 2.f * t2;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) fs += __shfl_xor(fs, o);
            if (lane == 0) WT[w] = fs;                     // (summed by thread 0 behind the first barrier of step 0)
        }
        if (lr.on) {
            const float* xw = xb + (size_t)2 * K * LQP_BLK;
            float rho = sqrtf(xw[0] + xw[1]) / (float)sqrt((double)lr.n);
            rho = tmin(tmax(rho, lr.rho_min), lr.rho_max);
            if (tid == 0 && lr.rho_out) *lr.rho_out = rho;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (ti[s] >= 0 && ti[s] == tj[s] && qi == qj) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (quad_row(q, lh) == li && ti[s] * 64 + 32 * qi + li < lr.n) T[s][q] += rho;
                }
            }
        }
        unsigned long long dbt[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dstage = 0;
        // publish the pivot tile and the panel tiles of step kk this wave holds (as soon as ITS quadrants have step kk-1's
        // update: the store drain then overlaps with the wait for the slowest wave)
        bool xlocal_p = false;          // (set before the first publish)
        auto publish = [&](const int kk) {
            float* xbp = xb + (size_t)(kk & 1) * K * LQP_BLK;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = ti[s], j = tj[s];
                if (i >= 0 && (i == kk || j == kk)) {
                    const int slot = (i == kk && j == kk) ? K - 1 : (j == kk ? i - 1 : j);      // P_i: i > kk -> i - 1, i < kk -> i
                    unsigned int* dst = (unsigned int*)(xbp + (size_t)slot * LQP_BLK + (32 * qi) * 64 + 32 * qj + li);
                    if constexpr (F16) {
                        // max |.| of this quadrant next to the tiles: the staging waves of every workgroup scale the panel by it
                        if (slot != K - 1) {
                            float mq = 0.f;
#pragma unroll
                            for (int q = 0; q < 16; ++q) mq = tmax(mq, tabs(T[s][q]));
                            mq = wave_max(mq);
                            unsigned int* mdst = (unsigned int*)(xb + (size_t)2 * K * LQP_BLK + 64 + (kk & 1) * 4 * K + slot * 4 + 2 * qi + qj);
                            if (lane == 0) {
                                if (xlocal_p) __hip_atomic_store(mdst, __builtin_bit_cast(unsigned int, mq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                else __hip_atomic_store(mdst, __builtin_bit_cast(unsigned int, mq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                    if (xlocal_p) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float tv = T[s][q];
                            __hip_atomic_store(dst + quad_row(q, lh) * 64, __builtin_bit_cast(unsigned int, tv),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float tv = T[s][q];
                            __hip_atomic_store(dst + quad_row(q, lh) * 64, __builtin_bit_cast(unsigned int, tv),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        };
        if (tid == 0 && !lr.q) flag[1] = lr.xcd_local ? partners_on_my_xcd(false) : 0;
        __syncthreads();
        const bool xlocal = flag[1] != 0;
        xlocal_p = xlocal;
        publish(0);
        for (int k = 0; k < K; ++k) {
            if (dbg) dt0 = clock64();
            float* xbk = xb + (size_t)(k & 1) * K * LQP_BLK;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                       // (also closes step k-1: every wave is done with the LDS panel)
            if (tid == 0) {
                float fmine = 0.f;
                if (k == 0 && (lr.dsc || lr.q) && lr.fro_self)      // this workgroup's half of the norm travels in the step-0 granule
                    for (int ww = 0; ww < RS_NW; ++ww) fmine += WT[ww];
                const unsigned long long gran = (unsigned long long)(epoch + (unsigned int)k + 1u) | ((unsigned long long)__float_as_uint(fmine) << 32);
                if (xlocal) __hip_atomic_store(fl64 + part, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(fl64 + part, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                float fparts[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    fparts[q] = fmine;
                    if (q == part) continue;
                    unsigned long long got;
                    while ((unsigned int)(got = __hip_atomic_load(fl64 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < epoch + (unsigned int)k + 1u) {
                        __builtin_amdgcn_s_sleep(2);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ULL) {      // 1 s: give up, results are flagged
                            __hip_atomic_store(status_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    fparts[q] = __uint_as_float((unsigned int)(got >> 32));
                }
                // (no acquire fence: every load of the partner's tiles bypasses this CU's L1 -- ld16_handoff -- and the polls
                //  above have matched before anybody loads, the barrier below in between)
                if (k == 0 && (lr.dsc || lr.q) && lr.fro_self) {
                    // rho = clamp(||Qs||_F / sqrt(n)) (reference :200-203), the same bits in both workgroups
                    float fsum = fparts[0];             // (summed in part order: the same bits in every workgroup)
#pragma unroll
                    for (int q = 1; q < NP; ++q) fsum += fparts[q];
                    float rho = sqrtf(fsum) / (float)sqrt((double)lr.n);
                    rho = tmin(tmax(rho, lr.rho_min), lr.rho_max);
                    WT[RS_NW] = rho;
                    if (lr.rho_out) *lr.rho_out = rho;
                }
            }
            __syncthreads();
            float diag_add = 0.f;
            if (k == 0 && (lr.dsc || lr.q)) {
                // rho on the diagonal: of the tiles in registers, and (diag_add) of the pivot tile as it is staged -- tile
                // (0, 0) was published without it
                diag_add = lr.fro_self ? WT[RS_NW] : lr.rho_given;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (ti[s] >= 0 && ti[s] == tj[s] && qi == qj) {
                        int dq = (ti[s] * 64 + 32 * qi + li < lr.n) ? li - 4 * lh : -1;      // (opaque: see the norm above)
                        asm volatile("" : "+v"(dq));
#pragma unroll
                        for (int q = 0; q < 16; ++q) T[s][q] += (dq == (q & 3) + 8 * (q >> 2)) ? diag_add : 0.f;
                    }
                }
            }
            if (dbg) { const unsigned long long t = clock64(); dbt[0] += t - dt0; dt0 = t; }
            // ---- pivot tile -> W, W^T by waves 0..3 | panel tiles -> LDS by waves 4..7 (slot s holds P_i = A_ik, i.e.
            //      block (k, i) transposed when i < k) ----
            wg_pivot_block_mfma<false, PIVOT, false, F16>(xbk + (size_t)(K - 1) * LQP_BLK, W, WT, pcol, flag, k * 64, nullptr, 0, diag_add, true,
                                                          pcol + 128);
            // (lane parts of every LDS / global address of this step, opaque: as loop invariants of the step loop they
            //  would be formed once, held in registers -- one per distinct address -- and spilled with the tiles)
            int li_s = li, lh_s = lh, tid_s = tid;
            asm volatile("" : "+v"(li_s), "+v"(lh_s), "+v"(tid_s));
            if constexpr (!PIVOT) {
                const int tt = tid_s - 256, r0 = tt >> 3, c8 = (tt & 7) * 8;
                const float* const src_l = xbk + r0 * 64 + c8;
                float* const yrow_l = Y + r0 * SPD_LS + c8;            // row-major destination
                float* const ycol_l = Y + c8 * SPD_LS + r0;            // transposed destination
                if constexpr (F16) {
                    // ---- straight into the split image of the panel, in the scale of each tile
                    //      (the maxima of all tiles requested up front: one trip to the L2 instead of one per tile) ----
                    constexpr int NIT = 2 * (K - 1);      // (one half-tile in flight per staging wave; four: measured slower, 0.31 against 0.28 ms -- registers)
                    const unsigned long long* const mq = (const unsigned long long*)(xb + (size_t)2 * K * LQP_BLK + 64 + (k & 1) * 4 * K);
                    unsigned long long mraw[2 * (K - 1)];
#pragma unroll
                    for (int s0 = 0; s0 < K - 1; ++s0) {
                        mraw[2 * s0] = __hip_atomic_load(mq + 2 * s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        mraw[2 * s0 + 1] = __hip_atomic_load(mq + 2 * s0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // A thread takes eight consecutive columns t0 .. t0 + 7 of ONE row of the panel slot: of a tile that is stored as the
                    // panel wants it (slot >= k: block (i, k)) two 16-byte loads of its row; of a tile stored transposed (slot < k:
                    // block (k, i)) eight 4-byte loads down its column -- lane = row of the panel, so every load instruction still
                    // reads 256 contiguous bytes.  (The transposed tiles used to be loaded by rows and scattered into the image with
                    // 2-byte LDS stores, sixteen per thread and half-tile, 4-way bank conflicts: LDS time the pivot chain's queue
                    // traffic waited behind.)
                    float rv[8];
                    auto request = [&](const int it, float (&v)[8]) {
                        const int s0 = it >> 1, hf = it & 1;
                        if (s0 >= k) {
                            const float* src = src_l + s0 * LQP_BLK + 32 * hf * 64;
                            const V4<float> a = ld16_handoff(src), b = ld16_handoff(src + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] = a.v[e]; v[4 + e] = b.v[e]; }
                        } else {
                            const unsigned int* src = (const unsigned int*)(xbk + s0 * LQP_BLK + (8 * (tt >> 6) + 32 * hf) * 64 + (tt & 63));
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                v[e] = __uint_as_float(__hip_atomic_load(src + e * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        }
                    };
                    request(0, rv);
                    // the tiles' scales from the maxima their owners published with them (four quadrants; both 32-row blocks of a
                    // slot share it)
                    float sPt[K - 1];
#pragma unroll
                    for (int s0 = 0; s0 < K - 1; ++s0) {
                        const unsigned long long m01 = mraw[2 * s0], m23 = mraw[2 * s0 + 1];
                        const float mt = tmax(tmax(__uint_as_float((unsigned int)m01), __uint_as_float((unsigned int)(m01 >> 32))),
                                              tmax(__uint_as_float((unsigned int)m23), __uint_as_float((unsigned int)(m23 >> 32))));
                        float isPt;
                        f2_scale_of(mt, sPt[s0], isPt);
                        if (tt == s0) (pcol + 160)[s0] = isPt;
                    }
                    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                    char* const Yc_ = (char*)Y;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int s0 = it >> 1, hf = it & 1;
                        float va[4], vb[4];
                        h16x4 ha, ma, hb, mb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            va[e] = rv[e] * sPt[s0]; vb[e] = rv[4 + e] * sPt[s0];
                            ha[e] = (_Float16)va[e]; ma[e] = (_Float16)(va[e] - (float)ha[e]);
                            hb[e] = (_Float16)vb[e]; mb[e] = (_Float16)(vb[e] - (float)hb[e]);
                        }
                        if (it + 1 < NIT) request(it + 1, rv);
                        // row `prow` of the slot, columns t0 .. t0 + 7: slice t0 / 16, element half (t0 / 8) & 1 of the cells of both
                        // lane halves (t0 .. t0 + 3: h = 0, t0 + 4 .. t0 + 7: h = 1; lqp_f16x2.hpp)
                        const int prow = s0 >= k ? r0 + 32 * hf : (tt & 63);
                        const int t0 = s0 >= k ? c8 : 8 * (tt >> 6) + 32 * hf;
                        char* d = Yc_ + (s0 * 64 + prow) * F2_ROW + 64 * (t0 >> 4) + 8 * ((t0 >> 3) & 1);
                        *(h16x4*)d = ha; *(h16x4*)(d + 16) = ma;
                        *(h16x4*)(d + 32) = hb; *(h16x4*)(d + 48) = mb;
                    }
                } else {
#pragma unroll
                for (int s0 = 0; s0 < K - 1; ++s0) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const float* src = src_l + s0 * LQP_BLK + 32 * hf * 64;
                        const V4<float> a = ld16_handoff(src), b = ld16_handoff(src + 4);
                        if (s0 >= k) {
                            *(V4<float>*)(yrow_l + (s0 * 64 + 32 * hf) * SPD_LS) = a;
                            *(V4<float>*)(yrow_l + (s0 * 64 + 32 * hf) * SPD_LS + 4) = b;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                ycol_l[(s0 * 64 + e) * SPD_LS + 32 * hf] = a.v[e];
                                ycol_l[(s0 * 64 + 4 + e) * SPD_LS + 32 * hf] = b.v[e];
                            }
                        }
                    }
                }
                }
                if (dbg) dstage += clock64() - dt0;                      // (a staging wave's own work of this phase)
                // (these waves now wait ~20 k cycles for the pivot block: room for the vectors the setup kernel left)
                if (k == 0 && lr.q && w == RS_NW - 1) hooks.deferred(dkeep);
            }
            __syncthreads();
            if (dbg) { const unsigned long long t = clock64(); dbt[1] += t - dt0; dt0 = t; }
            if constexpr (F16) {
            // ================= two-half operands on the float16 matrix pipe =================
            float* const f2w = pcol + 128;                              // [RS_NW] wave maxima of |W|
            float* const ysc = pcol + 144;                              // [2 (K-1)] 1 / scale of the 32-row blocks of Y
            char* const Yc = (char*)Y;
            char* const Wc = (char*)W;
            char* const WTc = (char*)WT;
            const int lane_b = li_s * F2_ROW + 32 * lh_s;               // the lane's cell of slice 0 of row li of an image
            // ---- W, W^T -> split images in place (thread = one cell of one row of each; lanes 2c, 2c+1 share the bytes of a
            //      row's 16 columns and sit in one wave: every read below precedes every write) ----
            const int crow = tid_s >> 3, ccs = (tid_s >> 1) & 3, cch = tid_s & 1;
            float sW, isW;
            f2_scale_of(tmax(f2w[1], f2w[2]), sW, isW);                 // (max |W| from the two waves that stored it: wg_pivot_block_mfma)
            {
                float vw[8], vt[8];
                f2_load_cell_f32(W + crow * SPD_LS, ccs, cch, vw);
                f2_load_cell_f32(WT + crow * SPD_LS, ccs, cch, vt);
                h16x8 hi, mid;
#pragma unroll
                for (int j = 0; j < 8; ++j) { vw[j] *= sW; vt[j] *= sW; }
                f2_split8(vw, hi, mid);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave's float32 reads are through before any of its lanes writes)
                f2_write_cell(Wc + crow * F2_ROW + 64 * ccs + 32 * cch, hi, mid);
                f2_split8(vt, hi, mid);
                f2_write_cell(WTc + crow * F2_ROW + 64 * ccs + 32 * cch, hi, mid);
            }
            const float* const psc = pcol + 160;                        // [K - 1] 1 / scale of the staged panel slots
            // this lane's four cells of its row of a 32-row block of the staged panel (split by the staging waves)
            auto load_p = [&](const int rb, F2Cell (&pb)[4], float& isP) {
                const char* prow = Yc + ((rb >> 1) * 64 + 32 * (rb & 1)) * F2_ROW + lane_b;
#pragma unroll
                for (int c = 0; c < 4; ++c) pb[c] = f2_read_cell(prow + 64 * c);
                isP = psc[rb >> 1];
            };
            F2Cell pb[4];
            float isP = 1.f;
            const int rb0 = __builtin_amdgcn_readfirstlane(w);
            if (rb0 < 2 * (K - 1)) load_p(rb0, pb, isP);
            __syncthreads();
            if (dbg) { const unsigned long long t = clock64(); dbt[2] += t - dt0; dt0 = t; }
            // ---- Y^T = W P^T: the rows of Y across the lanes, a lane's sixteen entries of an accumulator are the two
            //      cells it stores (register q = column 8 (q >> 2) + 4 lh + (q & 3) of the 32-column half) ----
            // (No pass for the block's scale: both operands are below 2^15 in their scales, so the 64-term sums stay below 2^36 --
            //  taken times 2^-21 they are below 2^15 whatever the data, and a float16 pair has 39 bits of range for float32's 24:
            //  a scale that is a few powers of two too cautious costs nothing.  Y's scale is sP sW 2^-21.)
            auto make_y = [&](const int rb, const F2Cell (&pbl)[4], const float isPl) {
                const char* wa = Wc + lane_b;
                char* dst = Yc + ((rb >> 1) * 64 + 32 * (rb & 1)) * F2_ROW + lane_b;
                if (lane == 0) ysc[rb] = (isW * isPl) * 2097152.f;
                auto store_half = [&](const f32x16& a, const int c0) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = a[8 * c + j] * 4.76837158203125e-7f;
                        h16x8 hi, mid;
                        f2_split8(v, hi, mid);
                        f2_write_cell(dst + 64 * (c0 + c), hi, mid);
                    }
                };
                {   // columns 0..31 (W is lower triangular: its rows < 32 end at column 31 -- slices 0, 1)
                    f32x16 a0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) a0[q] = 0.f;
#pragma unroll
                    for (int c = 0; c < 2; ++c) a0 = f2_mma(f2_read_cell(wa + 64 * c), pbl[c], a0);
                    store_half(a0, 0);
                }
                {
                    f32x16 a1;
#pragma unroll
                    for (int q = 0; q < 16; ++q) a1[q] = 0.f;
                    if constexpr (PIVOT) {
                        // (the pivot block's waves hold two tiles fewer: room for the four cells of W at once -- one LDS round trip
                        //  in front of the twelve matrix instructions instead of four between them)
                        F2Cell wc[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) wc[c] = f2_read_cell(wa + 32 * F2_ROW + 64 * c);
#pragma unroll
                        for (int c = 0; c < 4; ++c) a1 = f2_mma(wc[c], pbl[c], a1);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) a1 = f2_mma(f2_read_cell(wa + 32 * F2_ROW + 64 * c), pbl[c], a1);
                    }
                    store_half(a1, 2);
                }
            };
            if (rb0 < 2 * (K - 1)) make_y(rb0, pb, isP);
            {
                const int rb1 = __builtin_amdgcn_readfirstlane(w + RS_NW);
                if (rb1 < 2 * (K - 1)) {
                    load_p(rb1, pb, isP);
                    make_y(rb1, pb, isP);
                }
            }
            __syncthreads();
            if (dbg) { const unsigned long long t = clock64(); dbt[3] += t - dt0; dt0 = t; }
            // ---- every resident quadrant by its kind ----
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                int i = ti[s], j = tj[s];
                asm volatile("" : "+s"(i), "+s"(j));
                if (i < 0) continue;
                if (i != k && j != k) {
                    const int si = i < k ? i : i - 1, sj = j < k ? j : j - 1;
                    const f32x16 a = f2_quadrant<0, 4>(Yc + (si * 64 + 32 * qi) * F2_ROW + lane_b, Yc + (sj * 64 + 32 * qj) * F2_ROW + lane_b);
                    const float un = -(ysc[2 * si + qi] * ysc[2 * sj + qj]);
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = __builtin_fmaf(a[q], un, T[s][q]);
                } else if (i == k && j == k) {    // -W^T W (W^T is upper triangular: rows >= 32 only see columns >= 32)
                    const char* xa = WTc + (32 * qi) * F2_ROW + lane_b;
                    const char* zb = WTc + (32 * qj) * F2_ROW + lane_b;
                    const f32x16 a = (qi | qj) ? f2_quadrant<2, 4>(xa, zb) : f2_quadrant<0, 4>(xa, zb);
                    const float un = -(isW * isW);
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = a[q] * un;
                } else if (j == k) {              // tile (i, k), i > k: Y_i W
                    const char* xa = Yc + ((i - 1) * 64 + 32 * qi) * F2_ROW + lane_b;
                    const char* zb = WTc + (32 * qj) * F2_ROW + lane_b;
                    const f32x16 a = qj == 1 ? f2_quadrant<2, 4>(xa, zb) : f2_quadrant<0, 4>(xa, zb);
                    const float un = ysc[2 * (i - 1) + qi] * isW;
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = a[q] * un;
                } else {                          // tile (k, j), j < k: W^T Y_j^T
                    const char* xa = WTc + (32 * qi) * F2_ROW + lane_b;
                    const char* zb = Yc + (j * 64 + 32 * qj) * F2_ROW + lane_b;
                    const f32x16 a = qi == 1 ? f2_quadrant<2, 4>(xa, zb) : f2_quadrant<0, 4>(xa, zb);
                    const float un = isW * ysc[2 * j + qj];
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = a[q] * un;
                }
            }
            } else {
            // ---- Y_i = P_i W^T in place: a wave takes whole 32-row blocks (both column halves) ----
            constexpr int WOFF = (K - 1) * 64 * SPD_LS, WTOFF = WOFF + 64 * SPD_LS;      // W, W^T behind the panel
            const float* const yF = Y + li_s * SPD_LS + 32 * lh_s;     // operand row li at the lane's k range (full)
            const float* const yH = Y + li_s * SPD_LS + 16 * lh_s;     // ... (half k range; + 32: the upper half)
            float* const yC = Y + (4 * lh_s) * SPD_LS + li_s;          // element (quad_row(q, lh), li) at + ((q&3) + 8 (q>>2)) * SPD_LS
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int rb = __builtin_amdgcn_readfirstlane(w + RS_NW * u);
                if (rb < 2 * (K - 1)) {
                    const int xoff = ((rb >> 1) * 64 + 32 * (rb & 1)) * SPD_LS;
                    const f32x16 a0 = spd_quadrant_lp<1>(yH + xoff, yH + WOFF);
                    const f32x16 a1 = spd_quadrant_lp<0>(yF + xoff, yF + WOFF + 32 * SPD_LS);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        yC[xoff + ((q & 3) + 8 * (q >> 2)) * SPD_LS] = a0[q];
                        yC[xoff + ((q & 3) + 8 * (q >> 2)) * SPD_LS + 32] = a1[q];
                    }
                }
            }
            __syncthreads();
            if (dbg) { const unsigned long long t = clock64(); dbt[3] += t - dt0; dt0 = t; }
            // ---- every resident quadrant by its kind ----
            const int oi = 32 * qi * SPD_LS, oj = 32 * qj * SPD_LS;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // (opaque per step: what the compiler can derive from a slot's tile indices alone -- lane addresses of its panel
                //  rows -- it forms once before the step loop, one register per slot, and spills them with the tiles)
                int i = ti[s], j = tj[s];
                asm volatile("" : "+s"(i), "+s"(j));
                if (i < 0) continue;
                if (i != k && j != k) {
                    const int si = i < k ? i : i - 1, sj = j < k ? j : j - 1;
                    T[s] -= spd_quadrant_lp<0>(yF + si * 64 * SPD_LS + oi, yF + sj * 64 * SPD_LS + oj);
                } else if (i == k && j == k) {    // (W^T is upper triangular: rows >= 32 only see k >= 32 -- same halves,
                                                  //  hence the same summation order and bits, as the multi-launch sweep)
                    const f32x16 a = (qi | qj) ? spd_quadrant_lp<1>(yH + 32 + WTOFF + oi, yH + 32 + WTOFF + oj)
                                               : spd_quadrant_lp<0>(yF + WTOFF, yF + WTOFF);
#pragma unroll
                    for (int q = 0; q < 16; ++q) T[s][q] = -a[q];
                } else if (j == k) {              // tile (i, k), i > k: Y_i W
                    const int yo = (i - 1) * 64 * SPD_LS + oi;
                    T[s] = qj == 1 ? spd_quadrant_lp<1>(yH + 32 + yo, yH + 32 + WTOFF + 32 * SPD_LS) : spd_quadrant_lp<0>(yF + yo, yF + WTOFF);
                } else {                          // tile (k, j), j < k: W^T Y_j^T
                    const int yo = j * 64 * SPD_LS + oj;
                    T[s] = qi == 1 ? spd_quadrant_lp<1>(yH + 32 + WTOFF + 32 * SPD_LS, yH + 32 + yo) : spd_quadrant_lp<0>(yF + WTOFF, yF + yo);
                }
            }
            }
            if (dbg) { const unsigned long long t = clock64(); dbt[4] += t - dt0; dt0 = t; }
            if (k + 1 < K) publish(k + 1);
            if (dbg) { const unsigned long long t = clock64(); dbt[5] += t - dt0; }
        }
        if (dbg && tid == lr.dbg_tid) {
            for (int q = 0; q < 6; ++q) dbg[q] = dbt[q];
            if (lr.dbg_tid != 0) dbg[6] = dstage;
        }
        // ---- the finished tiles to their home blocks ----
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (ti[s] >= 0) {
                float* C = Hdst + (size_t)sym_idx(ti[s], tj[s], K) * LQP_BLK + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = T[s][q];
            }
        }
    };
    if (w < 4) body(std::true_type());
    else body(std::false_type());
    __syncthreads();
    if (tid == 0 && flag[0] != 0 && *info == 0) *info = flag[0];
}

// (LDS of the resident sweep: the block sweep's + 64 bytes of flags / counters; with the pass over Q inside the sweep also the
//  scaling vector)
__host__ __device__ inline int rs_q_lds_bytes(int K) { return spd_lds_bytes(K) + 64 + 64 * K * 4; }


// (+ 64: the sums of the late rho; + 2 x 4 K: max |.| of every quadrant of the published tiles, by step parity -- the float16-pipe sweep)
__host__ __device__ constexpr size_t rs2_xb_floats(int K) { return (size_t)2 * K * LQP_BLK + 64 + 8 * K; }

// ---------------------------------------------------------------------------
// Blocked Cholesky of an SPD matrix held as packed lower blocks, in place (the symmetric backward system):
//   block (i,k), i > k  <-  L_ik;   block (k,k)  <-  W_k = L_kk^-1   (so the solves below need no substitution
//   inside a block).  Same machinery as the sweep, restricted to the trailing part: pivot block through
//   wg_pivot_block, panel Y_i = A_ik W^T and the tile updates A_ij -= Y_i Y_j^T on MFMA.
// info: 0, or 1 + index of the first non-positive pivot.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wg_chol_factor(float* __restrict__ Hs, const int K, int* __restrict__ info, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = tid >> 4, cq = tid & 15, li = lane & 31, lh = lane >> 5;
    const int nslot = K > 1 ? K - 1 : 1;
    float* Y = (float*)smem;
    float* W = Y + (size_t)nslot * 64 * SPD_LS;
    float* WT = W + 64 * SPD_LS;
    float* pcol = WT + 64 * SPD_LS;
    int* flag = (int*)(pcol + PIV_LDS);
    if (tid == 0) flag[0] = 0;
    for (int k = 0; k < K; ++k) {
        const int np = K - 1 - k;                           // panel blocks below the pivot: slot s <-> row k+1+s
#if LQP_PIV_MFMA
        // (matrix-core pivot block on waves 0..3; the other twelve stage the panel meanwhile)
        wg_pivot<LQP_PIV_WAVES>(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK, W, WT, pcol, flag, k * 64);
        if (w >= 4) {
            for (int v0 = tid - 256; v0 < 1024 * np; v0 += LQP_NT - 256) {
                const int s = v0 >> 10, v = v0 & 1023;
                *(V4<float>*)(Y + ((size_t)s * 64 + (v >> 4)) * SPD_LS + (v & 15) * 4) =
                    *(const V4<float>*)(Hs + (size_t)sym_idx(k + 1 + s, k, K) * LQP_BLK + v * 4);
            }
        }
#else
        V4<float> preg[SPD_MAXK - 1];
#pragma unroll
        for (int s = 0; s < SPD_MAXK - 1; ++s)
            if (s < np) preg[s] = *(const V4<float>*)(Hs + (size_t)sym_idx(k + 1 + s, k, K) * LQP_BLK + tid * 4);
        wg_pivot<LQP_PIV_WAVES>(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK, W, WT, pcol, flag, k * 64);
#pragma unroll
        for (int s = 0; s < SPD_MAXK - 1; ++s)
            if (s < np) *(V4<float>*)(Y + ((size_t)s * 64 + r) * SPD_LS + cq * 4) = preg[s];
#endif
        __syncthreads();
        // the pre-inverted diagonal block
        *(V4<float>*)(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK + tid * 4) = *(const V4<float>*)(W + r * SPD_LS + cq * 4);
        // ---- Y_i = P_i W^T (in place: all products first, then the writes) ----
        {
            const int ntask = np * 4;
            f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                if (task < ntask) {
                    const int s = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                    const float* Xp = Y + ((size_t)s * 64 + 32 * qi) * SPD_LS;
                    acc[u] = qj == 0 ? spd_quadrant<1, false>(Xp, W) : spd_quadrant(Xp, W + 32 * SPD_LS);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                if (task < ntask) {
                    const int s = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                    float* dst = Y + ((size_t)s * 64 + 32 * qi) * SPD_LS + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) dst[quad_row(q, lh) * SPD_LS] = acc[u][q];
                }
            }
            __syncthreads();
        }
        // ---- L_ik = Y_i to its block; trailing tiles A_ij -= Y_i Y_j^T (i >= j > k) ----
        for (int s = 0; s < np; ++s)
            *(V4<float>*)(Hs + (size_t)sym_idx(k + 1 + s, k, K) * LQP_BLK + tid * 4) =
                *(const V4<float>*)(Y + ((size_t)s * 64 + r) * SPD_LS + cq * 4);
        {
            const int nupd = np * (np + 1) / 2 * 4;
            for (int task = __builtin_amdgcn_readfirstlane(w); task < nupd; task += LQP_NW) {
                const int qi = (task >> 1) & 1, qj = task & 1, p = task >> 2;
                int si = 0;
                while ((si + 1) * (si + 2) / 2 <= p) ++si;
                const int sj = p - si * (si + 1) / 2;
                if (si == sj && qi == 0 && qj == 1) continue;      // diagonal tile: mirrored from its (1,0) quadrant
                float* T0 = Hs + (size_t)sym_idx(k + 1 + si, k + 1 + sj, K) * LQP_BLK;
                float* C = T0 + (32 * qi) * 64 + 32 * qj + li;
                f32x16 cur;
#pragma unroll
                for (int q = 0; q < 16; ++q) cur[q] = C[quad_row(q, lh) * 64];
                const f32x16 acc = spd_quadrant(Y + ((size_t)si * 64 + 32 * qi) * SPD_LS,
                                                Y + ((size_t)sj * 64 + 32 * qj) * SPD_LS);
                cur -= acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                if (si == sj && qi == 1 && qj == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) T0[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0 && flag[0] != 0 && *info == 0) *info = flag[0];
}

// ---- the same factorisation for SPD_MAXK < K <= SPD_BIGK (free sets above 512 variables): the panel of a step (up to
// K - 1 blocks) no longer fits the LDS next to W and W^T.  Y_i = L_ik is formed in chunks of SPD_MAXK - 1 blocks and goes
// straight to its final block (i, k) (it stays in L2); the trailing updates A_ij -= L_ik L_jk^T then walk over pairs of
// GROUPS of BIG_G panel blocks staged from there (over W and W^T, dead in that phase), as in wg_spd_sweep_big.
// In place, one workgroup per matrix; LDS as wg_chol_factor at K = SPD_MAXK.
__device__ __forceinline__ void wg_chol_factor_big(float* __restrict__ Hs, const int K, int* __restrict__ info, char* smem) {
    constexpr int CH = SPD_MAXK - 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = tid >> 4, cq = tid & 15, li = lane & 31, lh = lane >> 5;
    float* Y = (float*)smem;
    float* W = Y + (size_t)CH * 64 * SPD_LS;
    float* WT = W + 64 * SPD_LS;
    float* pcol = WT + 64 * SPD_LS;
    int* flag = (int*)(pcol + PIV_LDS);
    if (tid == 0) flag[0] = 0;
    for (int k = 0; k < K; ++k) {
        const int np = K - 1 - k;                           // panel blocks below the pivot: index a <-> row k+1+a
        wg_pivot<LQP_PIV_WAVES>(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK, W, WT, pcol, flag, k * 64);
        __syncthreads();
        // the pre-inverted diagonal block
        *(V4<float>*)(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK + tid * 4) = *(const V4<float>*)(W + r * SPD_LS + cq * 4);
        // ---- Y phase: chunk of the panel -> LDS, Y_i = P_i W^T in place, L_ik = Y_i -> its block ----
        for (int c0 = 0; c0 < np; c0 += CH) {
            const int cn = (np - c0) < CH ? (np - c0) : CH;
            for (int u = 0; u < cn; ++u)
                *(V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4) =
                    *(const V4<float>*)(Hs + (size_t)sym_idx(k + 1 + c0 + u, k, K) * LQP_BLK + tid * 4);
            __syncthreads();
            {
                const int ntask = cn * 4;                   // <= 28: at most two quadrants per wave
                f32x16 acc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                    if (task < ntask) {
                        const int sl = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                        const float* Xp = Y + ((size_t)sl * 64 + 32 * qi) * SPD_LS;
                        acc[u] = qj == 0 ? spd_quadrant<1, false>(Xp, W) : spd_quadrant(Xp, W + 32 * SPD_LS);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int task = __builtin_amdgcn_readfirstlane(w + 16 * u);
                    if (task < ntask) {
                        const int sl = task >> 2, qi = (task >> 1) & 1, qj = task & 1;
                        float* dst = Y + ((size_t)sl * 64 + 32 * qi) * SPD_LS + 32 * qj + li;
#pragma unroll
                        for (int q = 0; q < 16; ++q) dst[quad_row(q, lh) * SPD_LS] = acc[u][q];
                    }
                }
                __syncthreads();
            }
            for (int u = 0; u < cn; ++u)
                *(V4<float>*)(Hs + (size_t)sym_idx(k + 1 + c0 + u, k, K) * LQP_BLK + tid * 4) =
                    *(const V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4);
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        // ---- update phase: A_ij -= L_ik L_jk^T for i >= j > k, by pairs of panel groups ----
        const int ng = (np + BIG_G - 1) / BIG_G;
        for (int ga = 0; ga < ng; ++ga) {
            for (int gb = 0; gb <= ga; ++gb) {
                const int a0 = BIG_G * ga, b0 = BIG_G * gb;
                const int an = (np - a0) < BIG_G ? (np - a0) : BIG_G, bn = (np - b0) < BIG_G ? (np - b0) : BIG_G;
                const int boff = gb == ga ? 0 : BIG_G;
                for (int u = 0; u < an; ++u)
                    *(V4<float>*)(Y + ((size_t)u * 64 + r) * SPD_LS + cq * 4) =
                        *(const V4<float>*)(Hs + (size_t)sym_idx(k + 1 + a0 + u, k, K) * LQP_BLK + tid * 4);
                if (gb != ga)
                    for (int u = 0; u < bn; ++u)
                        *(V4<float>*)(Y + ((size_t)(BIG_G + u) * 64 + r) * SPD_LS + cq * 4) =
                            *(const V4<float>*)(Hs + (size_t)sym_idx(k + 1 + b0 + u, k, K) * LQP_BLK + tid * 4);
                __syncthreads();
                const int npair = gb == ga ? an * (an + 1) / 2 : an * bn;
                auto decode = [&](const int task, int& ua, int& ub, bool& skip, bool& mirror) -> float* {
                    const int qi = (task >> 1) & 1, qj = task & 1, p = task >> 2;
                    if (gb == ga) {
                        ua = 0;
                        while ((ua + 1) * (ua + 2) / 2 <= p) ++ua;
                        ub = p - ua * (ua + 1) / 2;
                    } else {
                        ua = p / bn;
                        ub = p - ua * bn;
                    }
                    const int si = a0 + ua, sj = b0 + ub;                  // si >= sj
                    skip = si == sj && qi == 0 && qj == 1;                  // diagonal tile: mirrored from its (1,0) quadrant
                    mirror = si == sj && qi == 1 && qj == 0;
                    return Hs + (size_t)sym_idx(k + 1 + si, k + 1 + sj, K) * LQP_BLK;
                };
                auto load_c = [&](const float* T0, const int task, f32x16& c) {
                    const float* C = T0 + (32 * ((task >> 1) & 1)) * 64 + 32 * (task & 1) + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) c[q] = C[quad_row(q, lh) * 64];
                };
                const int ntask = npair * 4;
                int task = __builtin_amdgcn_readfirstlane(w);
                int ua = 0, ub = 0; bool skip = false, mirror = false;
                float* T0 = nullptr;
                f32x16 nxt;
                if (task < ntask) { T0 = decode(task, ua, ub, skip, mirror); if (!skip) load_c(T0, task, nxt); }
                while (task < ntask) {
                    const int qi = (task >> 1) & 1, qj = task & 1;
                    f32x16 cur = nxt;
                    float* Tc = T0;
                    const int cua = ua, cub = ub;
                    const bool cskip = skip, cmirror = mirror;
                    const int nt = task + LQP_NW;
                    if (nt < ntask) { T0 = decode(nt, ua, ub, skip, mirror); if (!skip) load_c(T0, nt, nxt); }
                    if (!cskip) {
                        const f32x16 acc = spd_quadrant(Y + ((size_t)cua * 64 + 32 * qi) * SPD_LS,
                                                        Y + ((size_t)(boff + cub) * 64 + 32 * qj) * SPD_LS);
                        cur -= acc;
                        float* C = Tc + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                        for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                        if (cmirror) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) Tc[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                        }
                    }
                    task = nt;
                }
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0 && flag[0] != 0 && *info == 0) *info = flag[0];
}

// ---- the same factorisation with a LOOK-AHEAD pivot chain ---------------------------------------------------
// The 64-column elimination chain of a diagonal block (wg_pivot_block) is latency bound and keeps four waves busy
// for ~15 us while the other twelve wait; the panel product and the tile updates of a step take about as long again.
// Here the workgroup is split by role: waves 0..3 run the chain of block k+1 as soon as tile (k+1,k+1) has received
// step k's update (always the first tile the others touch), waves 4..15 do everything else of step k meanwhile.  The two
// groups meet only through LDS words:
//   pv   (count)  3 quadrant writers per step have put the next diagonal tile into the LDS staging tile St
//   wrd  (step)   W / W^T of step k are in LDS
//   ydn  (step)   the tile waves are done reading W / W^T of step k (they may be overwritten)
// Results are bit-identical to wg_chol_factor: same products, same order inside every tile.
// LDS: the layout of wg_chol_factor for K blocks plus the 64x64 staging tile; K <= SPD_MAXK - 1 keeps it inside
// spd_lds_bytes(SPD_MAXK).
constexpr int CHOL_LA_CHAIN = 4;
__host__ __device__ inline int chol_la_lds_bytes(int K) { return spd_lds_bytes(K) + 64 + 64 * 64 * 4; }

// F16: the tile waves' products (Y = P W^T, the trailing updates) on the float16 matrix pipe with two-half operands
// (lqp_f16x2.hpp), as in the resident sweep.  W stays the chain waves' float32 copy (they overwrite it as soon as the tile
// waves let go): every tile wave builds the six cells of W it multiplies with in its registers; L_ik = Y goes to its block
// in global memory straight from the accumulators, its split image replaces the panel rows in LDS.
template <bool F16 = false>
__device__ __forceinline__ void wg_chol_factor_la(float* __restrict__ Hs, const int K, int* __restrict__ info, char* smem,
                                                  unsigned long long* __restrict__ dbg = nullptr) {
    constexpr int NCH = CHOL_LA_CHAIN, NTW = LQP_NW - NCH, NTT = NTW * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nslot = K > 1 ? K - 1 : 1;
    float* Y = (float*)smem;
    float* W = Y + (size_t)nslot * 64 * SPD_LS;
    float* WT = W + 64 * SPD_LS;
    float* pcol = WT + 64 * SPD_LS;
    int* flag = (int*)(pcol + PIV_LDS);
    int* sy = flag + 4;                                     // [0] chain sync, [1] tile sync, [2] pv, [3] wrd, [4] ydn
    float* St = (float*)(flag + 20);
    if (tid < 20) flag[tid] = 0;
    *(V4<float>*)(St + tid * 4) = *(const V4<float>*)(Hs + (size_t)sym_idx(0, 0, K) * LQP_BLK + tid * 4);
    __syncthreads();
    if (w < NCH) {
        // ================= the chain waves =================
        __builtin_amdgcn_s_setprio(3);
        int gt = 0;
        for (int k = 0; k < K; ++k) {
            unsigned long long c0 = dbg ? clock64() : 0;
            lds_wait_ge(sy + 2, 3 * k);                     // tile (k,k) is in St
            lds_wait_ge(sy + 4, k);                         // W / W^T of step k-1 are no longer read
            if (dbg && tid == 0) { const unsigned long long c1 = clock64(); dbg[4] += c1 - c0; c0 = c1; }
#if LQP_PIV_MFMA
            wg_pivot_block_mfma<true, true, false, F16>(St, W, WT, pcol, flag, k * 64, sy + 5, k, 0.f, false, pcol + 128);
#else
            wg_pivot_block<NCH, true>(St, W, WT, pcol, flag, k * 64, sy + 0, &gt);
#endif
            lds_group_sync<true>(sy + 0, gt += NCH);        // all four wrote their part of W / W^T
            if (tid == 0) __hip_atomic_store(sy + 3, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // the pre-inverted diagonal block (read again only by the solves, after the factorisation's last barrier)
            for (int v = tid; v < 1024; v += NCH * 64)
                *(V4<float>*)(Hs + (size_t)sym_idx(k, k, K) * LQP_BLK + v * 4) = *(const V4<float>*)(W + (v >> 4) * SPD_LS + (v & 15) * 4);
            if (dbg && tid == 0) dbg[5] += clock64() - c0;
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        // ================= the tile waves =================
        const int tw = w - NCH, tt = tid - NCH * 64;
        int gt = 0;
        for (int k = 0; k < K; ++k) {
            const int np = K - 1 - k;                       // panel blocks below the pivot: slot s <-> row k+1+s
            for (int s = 0; s < np; ++s) {
                const float* src = Hs + (size_t)sym_idx(k + 1 + s, k, K) * LQP_BLK;
                for (int v = tt; v < 1024; v += NTT)
                    *(V4<float>*)(Y + ((size_t)s * 64 + (v >> 4)) * SPD_LS + (v & 15) * 4) = *(const V4<float>*)(src + v * 4);
            }
            // this wave's first tile of the step (tile (k+1,k+1) for the first three waves) is fetched while the chain
            // waves still work on W
            const int nupd = np * (np + 1) / 2 * 4;
            f32x16 pre;
            if (tw < nupd && !(tw == 1)) {                  // (task 1 = the mirrored quadrant of a diagonal tile)
                const int qi = (tw >> 1) & 1, qj = tw & 1, p = tw >> 2;
                int si = 0;
                while ((si + 1) * (si + 2) / 2 <= p) ++si;
                const int sj = p - si * (si + 1) / 2;
                if (!(si == sj && qi == 0 && qj == 1)) {
                    const float* C = Hs + (size_t)sym_idx(k + 1 + si, k + 1 + sj, K) * LQP_BLK + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                    for (int q = 0; q < 16; ++q) pre[q] = C[quad_row(q, lh) * 64];
                }
            }
            lds_group_sync(sy + 1, gt += NTW);              // the panel is staged (while the chain waves still work on W)
            unsigned long long c0 = dbg ? clock64() : 0;
            lds_wait_ge(sy + 3, k + 1);                     // W / W^T of this step
            if (dbg && tt == 0) { const unsigned long long c1 = clock64(); dbg[6] += c1 - c0; c0 = c1; }
            // ---- Y_i = P_i W^T, in place: a wave owns 32 rows of a panel block and reads nothing else of Y ----
            [[maybe_unused]] float* const ysc = pcol + 144;           // F16: 1 / scale of the 32-row blocks of Y
            [[maybe_unused]] char* const Yc = (char*)Y;
            [[maybe_unused]] const int lane_b = li * F2_ROW + 32 * lh;
            if constexpr (F16) {
                float sW, isW;
                f2_scale_of(tmax((pcol + 128)[1], (pcol + 128)[2]), sW, isW);      // (max |W| from the two chain waves that stored it)
                for (int task = tw; task < np * 2; task += NTW) {
                    const int s = task >> 1, qi = task & 1;
                    // this lane's row of the block: its four cells, in the block's scale
                    const float* prow = Y + ((size_t)s * 64 + 32 * qi + li) * SPD_LS;
                    float pv[4][8], mx = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f2_load_cell_f32(prow, c, lh, pv[c]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) mx = tmax(mx, tabs(pv[c][j]));
                    }
                    mx = wave_max(mx);
                    float sP, isP;
                    f2_scale_of(mx, sP, isP);
                    F2Cell pb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) pv[c][j] *= sP;
                        f2_split8(pv[c], pb[c].hi, pb[c].mid);
                    }
                    // Y^T = W P^T (the rows of Y across the lanes); W's cells from its float32 rows li (slices 0, 1: lower
                    // triangular) and 32 + li
                    f32x16 a0, a1;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float wv[8];
                        F2Cell wc;
                        if (c < 2) {
                            f2_load_cell_f32(W + li * SPD_LS, c, lh, wv);
#pragma unroll
                            for (int j = 0; j < 8; ++j) wv[j] *= sW;
                            f2_split8(wv, wc.hi, wc.mid);
                            a0 = f2_mma(wc, pb[c], a0);
                        }
                        f2_load_cell_f32(W + (32 + li) * SPD_LS, c, lh, wv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) wv[j] *= sW;
                        f2_split8(wv, wc.hi, wc.mid);
                        a1 = f2_mma(wc, pb[c], a1);
                    }
                    const float un = isW * isP;
                    if (lane == 0) ysc[task] = un * 2097152.f;
                    // L_ik: register q of an accumulator is column 8 (q >> 2) + 4 lh + (q & 3) of its 32-column half, row li
                    float* dstg = Hs + (size_t)sym_idx(k + 1 + s, k, K) * LQP_BLK + (32 * qi + li) * 64 + 4 * lh;
                    char* dstl = Yc + ((size_t)s * 64 + 32 * qi) * F2_ROW + lane_b;
#pragma unroll
                    for (int hsel = 0; hsel < 2; ++hsel) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float v[8];
#pragma unroll
                            for (int gq = 0; gq < 2; ++gq) {
                                V4<float> o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float a = hsel ? a1[8 * c + 4 * gq + e] : a0[8 * c + 4 * gq + e];
                                    o.v[e] = a * un;
                                    v[4 * gq + e] = a * 4.76837158203125e-7f;
                                }
                                *(V4<float>*)(dstg + 32 * hsel + 8 * (2 * c + gq)) = o;
                            }
                            h16x8 hi, mid;
                            f2_split8(v, hi, mid);
                            f2_write_cell(dstl + 64 * (2 * hsel + c), hi, mid);
                        }
                    }
                }
            } else {
            for (int task = tw; task < np * 2; task += NTW) {
                const int s = task >> 1, qi = task & 1;
                float* Xp = Y + ((size_t)s * 64 + 32 * qi) * SPD_LS;
                const f32x16 a0 = spd_quadrant<1, false>(Xp, W);
                const f32x16 a1 = spd_quadrant(Xp, W + 32 * SPD_LS);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    Xp[quad_row(q, lh) * SPD_LS + li] = a0[q];
                    Xp[quad_row(q, lh) * SPD_LS + 32 + li] = a1[q];
                }
            }
            }
            lds_group_sync(sy + 1, gt += NTW);              // Y complete, W / W^T free
            if (tt == 0) __hip_atomic_store(sy + 4, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // ---- L_ik = Y_i to its block; trailing tiles A_ij -= Y_i Y_j^T (i >= j > k), tile (k+1,k+1) first ----
            {
                for (int task = tw; task < nupd; task += NTW) {
                    const int qi = (task >> 1) & 1, qj = task & 1, p = task >> 2;
                    int si = 0;
                    while ((si + 1) * (si + 2) / 2 <= p) ++si;
                    const int sj = p - si * (si + 1) / 2;
                    if (si == sj && qi == 0 && qj == 1) continue;      // diagonal tile: mirrored from its (1,0) quadrant
                    float* T0 = Hs + (size_t)sym_idx(k + 1 + si, k + 1 + sj, K) * LQP_BLK;
                    float* C = T0 + (32 * qi) * 64 + 32 * qj + li;
                    f32x16 cur;
                    if (task == tw) cur = pre;
                    else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) cur[q] = C[quad_row(q, lh) * 64];
                    }
                    if constexpr (F16) {
                        const f32x16 acc = f2_quadrant<0, 4>(Yc + ((size_t)si * 64 + 32 * qi) * F2_ROW + lane_b,
                                                             Yc + ((size_t)sj * 64 + 32 * qj) * F2_ROW + lane_b);
                        const float un = -(ysc[2 * si + qi] * ysc[2 * sj + qj]);
#pragma unroll
                        for (int q = 0; q < 16; ++q) cur[q] = __builtin_fmaf(acc[q], un, cur[q]);
                    } else {
                    const f32x16 acc = spd_quadrant(Y + ((size_t)si * 64 + 32 * qi) * SPD_LS,
                                                    Y + ((size_t)sj * 64 + 32 * qj) * SPD_LS);
                    cur -= acc;
                    }
                    if (p == 0) {                                      // the next diagonal tile also goes to the chain waves
                        float* Sq = St + (32 * qi) * 64 + 32 * qj + li;
#pragma unroll
                        for (int q = 0; q < 16; ++q) Sq[quad_row(q, lh) * 64] = cur[q];
                        if (qi == 1 && qj == 0) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) St[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_fetch_add(sy + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) C[quad_row(q, lh) * 64] = cur[q];
                    if (si == sj && qi == 1 && qj == 0) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) T0[li * 64 + 32 + quad_row(q, lh)] = cur[q];
                    }
                }
                if constexpr (!F16) {              // (F16: L_ik went to its block from the accumulators)
                for (int s = 0; s < np; ++s) {
                    float* dst = Hs + (size_t)sym_idx(k + 1 + s, k, K) * LQP_BLK;
                    for (int v = tt; v < 1024; v += NTT)
                        *(V4<float>*)(dst + v * 4) = *(const V4<float>*)(Y + ((size_t)s * 64 + (v >> 4)) * SPD_LS + (v & 15) * 4);
                }
                }
            }
            lds_group_sync(sy + 1, gt += NTW);              // all tiles of this step are written (the next staging reads them)
            if (dbg && tt == 0) dbg[7] += clock64() - c0;
        }
    }
    __syncthreads();
    if (tid == 0 && flag[0] != 0 && *info == 0) *info = flag[0];
}

// L L^T X = V with the factor above for up to NR right-hand sides at once: X[c * xs + ..] (LDS, 64 K values each),
// c < nc <= NR, solved in place.  Every block of the factor is loaded once for all of them and the 7 barriers per block column are shared: the solve is bound by those
// round trips, not by arithmetic (1 + m separate solves of the backward pass: 73 k cycles at n = 500, m = 1).
// acc: NR * 64 K floats, t: NR * 64, part: NR * NW * 64 of LDS.
template <int NR>
__device__ __forceinline__ void wg_chol_solve_n(const float* __restrict__ Ls, const int K, float* __restrict__ X, const int xs,
                                                const int nc, float* __restrict__ acc, float* __restrict__ t,
                                                float* __restrict__ part, unsigned long long* __restrict__ dbg = nullptr) {
    constexpr int MC = SPD_MAXK - 1;
    unsigned long long c0 = dbg ? clock64() : 0;                         // blocks below a diagonal block, at most
    const int tid = threadIdx.x, r = tid >> 4, cq = tid & 15, lane = tid & 63, w = tid >> 6;
    const int K64 = K * 64;
    const int tc = tid >> 6, te = tid & 63;                  // (column, element) of the per-column 64-vectors
    auto blk4 = [&](const int i, const int j) -> V4<float> {
        return *(const V4<float>*)(Ls + (size_t)sym_idx(i, j, K) * LQP_BLK + tid * 4);
    };
    for (int e = tid; e < NR * K64; e += LQP_NT) acc[e] = 0.f;
    // All blocks of a block column are requested together, the next diagonal block one step ahead: one after the other
    // (a loop with a run-time trip count) every load's L2 latency sat in the chain.
    V4<float> bd = blk4(0, 0);
    __syncthreads();
    // ---- L y = v, column by column ----
    for (int j = 0; j < K; ++j) {
        V4<float> bc[MC];
#pragma unroll
        for (int u = 0; u < MC; ++u)
            if (j + 1 + u < K) bc[u] = blk4(j + 1 + u, j);
        if (tc < nc) t[tc * 64 + te] = X[(size_t)tc * xs + j * 64 + te] - acc[tc * K64 + j * 64 + te];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            if (c < nc) {
                const float s1 = rowgroup_sum<LQP_NT>(dot4(bd, *(const V4<float>*)(t + c * 64 + cq * 4)));
                if (cq == 0) X[(size_t)c * xs + j * 64 + r] = s1;
            }
        }
        if (j + 1 < K) bd = blk4(j + 1, j + 1);
        __syncthreads();
        V4<float> yj[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) yj[c] = *(const V4<float>*)(X + (size_t)(c < nc ? c : 0) * xs + j * 64 + cq * 4);
        for (int u0 = 0; j + 1 + u0 < K; u0 += MC) {         // (more than MC blocks below: K > SPD_MAXK, chunk by chunk)
            if (u0 > 0) {
#pragma unroll
                for (int u = 0; u < MC; ++u)
                    if (j + 1 + u0 + u < K) bc[u] = blk4(j + 1 + u0 + u, j);
            }
#pragma unroll
            for (int u = 0; u < MC; ++u) {
                const int i = j + 1 + u0 + u;
                if (i < K) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        if (c < nc) {
                            const float s1 = rowgroup_sum<LQP_NT>(dot4(bc[u], yj[c]));
                            if (cq == 0) acc[c * K64 + i * 64 + r] += s1;  // (row r of block row i always belongs to this thread)
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- L^T x = y, from the last block column up ----
    auto fold = [&](float (&a2)[4], const int c) {           // column sums of this wave's 4 rows -> part[c][w][64]
        col_fold<4>(a2);                                      // (lane swaps on pairs of elements, no LDS round trips)
        part[((size_t)c * LQP_NW + w) * 64 + cq * 4 + col_fold_elem(lane)] = a2[0];
    };
    if (dbg && threadIdx.x == 0) { const unsigned long long c1 = clock64(); dbg[4] += c1 - c0; c0 = c1; }
    bd = blk4(K - 1, K - 1);
    for (int j = K - 1; j >= 0; --j) {
        V4<float> bc[MC];
#pragma unroll
        for (int u = 0; u < MC; ++u)
            if (j + 1 + u < K) bc[u] = blk4(j + 1 + u, j);
        float a2[NR][4];
#pragma unroll
        for (int c = 0; c < NR; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) a2[c][e] = 0.f;
        for (int u0 = 0; j + 1 + u0 < K; u0 += MC) {
            if (u0 > 0) {
#pragma unroll
                for (int u = 0; u < MC; ++u)
                    if (j + 1 + u0 + u < K) bc[u] = blk4(j + 1 + u0 + u, j);
            }
#pragma unroll
            for (int u = 0; u < MC; ++u) {
                const int i = j + 1 + u0 + u;
                if (i < K) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        const float xi = X[(size_t)(c < nc ? c : 0) * xs + i * 64 + r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) a2[c][e] += bc[u].v[e] * xi;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NR; ++c) fold(a2[c], c);
        __syncthreads();
        if (tc < nc) {
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < LQP_NW; ++ww) sum += part[((size_t)tc * LQP_NW + ww) * 64 + te];
            t[tc * 64 + te] = X[(size_t)tc * xs + j * 64 + te] - sum;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const float tr = t[c * 64 + r];
#pragma unroll
            for (int e = 0; e < 4; ++e) a2[c][e] = bd.v[e] * tr;
            fold(a2[c], c);
        }
        if (j > 0) bd = blk4(j - 1, j - 1);
        __syncthreads();
        if (tc < nc) {
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < LQP_NW; ++ww) sum += part[((size_t)tc * LQP_NW + ww) * 64 + te];
            X[(size_t)tc * xs + j * 64 + te] = sum;
        }
        __syncthreads();
    }
    if (dbg && threadIdx.x == 0) dbg[5] += clock64() - c0;
}

}  // namespace lqp
