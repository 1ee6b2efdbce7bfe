// Batched LU with partial pivoting: one workgroup factors one N x N matrix in
// place (row-major, leading dimension ld, ld % 64 == 0), LAPACK getrf layout
// and 1-based pivots.  Replaces torch.linalg.lu_factor at
// lqp_py/solve_box_qp_admm_torch.py:215,254 and lqp_py/lu_layer.py:10,31.
//
// Right-looking, panel width PB:
//   1. panel  : thread r keeps row k0+r of the panel in registers; per column
//               a wave max-|.| search (DPP inside 16-lane rows; first index
//               wins ties, as isamax), the pivot row is broadcast through
//               LDS, rank-1 update in registers;
//   2. swaps  : the PB row interchanges are composed into one gather map and
//               applied to the columns left and right of the panel, one
//               thread per column;
//   3. U12    : the same threads solve L11 * U12 = (P A)12 in registers;
//   4. update : A22 -= L21 * U12 with L21^T and U12 staged in LDS: MFMA
//               32x32x2 f32 tiles (C tile prefetched one tile ahead and used
//               as the accumulator: D = C - L21 U12), or 8x4 VALU tiles.
#pragma once
#include "lqp_common.hpp"

namespace lqp {

template <typename T, int PB> struct LuLds {
    // byte offsets inside the dynamic LDS block, all multiples of 32
    int lt, up, l11, rowp, rowj, wval, wrcp, widx, wtid, pidx, src, xdst, xsrc, cnt, total;
    __host__ __device__ explicit LuLds(int Mpad) {
        int o = 0;
        lt = o;   o += PB * Mpad * (int)sizeof(T);
        up = o;   o += PB * Mpad * (int)sizeof(T);
        l11 = o;  o += round_up(PB * (PB + 1) * (int)sizeof(T), 32);
        rowp = o; o += 2 * LQP_NW * PB * (int)sizeof(T);       // candidate pivot rows [parity][wave][PB]
        rowj = o; o += round_up(2 * PB * (int)sizeof(T), 32);  // row at the diagonal position [parity][PB]
        wval = o; o += round_up(2 * LQP_NW * (int)sizeof(T), 32);
        wrcp = o; o += round_up(2 * LQP_NW * (int)sizeof(T), 32);
        widx = o; o += round_up(2 * LQP_NW * 4, 32);
        wtid = o; o += round_up(2 * LQP_NW * 4, 32);
        pidx = o; o += round_up(PB * 4, 32);
        src = o;  o += Mpad * 4;
        xdst = o; o += round_up(PB * 4, 32);
        xsrc = o; o += round_up(PB * 4, 32);
        cnt = o;  o += 32;
        total = o;
    }
};

// ---- wave-level arg-max without control flow --------------------------------
__device__ __forceinline__ float readlane_t(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double readlane_t(double v, int l) {
    long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float hwmax(float a, float b) { return fmaxf(a, b); }     // v_max_f32 (DPP-fusable)
__device__ __forceinline__ double hwmax(double a, double b) { return fmax(a, b); }
template <typename T> __device__ __forceinline__ T row16_max(T m) {
    m = hwmax(m, dpp<0xB1>(m));
    m = hwmax(m, dpp<0x4E>(m));
    m = hwmax(m, dpp<0x124>(m));
    m = hwmax(m, dpp<0x128>(m));
    return m;
}
__device__ __forceinline__ int row16_min_i32(int m) {
    m = min(m, dpp_i32<0xB1>(m));
    m = min(m, dpp_i32<0x4E>(m));
    m = min(m, dpp_i32<0x124>(m));
    m = min(m, dpp_i32<0x128>(m));
    return m;
}
// max of `key` over the wave (wave-uniform) and the LOWEST lane that holds it: with keys
// ordered by row this is isamax's "first index wins ties".
template <typename T> __device__ __forceinline__ void wave_argmax(const T key, T& best, int& lane_of_best) {
    const T m = row16_max(key);
    best = hwmax(hwmax(readlane_t(m, 0), readlane_t(m, 16)), hwmax(readlane_t(m, 32), readlane_t(m, 48)));
    lane_of_best = __ffsll((unsigned long long)__ballot(key == best)) - 1;
}

// MFMA trailing update (f32 only): each wave owns 32x32 tiles of A22.
// A operand (32x2 slice of L21): lane l holds L21[i = l&31][k = l>>5];
// B operand (2x32 slice of U12): lane l holds U12[k = l>>5][j = l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PB, int NT>
__device__ __forceinline__ void lu_trailing_mfma_f32(float* __restrict__ A22, const int ld, const int M2,
                                                      const float* __restrict__ LT, const float* __restrict__ UP,
                                                      const int Mpad) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nt = (M2 + 31) >> 5;
    const int ntiles = nt * nt;
    // lane part of every address (same for all 16 accumulator registers); the tile / register
    // part is wave-uniform and stays in SGPRs
    const int voff = 4 * lh * ld + li;
    auto load_c = [&](int t, f32x16& c) {
        const int ti = t / nt, tj = t - ti * nt;
        const float* base = A22 + (size_t)(ti << 5) * ld + (tj << 5);
        const bool colok = (tj << 5) + li < M2;
        const int rlim = M2 - (ti << 5) - 4 * lh;          // valid iff qrow < rlim
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int qrow = (q & 3) + 8 * (q >> 2);
            c[q] = (colok && qrow < rlim) ? base[(size_t)qrow * ld + voff] : 0.f;
        }
    };
    for (int t = __builtin_amdgcn_readfirstlane(w); t < ntiles; t += NT / 64) {
        f32x16 cur;
        load_c(t, cur);                         // in flight underneath the MFMA chain below
        const int ti = t / nt, tj = t - ti * nt;
        const int i0 = ti << 5, j0 = tj << 5;
        const float* lt = LT + i0 + li + lh * Mpad;
        const float* up = UP + j0 + li + lh * Mpad;
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
        float* base = A22 + (size_t)i0 * ld + j0;
        const bool colok = j0 + li < M2;
        const int rlim = M2 - i0 - 4 * lh;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int qrow = (q & 3) + 8 * (q >> 2);
            if (colok && qrow < rlim) base[(size_t)qrow * ld + voff] = cur[q];
        }
    }
}

// The same for float64: 16 x 16 tiles of v_mfma_f64_16x16x4_f64 (the one-workgroup kernels above 512 rows -- the sizes the
// two-workgroup LU does not take -- ran their trailing update on the vector unit: 4.0 ms per factorisation at N = 522, B = 128).
//   A operand (16x4 slice of L21): lane l holds L21[i = l & 15][k = l >> 4];  B operand: U12[k = l >> 4][j = l & 15];
//   C/D: lane l holds rows 4 q + (l >> 4), q = 0..3, of column l & 15 (tools/microbench/mfma_f64_layout.hip)
typedef double f64x4_lu __attribute__((ext_vector_type(4)));
template <int PB, int NT>
__device__ __forceinline__ void lu_trailing_mfma_f64(double* __restrict__ A22, const int ld, const int M2,
                                                      const double* __restrict__ LT, const double* __restrict__ UP,
                                                      const int Mpad) {
    static_assert(PB % 4 == 0, "panel width in steps of the instruction's depth");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lh = lane >> 4;
    const int nt = (M2 + 15) >> 4;
    const int ntiles = nt * nt;
    for (int t = __builtin_amdgcn_readfirstlane(w); t < ntiles; t += NT / 64) {
        const int ti = t / nt, tj = t - ti * nt;
        const int i0 = ti << 4, j0 = tj << 4;
        double* base = A22 + (size_t)(i0 + lh) * ld + j0 + li;
        const bool colok = j0 + li < M2;
        const int rlim = M2 - i0 - lh;                    // register q holds row 4 q + lh
        f64x4_lu cur;
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = (colok && 4 * q < rlim) ? base[(size_t)(4 * q) * ld] : 0.0;
        const double* lt = LT + i0 + li + lh * Mpad;
        const double* up = UP + j0 + li + lh * Mpad;
        f64x4_lu acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < PB; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[kk * Mpad], up[kk * Mpad], acc, 0, 0, 0);
        cur -= acc;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (colok && 4 * q < rlim) base[(size_t)(4 * q) * ld] = cur[q];
    }
}

// LDS scratch of the panel factorisation
template <typename T> struct PanelLds {
    T* rowP; T* wval; T* wrcp; int* widx; int* wtid; int* pidx; int* cnt;
};

// The PB columns of one panel.  `sync` is the barrier among the waves that hold panel rows: the whole
// workgroup (plain variant) or the panel group only (lookahead variant, the other waves are busy with
// the previous panel's trailing update).
template <typename T, int PB, typename SYNC>
__device__ __forceinline__ void lu_panel_columns(V4<T> (&row4)[PB / 4], int& curpos, bool& done, const bool wact,
                                                 const int pb, const int k0, const int r, const PanelLds<T>& S,
                                                 SYNC&& sync) {
    typedef V4<T> vec;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T* const rowP = S.rowP; T* const wval = S.wval; T* const wrcp = S.wrcp;
    int* const widx = S.widx; int* const wtid = S.wtid; int* const pidx = S.pidx; int* const cnt = S.cnt;
                // ---- unblocked panel factorisation, one column at a time ----
                // Rows never move between threads: thread r keeps the row it loaded, `curpos` is the
                // position LAPACK's explicit interchanges would have moved that row to, `done` marks
                // rows already used as pivots.  Per column: wave arg-max of |a_rj| over the live rows
                // (ties: smallest position, as isamax), each wave's winner publishes its row, ONE
                // barrier, cross-wave arg-max, branch-free rank-1 update against the broadcast pivot
                // row.  The publish buffers alternate with the column parity, so column j+1 never
                // overwrites what a slow wave still reads for column j.
    #pragma unroll
                for (int j = 0; j < PB; ++j) {
                    if (j < pb) {
                        const int par = j & 1;
                        T* cand = rowP + par * (LQP_NW * PB);
                        T* wv = wval + par * LQP_NW;
                        int* wi = widx + par * LQP_NW;
                        int* wt = wtid + par * LQP_NW;
                        T* wr = wrcp + par * LQP_NW;
                        if (wact) {
                            const T aj = row4[j >> 2].v[j & 3];
                            const T key = done ? T(-1) : tabs(aj);
                            T myrcp = T(1) / aj;                 // independent of the arg-max chain below
                            asm volatile("" : "+v"(myrcp));      // keep it out of the winner-only branch
                            T bw; int lb;
                            wave_argmax(key, bw, lb);
                            const unsigned long long tied = __ballot(key == bw);
                            if (__popcll(tied) > 1) {                            // rare: smallest position wins
                                int cp = (key == bw) ? curpos : 0x7fffffff;
                                cp = row16_min_i32(cp);
                                cp = min(min(__builtin_amdgcn_readlane(cp, 0), __builtin_amdgcn_readlane(cp, 16)),
                                         min(__builtin_amdgcn_readlane(cp, 32), __builtin_amdgcn_readlane(cp, 48)));
                                lb = __ffsll((unsigned long long)__ballot(key == bw && curpos == cp)) - 1;
                            }
                            if (lane == lb) {
                                wv[w] = bw;
                                wr[w] = myrcp;
                                wi[w] = curpos;
                                wt[w] = r;
    #pragma unroll
                                for (int q = 0; q < PB / 4; ++q) *(vec*)(cand + w * PB + 4 * q) = row4[q];
                            }
                        } else if (lane == 0) {
                            wv[w] = T(-2);                                       // can never win
                        }
                        sync();
                        if (wact) {
                            // cross-wave arg-max: lane l looks at wave (l & 15)'s winner
                            const T cv = wv[lane & 15];
                            const int ci = wi[lane & 15];
                            const int ct = wt[lane & 15];
                            const T best = row16_max(cv);
                            const unsigned long long tied = __ballot(cv == best) & 0xFFFFull;
                            int ww = __ffsll(tied) - 1;
                            if (__popcll(tied) > 1) {
                                const int cp = row16_min_i32((cv == best) ? ci : 0x7fffffff);
                                ww = __ffsll((unsigned long long)__ballot(cv == best && ci == cp) & 0xFFFFull) - 1;
                            }
                            const int pivpos = __builtin_amdgcn_readlane(ci, ww);    // pivot row's position before the swap
                            const int bi = __builtin_amdgcn_readlane(ct, ww);        // thread that owns the pivot row
                            // the whole pivot row, 16 B per LDS read, all reads independent
                            const T* rowPc = cand + ww * PB;
                            vec pr[PB / 4];
    #pragma unroll
                            for (int q = j >> 2; q < PB / 4; ++q) pr[q] = *(const vec*)(rowPc + 4 * q);
                            const T rinv = wr[ww];
                            if (r == bi) {
                                pidx[j] = pivpos;      // pivots / info go to global memory once per panel
                                if (!(best > T(0)) && cnt[1] == 0) cnt[1] = k0 + j + 1;
                                curpos = j;
                                done = true;
                            } else if (curpos == j) {
                                curpos = pivpos;       // the row that sat on the diagonal takes the pivot's old place
                            }
                            // branch-free elimination: finished rows (and a zero pivot column, as getf2) use l = 0
                            const bool upd = !done && (best > T(0));
                            const T aj = row4[j >> 2].v[j & 3];
                            const T l = upd ? aj * rinv : T(0);
                            row4[j >> 2].v[j & 3] = upd ? l : aj;
    #pragma unroll
                            for (int c = j + 1; c < PB; ++c)
                                row4[c >> 2].v[c & 3] -= l * pr[c >> 2].v[c & 3];
                        }
                    }
                }

}

// dbg (optional): 4 cycle counters per workgroup: panel, swaps+U12, trailing update, total
// NT = threads in the workgroup (512 or 1024); rows per panel N - k0 <= NT and columns N - PB <= NT
template <typename T, int PB, bool USE_MFMA, int NT>
__device__ __forceinline__ void wg_lu_factor(T* __restrict__ A, const int N, const int ld, int* __restrict__ ipiv,
                                             int* __restrict__ info, char* __restrict__ smem,
                                             unsigned long long* __restrict__ dbg) {
    const int Mpad = round_up(N, 64);
    const LuLds<T, PB> L(Mpad);
    T* LT = (T*)(smem + L.lt);
    T* UP = (T*)(smem + L.up);
    T* L11 = (T*)(smem + L.l11);
    T* rowP = (T*)(smem + L.rowp);
    T* rowJ = (T*)(smem + L.rowj);
    T* wval = (T*)(smem + L.wval);
    T* wrcp = (T*)(smem + L.wrcp);
    int* widx = (int*)(smem + L.widx);
    int* wtid = (int*)(smem + L.wtid);
    int* pidx = (int*)(smem + L.pidx);
    int* src = (int*)(smem + L.src);
    int* xdst = (int*)(smem + L.xdst);
    int* xsrc = (int*)(smem + L.xsrc);
    int* cnt = (int*)(smem + L.cnt);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    typedef V4<T> vec;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);     // first row handled by this wave
    if (tid == 0) cnt[1] = 0;       // first zero pivot (1-based) seen by this factorisation, 0 = none
    if (tid < 2 * LQP_NW) wval[tid] = T(-2);     // slots of waves that do not exist (NT < 1024) never win
    unsigned long long t_panel = 0, t_swap = 0, t_trail = 0, t0 = 0, t_begin = 0;
    if (dbg) t_begin = clock64();

    for (int k0 = 0; k0 < N; k0 += PB) {
        const int pb = (N - k0 < PB) ? (N - k0) : PB;
        const int M = N - k0;
        const int M2 = M - pb;
        const int r = tid;
        const bool act = r < M;
        if (dbg) t0 = clock64();
        {
            // this thread's panel row as PB/4 four-wide vectors: row element c = row4[c >> 2].v[c & 3]
            vec row4[PB / 4];
            int curpos = r;             // LAPACK position of this thread's row
            bool done = !act;           // already a pivot row (or no row at all)
            const bool wact = wbase < M;   // this wave holds rows of the panel (loop-invariant, wave-uniform)
            // ---- load this thread's panel row ----
            if (pb == PB) {
#pragma unroll
                for (int q = 0; q < PB / 4; ++q) {
                    if (act) row4[q] = *(const vec*)(A + (size_t)(k0 + r) * ld + k0 + 4 * q);
                    else { row4[q].v[0] = row4[q].v[1] = row4[q].v[2] = row4[q].v[3] = T(0); }
                }
            } else {
#pragma unroll
                for (int c = 0; c < PB; ++c)
                    row4[c >> 2].v[c & 3] = (act && c < pb) ? A[(size_t)(k0 + r) * ld + k0 + c] : T(0);
            }
            if (tid == 0) *cnt = 0;

            // ---- unblocked panel factorisation, one column at a time (barrier = whole workgroup) ----
            {
                const PanelLds<T> S{rowP, wval, wrcp, widx, wtid, pidx, cnt};
                lu_panel_columns<T, PB>(row4, curpos, done, wact, pb, k0, r, S, [] { __syncthreads(); });
            }

            // ---- write the factored panel back at its final position; stage L11 and L21^T in LDS ----
            if (act) {
                if (pb == PB) {
#pragma unroll
                    for (int q = 0; q < PB / 4; ++q)
                        *(vec*)(A + (size_t)(k0 + curpos) * ld + k0 + 4 * q) = row4[q];
                } else {
#pragma unroll
                    for (int c = 0; c < PB; ++c)
                        if (c < pb) A[(size_t)(k0 + curpos) * ld + k0 + c] = row4[c >> 2].v[c & 3];
                }
                if (curpos < pb) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) L11[curpos * (PB + 1) + c] = row4[c >> 2].v[c & 3];
                } else {
#pragma unroll
                    for (int c = 0; c < PB; ++c) LT[c * Mpad + (curpos - pb)] = row4[c >> 2].v[c & 3];
                }
                src[curpos] = r;                        // position -> original (relative) row
                // a displaced top row that ended below the panel top (its source is always < pb)
                if (curpos >= pb && curpos != r) {
                    const int q = atomicAdd(cnt, 1);
                    xdst[q] = curpos;
                    xsrc[q] = r;
                }
            }
        }
        __syncthreads();
        if (tid < pb) ipiv[k0 + tid] = k0 + pidx[tid] + 1;
        if (dbg) { const unsigned long long t1 = clock64(); t_panel += t1 - t0; t0 = t1; }
        const int ne = __builtin_amdgcn_readfirstlane(*cnt);
        bool anyswap = ne > 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j < pb && __builtin_amdgcn_readfirstlane(src[j]) != j) anyswap = true;

        // ---- apply the interchanges left and right of the panel; U12 = L11^-1 (PA)12 ----
        if (tid < N - pb) {
            const bool right = tid >= k0;
            const int col = right ? tid + pb : tid;
            if (right || anyswap) {
                T* Ac = A + (size_t)k0 * ld + col;
                T top[PB];
                // (1) new top rows: sources anywhere in the panel rows
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    top[j] = (j < pb) ? Ac[(size_t)__builtin_amdgcn_readfirstlane(src[j]) * ld] : T(0);
                // (2) displaced top rows go down, 8 at a time (their sources are original
                //     top rows, which are only overwritten in step (4))
                for (int q0 = 0; q0 < ne; q0 += 8) {
                    T ext[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        ext[q] = (q0 + q < ne) ? Ac[(size_t)__builtin_amdgcn_readfirstlane(xsrc[q0 + q]) * ld] : T(0);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q0 + q < ne) Ac[(size_t)__builtin_amdgcn_readfirstlane(xdst[q0 + q]) * ld] = ext[q];
                }
                // (3) U12 column: forward substitution with the unit-lower L11
                if (right) {
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
#pragma unroll
                        for (int i = j + 1; i < PB; ++i) top[i] -= L11[i * (PB + 1) + j] * top[j];
                    }
#pragma unroll
                    for (int j = 0; j < PB; ++j) UP[j * Mpad + (col - k0 - pb)] = top[j];
                }
                // (4) store the new top rows
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    if (j < pb && (right || __builtin_amdgcn_readfirstlane(src[j]) != j)) Ac[(size_t)j * ld] = top[j];
            }
        }
        __syncthreads();
        if (dbg) { const unsigned long long t1 = clock64(); t_swap += t1 - t0; t0 = t1; }

        // ---- trailing update A22 -= L21 * U12 ----
        if (M2 > 0) {
            T* A22 = A + (size_t)(k0 + pb) * ld + (k0 + pb);
            if constexpr (USE_MFMA && sizeof(T) == 4) {
                lu_trailing_mfma_f32<PB, NT>((float*)A22, ld, M2, (const float*)LT, (const float*)UP, Mpad);
            } else if constexpr (USE_MFMA) {
                lu_trailing_mfma_f64<PB, NT>((double*)A22, ld, M2, (const double*)LT, (const double*)UP, Mpad);
            } else {
                const int tj_n = (M2 + 3) >> 2, ti_n = (M2 + 7) >> 3;
                for (int t = tid; t < ti_n * tj_n; t += NT) {
                    const int ti = t / tj_n, tj = t - ti * tj_n;
                    const int i0 = ti << 3, j0 = tj << 2;
                    T acc[8][4];
#pragma unroll
                    for (int a = 0; a < 8; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] = T(0);
#pragma unroll 4
                    for (int k = 0; k < PB; ++k) {
                        const vec a0 = *(const vec*)(LT + k * Mpad + i0);
                        const vec a1 = *(const vec*)(LT + k * Mpad + i0 + 4);
                        const vec b0 = *(const vec*)(UP + k * Mpad + j0);
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                acc[a][b] += a0.v[a] * b0.v[b];
                                acc[a + 4][b] += a1.v[a] * b0.v[b];
                            }
                    }
                    const bool fullj = j0 + 3 < M2;
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        if (i0 + a < M2) {
                            T* p = A22 + (size_t)(i0 + a) * ld + j0;
                            if (fullj) {
                                vec c = *(const vec*)p;
#pragma unroll
                                for (int b = 0; b < 4; ++b) c.v[b] -= acc[a][b];
                                *(vec*)p = c;
                            } else {
#pragma unroll
                                for (int b = 0; b < 4; ++b)
                                    if (j0 + b < M2) p[b] -= acc[a][b];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (dbg) { const unsigned long long t1 = clock64(); t_trail += t1 - t0; }
    }
    if (tid == 0 && cnt[1] != 0) *info = cnt[1];
    if (dbg && tid == 0) {
        dbg[0] = t_panel; dbg[1] = t_swap; dbg[2] = t_trail; dbg[3] = clock64() - t_begin;
    }
}

// (A look-ahead variant inside ONE workgroup -- waves 0-7 factoring panel k+1 while waves 8-15 ran panel k's trailing update -- was
//  measured no faster in rounds 1-2 and is gone from the sources; the look-ahead that pays runs on two CUs: lqp_lu2.hpp.)
// Panel width: L21^T and U12 (2 * PB * Mpad elements) must fit in LDS, and the panel row plus the
// broadcast pivot row (2 * PB elements per thread) must fit the register budget without spilling:
// 128 VGPRs in a 1024-thread workgroup (f32: 16 columns, f64: 8), 256 VGPRs in a 512-thread one
// (f32: 32, f64: 16).  Measured: a wider panel that spills loses to the narrower one.
template <typename T> __host__ __device__ inline int lu_threads(int N) { return N <= 512 ? 512 : 1024; }
template <typename T> __host__ __device__ inline int lu_panel_width(int N) {
    const int Mpad = round_up(N, 64);
    const int budget = 128 * 1024;
    const int widest = (sizeof(T) == 4 ? 16 : 8) * (lu_threads<T>(N) == 512 ? 2 : 1);
    if (widest >= 32 && 2 * 32 * Mpad * (int)sizeof(T) <= budget) return 32;
    if (widest >= 16 && 2 * 16 * Mpad * (int)sizeof(T) <= budget) return 16;
    return 8;
}

}  // namespace lqp
