// Batched LU with partial pivoting for matrices ABOVE the one-row-per-thread tier (1024 < N <= 2048): the same
// right-looking algorithm, LAPACK layout and pivot rule as wg_lu_factor (lqp_lu.hpp; replaces torch.linalg.lu_factor at
// lqp_py/solve_box_qp_admm_torch.py:215,254 and lqp_py/lu_layer.py:10,31 -- the reference's LAPACK calls take any size),
// with R = 2 panel rows per thread: thread t keeps the rows k0 + t and k0 + t + 1024 of the panel in registers.  The
// matrix stays in global memory (16 MB per matrix at N = 2048: L2 / Infinity Cache resident for small batches, HBM
// otherwise); L21^T and U12 of a panel are staged in LDS (2 * PB * Mpad elements: PB = 8 in f32, 4 in f64).  Kept apart
// from the tuned kernel on purpose: that one's register budget and barrier count are what the LU tier's speed rests on.
#pragma once
#include "lqp_lu.hpp"

namespace lqp {

// The PB columns of one panel, R rows per thread (row q of thread t: relative row t + q * NT).
template <typename T, int PB, int R, int NT>
__device__ __forceinline__ void lu_panel_columns_r(V4<T> (&row4)[R][PB / 4], int (&curpos)[R], bool (&done)[R],
                                                   const bool wact, const int pb, const int k0,
                                                   const PanelLds<T>& S) {
    typedef V4<T> vec;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T* const rowP = S.rowP; T* const wval = S.wval; T* const wrcp = S.wrcp;
    int* const widx = S.widx; int* const wtid = S.wtid; int* const pidx = S.pidx; int* const cnt = S.cnt;
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
                // this thread's best live row: largest |a_rj|, ties to the smaller LAPACK position (isamax)
                T key = T(-1), aj = T(1);
                int cp = 0x7fffffff, qs = 0;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const T a = row4[q][j >> 2].v[j & 3];
                    const T kq = done[q] ? T(-1) : tabs(a);
                    const bool better = kq > key || (kq == key && kq >= T(0) && curpos[q] < cp);
                    if (better) { key = kq; aj = a; cp = curpos[q]; qs = q; }
                }
                const T myrcp = T(1) / aj;
                T bw; int lb;
                wave_argmax(key, bw, lb);
                const unsigned long long tied = __ballot(key == bw);
                if (__popcll(tied) > 1) {                            // rare: smallest position wins
                    int c2 = (key == bw) ? cp : 0x7fffffff;
                    c2 = row16_min_i32(c2);
                    c2 = min(min(__builtin_amdgcn_readlane(c2, 0), __builtin_amdgcn_readlane(c2, 16)),
                             min(__builtin_amdgcn_readlane(c2, 32), __builtin_amdgcn_readlane(c2, 48)));
                    lb = __ffsll((unsigned long long)__ballot(key == bw && cp == c2)) - 1;
                }
                if (lane == lb) {
                    wv[w] = bw;
                    wr[w] = myrcp;
                    wi[w] = cp;
                    wt[w] = tid + qs * NT;
#pragma unroll
                    for (int q4 = 0; q4 < PB / 4; ++q4) {
                        vec o = row4[0][q4];
#pragma unroll
                        for (int q = 1; q < R; ++q)
                            if (qs == q) o = row4[q][q4];
                        *(vec*)(cand + w * PB + 4 * q4) = o;
                    }
                }
            } else if (lane == 0) {
                wv[w] = T(-2);                                       // can never win
            }
            __syncthreads();
            if (wact) {
                const T cv = wv[lane & 15];
                const int ci = wi[lane & 15];
                const int ct = wt[lane & 15];
                const T best = row16_max(cv);
                const unsigned long long tied = __ballot(cv == best) & 0xFFFFull;
                int ww = __ffsll(tied) - 1;
                if (__popcll(tied) > 1) {
                    const int c2 = row16_min_i32((cv == best) ? ci : 0x7fffffff);
                    ww = __ffsll((unsigned long long)__ballot(cv == best && ci == c2) & 0xFFFFull) - 1;
                }
                const int pivpos = __builtin_amdgcn_readlane(ci, ww);    // pivot row's position before the swap
                const int bi = __builtin_amdgcn_readlane(ct, ww);        // (thread + q * NT) that owns the pivot row
                const T* rowPc = cand + ww * PB;
                vec pr[PB / 4];
#pragma unroll
                for (int q4 = j >> 2; q4 < PB / 4; ++q4) pr[q4] = *(const vec*)(rowPc + 4 * q4);
                const T rinv = wr[ww];
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    if (tid + q * NT == bi) {
                        pidx[j] = pivpos;
                        if (!(best > T(0)) && cnt[1] == 0) cnt[1] = k0 + j + 1;
                        curpos[q] = j;
                        done[q] = true;
                    } else if (curpos[q] == j) {
                        curpos[q] = pivpos;
                    }
                    const bool upd = !done[q] && (best > T(0));
                    const T a = row4[q][j >> 2].v[j & 3];
                    const T l = upd ? a * rinv : T(0);
                    row4[q][j >> 2].v[j & 3] = upd ? l : a;
#pragma unroll
                    for (int c = j + 1; c < PB; ++c)
                        row4[q][c >> 2].v[c & 3] -= l * pr[c >> 2].v[c & 3];
                }
            }
        }
    }
}

template <typename T, int PB, bool USE_MFMA, int R>
__device__ __forceinline__ void wg_lu_factor_big(T* __restrict__ A, const int N, const int ld, int* __restrict__ ipiv,
                                                 int* __restrict__ info, char* __restrict__ smem) {
    constexpr int NT = LQP_NT;
    const int Mpad = round_up(N, 64);
    const LuLds<T, PB> L(Mpad);
    T* LT = (T*)(smem + L.lt);
    T* UP = (T*)(smem + L.up);
    T* L11 = (T*)(smem + L.l11);
    T* rowP = (T*)(smem + L.rowp);
    T* wval = (T*)(smem + L.wval);
    T* wrcp = (T*)(smem + L.wrcp);
    int* widx = (int*)(smem + L.widx);
    int* wtid = (int*)(smem + L.wtid);
    int* pidx = (int*)(smem + L.pidx);
    int* src = (int*)(smem + L.src);
    int* xdst = (int*)(smem + L.xdst);
    int* xsrc = (int*)(smem + L.xsrc);
    int* cnt = (int*)(smem + L.cnt);
    const int tid = threadIdx.x;
    typedef V4<T> vec;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    if (tid == 0) cnt[1] = 0;
    if (tid < 2 * LQP_NW) wval[tid] = T(-2);

    for (int k0 = 0; k0 < N; k0 += PB) {
        const int pb = (N - k0 < PB) ? (N - k0) : PB;
        const int M = N - k0;
        const int M2 = M - pb;
        {
            vec row4[R][PB / 4];
            int curpos[R];
            bool done[R];
            const bool wact = wbase < M;           // this wave holds live rows (its first row slot is inside the panel)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int r = tid + q * NT;
                const bool act = r < M;
                curpos[q] = r;
                done[q] = !act;
#pragma unroll
                for (int c = 0; c < PB; ++c)
                    row4[q][c >> 2].v[c & 3] = (act && c < pb) ? A[(size_t)(k0 + r) * ld + k0 + c] : T(0);
            }
            if (tid == 0) *cnt = 0;
            {
                const PanelLds<T> S{rowP, wval, wrcp, widx, wtid, pidx, cnt};
                lu_panel_columns_r<T, PB, R, NT>(row4, curpos, done, wact, pb, k0, S);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int r = tid + q * NT;
                if (r < M) {
                    const int cp = curpos[q];
#pragma unroll
                    for (int c = 0; c < PB; ++c)
                        if (c < pb) A[(size_t)(k0 + cp) * ld + k0 + c] = row4[q][c >> 2].v[c & 3];
                    if (cp < pb) {
#pragma unroll
                        for (int c = 0; c < PB; ++c) L11[cp * (PB + 1) + c] = row4[q][c >> 2].v[c & 3];
                    } else {
#pragma unroll
                        for (int c = 0; c < PB; ++c) LT[c * Mpad + (cp - pb)] = row4[q][c >> 2].v[c & 3];
                    }
                    src[cp] = r;
                    if (cp >= pb && cp != r) {
                        const int e = atomicAdd(cnt, 1);
                        xdst[e] = cp;
                        xsrc[e] = r;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < pb) ipiv[k0 + tid] = k0 + pidx[tid] + 1;
        const int ne = __builtin_amdgcn_readfirstlane(*cnt);
        bool anyswap = ne > 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j < pb && __builtin_amdgcn_readfirstlane(src[j]) != j) anyswap = true;

        // ---- interchanges left and right of the panel; U12 = L11^-1 (PA)12: one column at a time per thread ----
        for (int cc = tid; cc < N - pb; cc += NT) {
            const bool right = cc >= k0;
            const int col = right ? cc + pb : cc;
            if (right || anyswap) {
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
                    for (int j = 0; j < PB; ++j) UP[j * Mpad + (col - k0 - pb)] = top[j];
                }
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    if (j < pb && (right || __builtin_amdgcn_readfirstlane(src[j]) != j)) Ac[(size_t)j * ld] = top[j];
            }
        }
        __syncthreads();

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
#pragma unroll
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
    }
    if (tid == 0 && cnt[1] != 0) *info = cnt[1];
}

template <typename T> __host__ __device__ constexpr int lu_big_panel() { return sizeof(T) == 4 ? 8 : 4; }

}  // namespace lqp
