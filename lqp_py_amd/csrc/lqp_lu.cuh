// Batched LU with partial pivoting: one workgroup factors one N x N matrix in
// place (row-major, leading dimension ld, ld % 64 == 0), LAPACK getrf layout
// and 1-based pivots.  Replaces torch.linalg.lu_factor at
// lqp_py/solve_box_qp_admm_torch.py:215,254 and lqp_py/lu_layer.py:10,31.
//
// Right-looking, panel width PB:
//   1. panel  : thread r keeps row k0+r of the panel in registers; per column
//               a wave max-|.| search (first index wins ties, as isamax),
//               the pivot row is broadcast through LDS, rank-1 update in regs;
//   2. swaps  : the PB row interchanges are composed into one gather map and
//               applied to the columns left and right of the panel, one
//               thread per column (all loads before all stores);
//   3. U12    : the same threads solve L11 * U12 = (P A)12 in registers;
//   4. update : A22 -= L21 * U12 with L21^T and U12 staged in LDS
//               (8x4 register tiles, or MFMA 32x32x2 f32 tiles).
#pragma once
#include "lqp_common.cuh"

namespace lqp {

template <typename T, int PB> struct LuLds {
    // byte offsets inside the dynamic LDS block, all multiples of 32
    int lt, up, l11, rowp, rowj, wval, widx, pidx, src, xdst, xsrc, cnt, total;
    __host__ __device__ explicit LuLds(int Mpad) {
        int o = 0;
        lt = o;   o += PB * Mpad * (int)sizeof(T);
        up = o;   o += PB * Mpad * (int)sizeof(T);
        l11 = o;  o += round_up(PB * (PB + 1) * (int)sizeof(T), 32);
        rowp = o; o += round_up(PB * (int)sizeof(T), 32);
        rowj = o; o += round_up(PB * (int)sizeof(T), 32);
        wval = o; o += round_up(LQP_NW * (int)sizeof(T), 32);
        widx = o; o += round_up(LQP_NW * 4, 32);
        pidx = o; o += round_up(PB * 4, 32);
        src = o;  o += Mpad * 4;
        xdst = o; o += round_up(PB * 4, 32);
        xsrc = o; o += round_up(PB * 4, 32);
        cnt = o;  o += 32;
        total = o;
    }
};

// MFMA trailing update (f32 only): each wave owns 32x32 tiles of A22.
// A operand (32x2 slice of L21): lane l holds L21[i = l&31][k = l>>5];
// B operand (2x32 slice of U12): lane l holds U12[k = l>>5][j = l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PB>
__device__ __forceinline__ void lu_trailing_mfma_f32(float* __restrict__ A22, const int ld, const int M2,
                                                      const float* __restrict__ LT, const float* __restrict__ UP,
                                                      const int Mpad) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nt = (M2 + 31) >> 5;
    for (int t = w; t < nt * nt; t += LQP_NW) {
        const int ti = t / nt, tj = t - ti * nt;
        const int i0 = ti << 5, j0 = tj << 5;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int kk = 0; kk < PB; kk += 2) {
            const float a = LT[(kk + lh) * Mpad + i0 + li];
            const float b = UP[(kk + lh) * Mpad + j0 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        const int col = j0 + li;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (row < M2 && col < M2) {
                float* p = A22 + (size_t)row * ld + col;
                *p = *p - acc[q];
            }
        }
    }
}

template <typename T, int PB, bool USE_MFMA>
__device__ void wg_lu_factor(T* __restrict__ A, const int N, const int ld, int* __restrict__ ipiv,
                             int* __restrict__ info, char* __restrict__ smem) {
    const int Mpad = round_up(N, 64);
    const LuLds<T, PB> L(Mpad);
    T* LT = (T*)(smem + L.lt);
    T* UP = (T*)(smem + L.up);
    T* L11 = (T*)(smem + L.l11);
    T* rowP = (T*)(smem + L.rowp);
    T* rowJ = (T*)(smem + L.rowj);
    T* wval = (T*)(smem + L.wval);
    int* widx = (int*)(smem + L.widx);
    int* pidx = (int*)(smem + L.pidx);
    int* src = (int*)(smem + L.src);
    int* xdst = (int*)(smem + L.xdst);
    int* xsrc = (int*)(smem + L.xsrc);
    int* cnt = (int*)(smem + L.cnt);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    typedef V4<T> vec;

    for (int k0 = 0; k0 < N; k0 += PB) {
        const int pb = (N - k0 < PB) ? (N - k0) : PB;
        const int M = N - k0;
        const int M2 = M - pb;
        const int r = tid;
        const bool act = r < M;
        T row[PB];
        // ---- load this thread's panel row ----
        if (pb == PB) {
#pragma unroll
            for (int c = 0; c < PB; c += 4) {
                vec v;
                if (act) v = *(const vec*)(A + (size_t)(k0 + r) * ld + k0 + c);
                else { v.v[0] = v.v[1] = v.v[2] = v.v[3] = T(0); }
                row[c] = v.v[0]; row[c + 1] = v.v[1]; row[c + 2] = v.v[2]; row[c + 3] = v.v[3];
            }
        } else {
#pragma unroll
            for (int c = 0; c < PB; ++c) row[c] = (act && c < pb) ? A[(size_t)(k0 + r) * ld + k0 + c] : T(0);
        }
        if (tid == 0) *cnt = 0;

        // ---- unblocked panel factorisation, one column at a time ----
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            if (j < pb) {
                T key = (act && r >= j) ? tabs(row[j]) : T(-1);
                int idx = r;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const T ok = __shfl_xor(key, off);
                    const int oi = __shfl_xor(idx, off);
                    if (ok > key || (ok == key && oi < idx)) { key = ok; idx = oi; }
                }
                if (lane == 0) { wval[w] = key; widx[w] = idx; }
                __syncthreads();
                T best = wval[0];
                int bi = widx[0];
#pragma unroll
                for (int i = 1; i < LQP_NW; ++i) {
                    const T v = wval[i];
                    const int ii = widx[i];
                    if (v > best || (v == best && ii < bi)) { best = v; bi = ii; }
                }
                if (tid == 0) {
                    ipiv[k0 + j] = k0 + bi + 1;
                    pidx[j] = bi;
                    if (!(best > T(0)) && *info == 0) *info = k0 + j + 1;
                }
                if (r == bi) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) rowP[c] = row[c];
                }
                if (r == j) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) rowJ[c] = row[c];
                }
                __syncthreads();
                if (r == j) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) row[c] = rowP[c];
                } else if (r == bi) {
#pragma unroll
                    for (int c = 0; c < PB; ++c) row[c] = rowJ[c];
                }
                if (act && r > j) {
                    const T pv = rowP[j];
                    if (pv != T(0)) {
                        const T l = row[j] * (T(1) / pv);
                        row[j] = l;
#pragma unroll
                        for (int c = j + 1; c < PB; ++c) row[c] -= l * rowP[c];
                    }
                }
            }
        }

        // ---- write the factored panel back; stage L11 and L21^T in LDS ----
        if (act) {
            if (pb == PB) {
#pragma unroll
                for (int c = 0; c < PB; c += 4) {
                    vec v;
                    v.v[0] = row[c]; v.v[1] = row[c + 1]; v.v[2] = row[c + 2]; v.v[3] = row[c + 3];
                    *(vec*)(A + (size_t)(k0 + r) * ld + k0 + c) = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < PB; ++c)
                    if (c < pb) A[(size_t)(k0 + r) * ld + k0 + c] = row[c];
            }
            if (r < pb) {
#pragma unroll
                for (int c = 0; c < PB; ++c) L11[r * (PB + 1) + c] = row[c];
            } else {
#pragma unroll
                for (int c = 0; c < PB; ++c) LT[c * Mpad + (r - pb)] = row[c];
            }
        }
        // ---- compose the pb interchanges: where does original (relative) row r end up? ----
        if (act) {
            int pos = r;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                if (j < pb) {
                    const int pj = pidx[j];
                    if (pos == j) pos = pj;
                    else if (pos == pj) pos = j;
                }
            }
            src[pos] = r;
        }
        __syncthreads();
        if (act && r >= pb && src[r] != r) {
            const int q = atomicAdd(cnt, 1);
            xdst[q] = r;
            xsrc[q] = src[r];
        }
        __syncthreads();
        const int ne = *cnt;
        bool anyswap = ne > 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j < pb && src[j] != j) anyswap = true;

        // ---- apply the interchanges left and right of the panel; U12 = L11^-1 (PA)12 ----
        if (tid < N - pb) {
            const bool right = tid >= k0;
            const int col = right ? tid + pb : tid;
            if (right || anyswap) {
                T top[PB], ext[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    top[j] = (j < pb) ? A[(size_t)(k0 + src[j]) * ld + col] : T(0);
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    ext[q] = (q < ne) ? A[(size_t)(k0 + xsrc[q]) * ld + col] : T(0);
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
                    if (j < pb && (right || src[j] != j)) A[(size_t)(k0 + j) * ld + col] = top[j];
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    if (q < ne) A[(size_t)(k0 + xdst[q]) * ld + col] = ext[q];
            }
        }
        __syncthreads();

        // ---- trailing update A22 -= L21 * U12 ----
        if (M2 > 0) {
            T* A22 = A + (size_t)(k0 + pb) * ld + (k0 + pb);
            if constexpr (USE_MFMA) {
                lu_trailing_mfma_f32<PB>((float*)A22, ld, M2, (const float*)LT, (const float*)UP, Mpad);
            } else {
                const int tj_n = (M2 + 3) >> 2, ti_n = (M2 + 7) >> 3;
                for (int t = tid; t < ti_n * tj_n; t += LQP_NT) {
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
    }
}

// Panel width that keeps L21^T and U12 (2 * PB * Mpad elements) inside LDS.
template <typename T> __host__ __device__ inline int lu_panel_width(int N) {
    const int Mpad = round_up(N, 64);
    const int budget = 128 * 1024;
    if (2 * 32 * Mpad * (int)sizeof(T) <= budget) return 32;
    if (2 * 16 * Mpad * (int)sizeof(T) <= budget) return 16;
    return 8;
}

}  // namespace lqp
