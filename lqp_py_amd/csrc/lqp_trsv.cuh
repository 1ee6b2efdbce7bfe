// Cached-factor triangular solves (getrs) as a pure HBM/L2 stream.
// Replaces torch.linalg.lu_solve at lqp_py/solve_box_qp_admm_torch.py:267
// (the ADMM hot loop) and lqp_py/lu_layer.py:33,52.
//
// Layout ("solve-ordered panels"): the N x N factor is cut into 64 x 64
// blocks (Np = 64*K, padded with the identity) and stored in exactly the
// order the solve consumes them,
//   L phase, k = 0..K-1   :  L(k,0) .. L(k,k-1),  inv(L(k,k))
//   U phase, k = K-1..0   :  U(k,K-1) .. U(k,k+1), inv(U(k,k))
// each block row-major and contiguous (16 KB f32 / 32 KB f64).  Thread t of
// the 1024-thread workgroup owns elements [4t, 4t+4) of every block, i.e.
// row t>>4, columns 4(t&15)..+3, so one block is one perfectly coalesced
// 16-B-per-lane (f32) load per thread and the whole solve is a linear read of
// K(K+1) blocks.  Diagonal blocks are pre-inverted so every block is the same
// dense 64x64 GEMV: no sequential substitution inside a block, only a
// 16-lane DPP reduction per block row.  Loads run LQP_PF blocks ahead of
// their use through a register ring (independent of the data dependence).
#pragma once
#include "lqp_common.cuh"

namespace lqp {

#ifndef LQP_PF
#define LQP_PF 8   // blocks in flight per thread (f32: 128 KB per workgroup)
#endif

__host__ __device__ inline size_t packed_blocks(int K) { return (size_t)K * (K + 1); }

// ---------------------------------------------------------------------------
// pack: LAPACK-layout LU (row-major, ld) + pivots  ->  solve-ordered panels,
// plus dest[r] = position of original rhs row r after the row interchanges.
// One workgroup per matrix; LDS = pack_lds_bytes<T>().
// ---------------------------------------------------------------------------
template <typename T> __host__ __device__ constexpr int pack_group() { return sizeof(T) == 4 ? 8 : 4; }
template <typename T> __host__ __device__ constexpr int pack_lds_bytes() {
    return pack_group<T>() * LQP_BLK * (int)sizeof(T);   // 128 KB: staged LU diagonal blocks
}

template <typename T>
__device__ __forceinline__ V4<T> load_padded4(const T* __restrict__ LU, const int N, const int ld,
                                              const int gr, const int gc, const bool vec_ok) {
    V4<T> v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v.v[e] = T(0);
    if (gr < N) {
        if (vec_ok && gc + 3 < N) v = *(const V4<T>*)(LU + (size_t)gr * ld + gc);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gc + e < N) v.v[e] = LU[(size_t)gr * ld + gc + e];
        }
    }
    return v;
}

template <typename T>
__device__ __forceinline__ void wg_pack_factor(const T* __restrict__ LU, const int N, const int ld,
                               const int* __restrict__ ipiv, T* __restrict__ packed,
                               int* __restrict__ dest, char* __restrict__ smem, const bool vec_ok,
                               const int part = 0) {
    // part 0: whole factor; part 1: L panels + inv(L diagonal) + rhs permutation; part 2: U panels + inv(U
    // diagonal).  The two halves are independent, so two workgroups (two CUs) can pack one factor.
    const bool doL = part != 2, doU = part != 1;
    const int K = round_up(N, LQP_NB) / LQP_NB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row = tid >> 4, cq = tid & 15;
    typedef V4<T> vec;
    T* Lpk = packed;
    T* Upk = packed + (size_t)(K * (K + 1) / 2) * LQP_BLK;

    // ---- off-diagonal blocks: straight copies with zero padding ----
    {
        size_t s = 0;
        for (int k = 0; doL && k < K; ++k) {
            for (int j = 0; j < k; ++j, ++s)
                *(vec*)(Lpk + s * LQP_BLK + tid * 4) = load_padded4(LU, N, ld, k * LQP_NB + row, j * LQP_NB + cq * 4, vec_ok);
            ++s;   // diagonal slot, filled below
        }
        s = 0;
        for (int k = K - 1; doU && k >= 0; --k) {
            for (int j = K - 1; j > k; --j, ++s)
                *(vec*)(Upk + s * LQP_BLK + tid * 4) = load_padded4(LU, N, ld, k * LQP_NB + row, j * LQP_NB + cq * 4, vec_ok);
            ++s;
        }
    }

    // ---- diagonal blocks: explicit inverses ----
    // Groups of G diagonal LU blocks are staged in LDS (identity-padded); wave
    // 2g inverts the unit-lower part of block g, wave 2g+1 the upper part.
    // Lane c carries column c of the inverse in registers (substitution with
    // wave-uniform LDS broadcast reads of the factor).
    constexpr int G = pack_group<T>();
    T* Tb = (T*)smem;
    for (int g0 = 0; g0 < K; g0 += G) {
        __syncthreads();
        for (int g = 0; g < G && g0 + g < K; ++g) {
            const int base = (g0 + g) * LQP_NB;
            vec v = load_padded4(LU, N, ld, base + row, base + cq * 4, vec_ok);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gi = base + row, gj = base + cq * 4 + e;
                if (gi >= N || gj >= N) v.v[e] = (row == cq * 4 + e) ? T(1) : T(0);
            }
            *(vec*)(Tb + g * LQP_BLK + tid * 4) = v;
        }
        __syncthreads();
        const int g = w >> 1;
        const bool lower = (w & 1) == 0;
        if (g < G && g0 + g < K && (lower ? doL : doU)) {
            const int kb = g0 + g;
            const T* Tk = Tb + g * LQP_BLK;
            const int c = lane;
            T x[LQP_NB];
            T* dst;
            if (lower) {
#pragma unroll
                for (int i = 0; i < LQP_NB; ++i) {
                    T sacc = (i == c) ? T(1) : T(0);
#pragma unroll
                    for (int j = 0; j < i; ++j) sacc -= Tk[i * LQP_NB + j] * x[j];
                    x[i] = sacc;
                }
                dst = Lpk + ((size_t)kb * (kb + 1) / 2 + kb) * LQP_BLK;
            } else {
#pragma unroll
                for (int i = LQP_NB - 1; i >= 0; --i) {
                    T sacc = (i == c) ? T(1) : T(0);
#pragma unroll
                    for (int j = i + 1; j < LQP_NB; ++j) sacc -= Tk[i * LQP_NB + j] * x[j];
                    x[i] = sacc / Tk[i * LQP_NB + i];
                }
                const int kr = K - 1 - kb;   // block rows the U phase visits before this one
                dst = Upk + ((size_t)kr * (kr + 1) / 2 + kr) * LQP_BLK;
            }
#pragma unroll
            for (int i = 0; i < LQP_NB; ++i) dst[i * LQP_NB + c] = x[i];
        }
    }

    // ---- destination of each rhs row under the LAPACK interchanges ----
    const int Np = K * LQP_NB;
    for (int r = tid; doL && r < Np; r += LQP_NT) {
        int pos = r;
        if (r < N) {
            for (int i = 0; i < N; ++i) {
                const int pi = ipiv[i] - 1;
                if (pos == i) pos = pi;
                else if (pos == pi) pos = i;
            }
        }
        dest[r] = pos;
    }
}

// ---------------------------------------------------------------------------
// streaming solve.  v (LDS, Np elements) holds P*rhs on entry (padding = 0)
// and the solution on exit; tmp is 64 elements of LDS scratch.  The register
// ring `buf` must have been primed with stream_prime() (the first LQP_PF
// blocks), which lets the caller issue those loads early.
// ---------------------------------------------------------------------------
template <typename T> struct BlockStream {
    V4<T> buf[LQP_PF];
};

template <typename T>
__device__ __forceinline__ void stream_prime(BlockStream<T>& st, const T* __restrict__ packed, const int S) {
    const V4<T>* p = (const V4<T>*)packed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < LQP_PF; ++i)
        if (i < S) st.buf[i] = p[(size_t)i * (LQP_BLK / 4)];
}

template <typename T>
__device__ __forceinline__ T dot4(const V4<T>& a, const V4<T>& b) {
    return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2] + a.v[3] * b.v[3];
}

// ---- on-chip residency of the head of the stream --------------------------------------------
// The loop kernel is at the per-CU streaming limit, so bytes that never leave the chip are pure
// gain: the first LQP_RREG blocks of the stream live in otherwise idle VGPRs (one 16-B vector per
// thread per block), the next LQP_RLDS blocks in LDS, for every iteration of the launch; only
// blocks [R0, S) are streamed through the prefetch ring.  f32 only; needs S >= R0 and, for the
// cyclic ring, (S - R0) % LQP_PF == 0 (K = 8: 72 blocks, 16 resident, 56 streamed).
#ifndef LQP_RREG
#define LQP_RREG 8      // measured: 8+8 resident blocks, PF 8 is spill-free and fastest (16 spills, 12 needs PF 4)
#endif
#ifndef LQP_RLDS
#define LQP_RLDS 8
#endif
#define LQP_R0 (LQP_RREG + LQP_RLDS)

template <typename T> struct ResidentRegs {
    V4<T> r[LQP_RREG];
};

template <typename T>
__device__ __forceinline__ void resident_load(ResidentRegs<T>& rr, T* __restrict__ lds_res, const T* __restrict__ packed) {
    const V4<T>* p = (const V4<T>*)packed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < LQP_RREG; ++i) rr.r[i] = p[(size_t)i * (LQP_BLK / 4)];
#pragma unroll
    for (int i = 0; i < LQP_RLDS; ++i)
        *(V4<T>*)(lds_res + (size_t)i * LQP_BLK + threadIdx.x * 4) = p[(size_t)(LQP_RREG + i) * (LQP_BLK / 4)];
}

// walk state of the blocked solve (all wave-uniform)
template <typename T> struct SolveWalk {
    int phase, k, j;
    T acc;
};

// consume one 64x64 block of the stream
template <typename T>
__device__ __forceinline__ void solve_block(SolveWalk<T>& wk, const V4<T>& blk, const int K, T* __restrict__ v,
                                            T* __restrict__ tmp, const int row, const int cq) {
    if (wk.j != wk.k) {
        const V4<T> yv = *(const V4<T>*)(v + wk.j * LQP_NB + cq * 4);
        wk.acc += dot4(blk, yv);
        wk.j += wk.phase ? -1 : 1;
    } else {
        const T a = row16_sum(wk.acc);
        if (cq == 0) tmp[row] = v[wk.k * LQP_NB + row] - a;
        wg_barrier_lds();
        const V4<T> tv = *(const V4<T>*)(tmp + cq * 4);
        const T y = row16_sum(dot4(blk, tv));
        if (cq == 0) v[wk.k * LQP_NB + row] = y;
        wg_barrier_lds();
        wk.acc = T(0);
        if (wk.phase == 0) {
            if (wk.k == K - 1) { wk.phase = 1; wk.j = K - 1; }
            else { ++wk.k; wk.j = 0; }
        } else {
            --wk.k; wk.j = K - 1;
        }
    }
}

// prime the ring with the first LQP_PF STREAMED blocks (those after the resident head)
template <typename T>
__device__ __forceinline__ void stream_prime_from(BlockStream<T>& st, const T* __restrict__ packed, const int first,
                                                  const int S) {
    const V4<T>* p = (const V4<T>*)packed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < LQP_PF; ++i)
        if (first + i < S) st.buf[i] = p[(size_t)(first + i) * (LQP_BLK / 4)];
}

// solve with a resident head: blocks [0, RREG) from registers, [RREG, R0) from LDS, [R0, S) streamed
template <typename T>
__device__ __forceinline__ void wg_packed_solve_resident(BlockStream<T>& st, const ResidentRegs<T>& rr,
                                                         const T* __restrict__ lds_res, const T* __restrict__ packed,
                                                         const int K, T* __restrict__ v, T* __restrict__ tmp,
                                                         const bool cyclic) {
    const int tid = threadIdx.x;
    const int row = tid >> 4, cq = tid & 15;
    const int S = K * (K + 1);
    SolveWalk<T> wk;
    wk.phase = 0; wk.k = 0; wk.j = 0; wk.acc = T(0);
#pragma unroll
    for (int s = 0; s < LQP_RREG; ++s) solve_block(wk, rr.r[s], K, v, tmp, row, cq);
    for (int s = 0; s < LQP_RLDS; ++s) {
        const V4<T> blk = *(const V4<T>*)(lds_res + (size_t)s * LQP_BLK + tid * 4);
        solve_block(wk, blk, K, v, tmp, row, cq);
    }
    const V4<T>* p = (const V4<T>*)packed + tid;
    const int Sr = S - LQP_R0;                       // streamed blocks
    for (int s0 = 0; s0 < Sr; s0 += LQP_PF) {
#pragma unroll
        for (int i = 0; i < LQP_PF; ++i) {
            const int s = s0 + i;
            if (s < Sr) {
                const V4<T> blk = st.buf[i];
                {
                    int nx = s + LQP_PF;
                    if (nx >= Sr && cyclic) nx -= Sr;
                    if (nx < Sr) st.buf[i] = p[(size_t)(LQP_R0 + nx) * (LQP_BLK / 4)];
                }
                solve_block(wk, blk, K, v, tmp, row, cq);
            }
        }
    }
}

// cyclic: S % LQP_PF == 0 and the caller will solve again with the same factor:
// the tail of this solve already fetches the head of the next one.
template <typename T>
__device__ __forceinline__ void wg_packed_solve(BlockStream<T>& st, const T* __restrict__ packed, const int K,
                                                T* __restrict__ v, T* __restrict__ tmp, const bool cyclic) {
    const int tid = threadIdx.x;
    const int row = tid >> 4, cq = tid & 15;
    const int S = K * (K + 1);
    const V4<T>* p = (const V4<T>*)packed + tid;
    // wave-uniform walk state
    int phase = 0;              // 0: L (k ascending, j ascending), 1: U (k descending, j descending)
    int k = 0, j = 0;           // current block row k, next column block j
    T acc = T(0);
    for (int s0 = 0; s0 < S; s0 += LQP_PF) {
#pragma unroll
        for (int i = 0; i < LQP_PF; ++i) {
            const int s = s0 + i;
            if (s < S) {
                const V4<T> blk = st.buf[i];
                {
                    int nx = s + LQP_PF;
                    if (nx >= S && cyclic) nx -= S;
                    if (nx < S) st.buf[i] = p[(size_t)nx * (LQP_BLK / 4)];
                }
                const bool diag = (j == k);
                if (!diag) {
                    const V4<T> yv = *(const V4<T>*)(v + j * LQP_NB + cq * 4);
                    acc += dot4(blk, yv);
                    j += phase ? -1 : 1;
                } else {
                    acc = row16_sum(acc);
                    if (cq == 0) tmp[row] = v[k * LQP_NB + row] - acc;
                    wg_barrier_lds();
                    const V4<T> tv = *(const V4<T>*)(tmp + cq * 4);
                    T y = row16_sum(dot4(blk, tv));
                    if (cq == 0) v[k * LQP_NB + row] = y;
                    wg_barrier_lds();
                    acc = T(0);
                    if (phase == 0) {
                        if (k == K - 1) { phase = 1; j = K - 1; /* k stays K-1 */ }
                        else { ++k; j = 0; }
                    } else {
                        --k; j = K - 1;
                    }
                }
            }
        }
    }
}

}  // namespace lqp
