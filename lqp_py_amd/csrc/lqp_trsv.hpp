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
#include "lqp_common.hpp"

namespace lqp {

#ifndef LQP_PF
#define LQP_PF 8   // blocks in flight per thread (f32: 128 KB per workgroup)
#endif
#ifndef LQP_PF64
#define LQP_PF64 6 // ... of the float64 stream (8 x 8 registers next to the solve's own spilled 39 of the loop kernel's 128)
#endif
template <typename T> __host__ __device__ constexpr int ring_pf() { return sizeof(T) == 8 ? LQP_PF64 : LQP_PF; }

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
        // (four blocks' loads before their stores, across block rows: one load in flight per thread left the copy at a block
        //  per memory round trip)
        if (doL) {
            int k = 1, j = 0;
            while (k < K) {
                vec t[4];
                long sl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sl[u] = -1;
                    if (k < K) {
                        t[u] = load_padded4(LU, N, ld, k * LQP_NB + row, j * LQP_NB + cq * 4, vec_ok);
                        sl[u] = (long)k * (k + 1) / 2 + j;
                        if (++j == k) { ++k; j = 0; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (sl[u] >= 0) *(vec*)(Lpk + (size_t)sl[u] * LQP_BLK + tid * 4) = t[u];
            }
        }
        if (doU) {
            int k = K - 2, j = K - 1;
            while (k >= 0) {
                vec t[4];
                long sl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sl[u] = -1;
                    if (k >= 0) {
                        const int kr = K - 1 - k;
                        t[u] = load_padded4(LU, N, ld, k * LQP_NB + row, j * LQP_NB + cq * 4, vec_ok);
                        sl[u] = (long)kr * (kr + 1) / 2 + (K - 1 - j);
                        if (--j == k) { --k; j = K - 1; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (sl[u] >= 0) *(vec*)(Upk + (size_t)sl[u] * LQP_BLK + tid * 4) = t[u];
            }
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
            // Lane c carries column c of the inverse.  The substitution runs in chunks of 16 rows: the chunk's 16 entries
            // live in registers; the entries of earlier chunks are read back from the destination block (this lane's own stores;
            // L1 / L2: every term a memory round trip, 96 in sequence per block) -- or, when this workgroup packs ONE half of the
            // factor (part != 0: the other triangle of the staged block is nobody's), from the staged block itself: row i of the
            // factor is read for row i of the inverse only, so a finished chunk's rows are free and take the inverse's rows.
            // (Stores under a per-lane condition, to keep the other triangle for a second wave: 339 / 944 spilled registers.  All
            // 64 entries in registers -- 128 VGPRs in float64 -- spilled 435 registers in k_pack<double>: 0.2 ms per call at
            // N = 266.)
            const bool inplace = part != 0;
            constexpr int CH = 16;
            T* dst;
            T* Tw = Tb + g * LQP_BLK;
            if (lower) {
                dst = Lpk + ((size_t)kb * (kb + 1) / 2 + kb) * LQP_BLK;
                for (int i0 = 0; i0 < LQP_NB; i0 += CH) {
                    T acc[CH];
#pragma unroll
                    for (int r = 0; r < CH; ++r) acc[r] = (i0 + r == c) ? T(1) : T(0);
#pragma unroll 2
                    for (int j = 0; j < i0; ++j) {
                        const T xj = inplace ? Tk[j * LQP_NB + c] : dst[j * LQP_NB + c];
#pragma unroll
                        for (int r = 0; r < CH; ++r) acc[r] -= Tk[(i0 + r) * LQP_NB + j] * xj;
                    }
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
#pragma unroll
                        for (int r2 = 0; r2 < r; ++r2) acc[r] -= Tk[(i0 + r) * LQP_NB + i0 + r2] * acc[r2];
                    }
#pragma unroll
                    for (int r = 0; r < CH; ++r) dst[(i0 + r) * LQP_NB + c] = acc[r];
                    if (inplace) {
#pragma unroll
                        for (int r = 0; r < CH; ++r) Tw[(i0 + r) * LQP_NB + c] = acc[r];
                    }
                }
            } else {
                const int kr = K - 1 - kb;   // block rows the U phase visits before this one
                dst = Upk + ((size_t)kr * (kr + 1) / 2 + kr) * LQP_BLK;
                for (int i0 = LQP_NB - CH; i0 >= 0; i0 -= CH) {
                    T acc[CH];
#pragma unroll
                    for (int r = 0; r < CH; ++r) acc[r] = (i0 + r == c) ? T(1) : T(0);
#pragma unroll 2
                    for (int j = LQP_NB - 1; j >= i0 + CH; --j) {
                        const T xj = inplace ? Tk[j * LQP_NB + c] : dst[j * LQP_NB + c];
#pragma unroll
                        for (int r = 0; r < CH; ++r) acc[r] -= Tk[(i0 + r) * LQP_NB + j] * xj;
                    }
#pragma unroll
                    for (int r = CH - 1; r >= 0; --r) {
#pragma unroll
                        for (int r2 = CH - 1; r2 > r; --r2) acc[r] -= Tk[(i0 + r) * LQP_NB + i0 + r2] * acc[r2];
                        acc[r] = acc[r] / Tk[(i0 + r) * LQP_NB + i0 + r];
                    }
#pragma unroll
                    for (int r = 0; r < CH; ++r) dst[(i0 + r) * LQP_NB + c] = acc[r];
                    if (inplace) {
#pragma unroll
                        for (int r = 0; r < CH; ++r) Tw[(i0 + r) * LQP_NB + c] = acc[r];
                    }
                }
            }
        }
    }

    // ---- destination of each rhs row under the LAPACK interchanges ----
    // (the interchanges through LDS: read from global memory one by one, N scalar loads in sequence were 40 us at N = 501)
    const int Np = K * LQP_NB;
    if (doL) {
        int* pv = (int*)smem;
        __syncthreads();                               // (the staged diagonal blocks are done with)
        for (int i = tid; i < N; i += LQP_NT) pv[i] = ipiv[i] - 1;
        __syncthreads();
        for (int r = tid; r < Np; r += LQP_NT) {
            int pos = r;
            if (r < N) {
                for (int i = 0; i < N; ++i) {
                    const int pi = pv[i];
                    pos = pos == i ? pi : (pos == pi ? i : pos);
                }
            }
            dest[r] = pos;
        }
    }
}

// ---------------------------------------------------------------------------
// streaming solve.  v (LDS, Np elements) holds P*rhs on entry (padding = 0) and the solution on exit;
// tmp is 64 elements of LDS scratch.
//
// NT threads walk the block stream together: thread t owns the EPT = 4096 / NT consecutive elements
// [EPT*t, EPT*(t+1)) of every block (NT = 1024: one 16-B vector, row t>>4; NT = 512: two vectors,
// row t>>3), so a block is one fully coalesced load instruction (or two) per thread, and the lanes that
// share a block row are adjacent (16 or 8 lanes): the row dot-product is reduced with DPP only.
// Loads run LQP_PF blocks ahead of their use through a register ring that wraps into the next solve.
// ---------------------------------------------------------------------------
template <typename T, int NT> struct Frag {
    static constexpr int NV = LQP_BLK / NT / 4;      // 16-B vectors per thread per block
    V4<T> q[NV];
};
template <typename T, int NT> struct BlockStream {
    Frag<T, NT> buf[ring_pf<T>()];
};

// (uniform block base + 32-bit per-thread byte offset: the load takes its base from SGPRs)
template <typename T, int NT>
__device__ __forceinline__ Frag<T, NT> frag_load(const T* __restrict__ blk) {
    Frag<T, NT> f;
    const char* base = (const char*)blk;
    const unsigned int off = threadIdx.x * (unsigned int)(Frag<T, NT>::NV * sizeof(V4<T>));
#pragma unroll
    for (int i = 0; i < Frag<T, NT>::NV; ++i) f.q[i] = *(const V4<T>*)(base + (off + i * (unsigned int)sizeof(V4<T>)));
    return f;
}
template <typename T, int NT>
__device__ __forceinline__ void frag_store(T* __restrict__ blk, const Frag<T, NT>& f) {
    V4<T>* p = (V4<T>*)blk + threadIdx.x * Frag<T, NT>::NV;
#pragma unroll
    for (int i = 0; i < Frag<T, NT>::NV; ++i) p[i] = f.q[i];
}

template <typename T>
__device__ __forceinline__ T dot4(const V4<T>& a, const V4<T>& b) {
    return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2] + a.v[3] * b.v[3];
}
// dot of this thread's fragment with the matching slice of an LDS vector (64 values at `vec64`)
template <typename T, int NT>
__device__ __forceinline__ T frag_dot(const Frag<T, NT>& f, const T* __restrict__ vec64, const int col0) {
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < Frag<T, NT>::NV; ++i) acc += dot4(f.q[i], *(const V4<T>*)(vec64 + col0 + 4 * i));
    return acc;
}
// sum over the lanes that share a block row (16 lanes for NT = 1024, 8 for NT = 512); all of them get it
template <int NT, typename T> __device__ __forceinline__ T rowgroup_sum(T v) {
    v += dpp<0xB1>(v);                // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);                // quad_perm [2,3,0,1]
    if constexpr (NT == 1024) {
        v += dpp<0x124>(v);           // row_ror:4
        v += dpp<0x128>(v);           // row_ror:8
    } else {
        v += dpp<0x141>(v);           // row_half_mirror: lane i <-> 7 - i inside each group of 8
    }
    return v;
}

template <typename T, int NT = LQP_NT>
__device__ __forceinline__ void stream_prime(BlockStream<T, NT>& st, const T* __restrict__ packed, const int S) {
#pragma unroll
    for (int i = 0; i < ring_pf<T>(); ++i)                 // (slots past the end get block 0 again: never used)
        st.buf[i] = frag_load<T, NT>(packed + (size_t)(i < S ? i : 0) * LQP_BLK);
}

// ---- on-chip residency of the head of the stream --------------------------------------------
// The loop kernel is at the per-CU streaming limit, so bytes that never leave the chip are pure
// gain: the first RREG blocks of the stream live in otherwise idle VGPRs, the next LQP_RLDS blocks in
// LDS, for every iteration of the launch; only blocks [R0, S) are streamed through the prefetch ring.
// f32 only; needs S >= R0 + LQP_PF and, for the cyclic ring, (S - R0) % LQP_PF == 0 (K = 8: 72 blocks).
// 1024 threads (128 VGPRs): 8 register blocks; 512 threads (256 VGPRs): 16.
#ifndef LQP_RREG
#define LQP_RREG 8      // measured: 8+8 resident blocks, PF 8 is spill-free and fastest (16 spills, 12 needs PF 4)
#endif
#ifndef LQP_RREG512
#define LQP_RREG512 12
#endif
#ifndef LQP_RLDS
#define LQP_RLDS 8
#endif
template <int NT> __host__ __device__ constexpr int resident_regs() { return NT == 512 ? LQP_RREG512 : LQP_RREG; }
template <int NT> __host__ __device__ constexpr int resident_total() { return resident_regs<NT>() + LQP_RLDS; }

template <typename T, int NT> struct ResidentRegs {
    Frag<T, NT> r[resident_regs<NT>()];
};

template <typename T, int NT>
__device__ __forceinline__ void resident_load(ResidentRegs<T, NT>& rr, T* __restrict__ lds_res, const T* __restrict__ packed) {
#pragma unroll
    for (int i = 0; i < resident_regs<NT>(); ++i) rr.r[i] = frag_load<T, NT>(packed + (size_t)i * LQP_BLK);
#pragma unroll
    for (int i = 0; i < LQP_RLDS; ++i)
        frag_store<T, NT>(lds_res + (size_t)i * LQP_BLK, frag_load<T, NT>(packed + (size_t)(resident_regs<NT>() + i) * LQP_BLK));
}

// walk state of the blocked solve (all wave-uniform)
template <typename T> struct SolveWalk {
    int phase, k, j;
    T acc;
};

// consume one 64x64 block of the stream
template <typename T, int NT>
__device__ __forceinline__ void solve_block(SolveWalk<T>& wk, const Frag<T, NT>& blk, const int K, T* __restrict__ v,
                                            T* __restrict__ tmp) {
    constexpr int EPT = LQP_BLK / NT;               // elements per thread per block
    constexpr int LPR = LQP_NB / EPT;               // lanes per block row
    const int row = threadIdx.x / LPR, cq = threadIdx.x % LPR, col0 = cq * EPT;
    wk.phase = __builtin_amdgcn_readfirstlane(wk.phase);       // (the walk is uniform: SGPRs, scalar branches)
    wk.k = __builtin_amdgcn_readfirstlane(wk.k);
    wk.j = __builtin_amdgcn_readfirstlane(wk.j);
    if (wk.j != wk.k) {
        wk.acc += frag_dot<T, NT>(blk, v + wk.j * LQP_NB, col0);
        wk.j += wk.phase ? -1 : 1;
    } else {
        const T a = rowgroup_sum<NT>(wk.acc);
        if (cq == 0) tmp[row] = v[wk.k * LQP_NB + row] - a;
        wg_barrier_lds();
        const T y = rowgroup_sum<NT>(frag_dot<T, NT>(blk, tmp, col0));
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
template <typename T, int NT>
__device__ __forceinline__ void stream_prime_from(BlockStream<T, NT>& st, const T* __restrict__ packed, const int first,
                                                  const int S) {
#pragma unroll
    for (int i = 0; i < LQP_PF; ++i)                 // (slots past the end get a valid block too: never used)
        st.buf[i] = frag_load<T, NT>(packed + (size_t)(first + i < S ? first + i : 0) * LQP_BLK);
}

// solve with a resident head: blocks [0, RREG) from registers, [RREG, R0) from LDS, [R0, S) streamed
template <typename T, int NT>
__device__ __forceinline__ void wg_packed_solve_resident(BlockStream<T, NT>& st, const ResidentRegs<T, NT>& rr,
                                                         const T* __restrict__ lds_res, const T* __restrict__ packed,
                                                         const int K, T* __restrict__ v, T* __restrict__ tmp,
                                                         const bool cyclic) {
    constexpr int R0 = resident_total<NT>();
    const int S = K * (K + 1);
    SolveWalk<T> wk;
    wk.phase = 0; wk.k = 0; wk.j = 0; wk.acc = T(0);
#pragma unroll
    for (int s = 0; s < resident_regs<NT>(); ++s) solve_block<T, NT>(wk, rr.r[s], K, v, tmp);
    for (int s = 0; s < LQP_RLDS; ++s) {
        const Frag<T, NT> blk = frag_load<T, NT>(lds_res + (size_t)s * LQP_BLK);
        solve_block<T, NT>(wk, blk, K, v, tmp);
    }
    const int Sr = S - R0;                           // streamed blocks, a multiple of LQP_PF (loop_resident_ok)
    (void)cyclic;                                    // (always: the ring wraps into the next solve)
    for (int s0 = 0; s0 < Sr; s0 += LQP_PF) {
#pragma unroll
        for (int i = 0; i < LQP_PF; ++i) {
            const int s = s0 + i;
            // one unconditional refill per step: the compiler can count the loads in flight and waits for the
            // oldest one only (conditional refills made it drain the ring, s_waitcnt vmcnt(0), before every block)
            const Frag<T, NT> blk = st.buf[i];
            int nx = s + LQP_PF;
            if (nx >= Sr) nx -= Sr;
            st.buf[i] = frag_load<T, NT>(packed + (size_t)(R0 + nx) * LQP_BLK);
            solve_block<T, NT>(wk, blk, K, v, tmp);
        }
    }
}

// plain streaming solve (no resident head).  cyclic: S % LQP_PF == 0 and the caller will solve again with
// the same factor: the tail of this solve already fetches the head of the next one.
template <typename T, int NT = LQP_NT>
__device__ __forceinline__ void wg_packed_solve(BlockStream<T, NT>& st, const T* __restrict__ packed, const int K,
                                                T* __restrict__ v, T* __restrict__ tmp, const bool cyclic) {
    const int S = K * (K + 1);
    // virtual stream length: a multiple of the ring depth, so that ANY factor can be walked cyclically (round 4: the ring used
    // to wrap only when S % LQP_PF == 0 -- K = 8 -- and was drained and re-primed before every solve otherwise, e.g. K = 5:
    // the float64 hard distribution); padding steps re-load block 0 and drop it
    constexpr int PF = ring_pf<T>();
    const int Sv = round_up(S, PF);
    SolveWalk<T> wk;
    wk.phase = 0; wk.k = 0; wk.j = 0; wk.acc = T(0);
    for (int s0 = 0; s0 < Sv; s0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int s = s0 + i;
            // one unconditional refill per step (exact vmcnt, see above); past the end it re-loads block 0
            const Frag<T, NT> blk = st.buf[i];
            int nx = s + PF;
            if (nx >= Sv && cyclic) nx -= Sv;
            st.buf[i] = frag_load<T, NT>(packed + (size_t)(nx < S ? nx : 0) * LQP_BLK);
            if (s < S) solve_block<T, NT>(wk, blk, K, v, tmp);
        }
    }
}

}  // namespace lqp
