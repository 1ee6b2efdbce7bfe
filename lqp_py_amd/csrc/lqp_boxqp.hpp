// Kernels of the box-QP ADMM layer (forward setup / loop / epilogue, adaptive
// rho, fixed-point backward, KKT solve).  One 1024-thread workgroup per QP.
#pragma once
#include <type_traits>
#include "lqp_common.hpp"
#include "lqp_lu.hpp"
#include "lqp_lu_big.hpp"
#include "lqp_lu2.hpp"
#include "lqp_lu_wide.hpp"
#include "lqp_trsv.hpp"
#include "lqp_spd.hpp"

namespace lqp {

// ---- device-side status block (ints) ---------------------------------------
enum { ST_DONE = 0, ST_FINAL_ITER = 1, ST_GATE = 2, ST_NFACTOR = 3, ST_RHO_UPDATED = 4,
       ST_TIMEOUT = 5, ST_NCHECK = 6, ST_NOTSPD = 7,
       ST_VOTE = 8,      // 4 words: {ranks leaving the symmetric path, ranks with a singular KKT matrix, -, -} (strict global stop)
       ST_ANY_LB = 12, ST_ANY_UB = 13,      // some lower / upper bound of the batch is finite (:129-130): the per-problem
                                            // answers of k_fwd_setup (FwdParams::bflags), OR-ed by the forward's last kernel
       ST_RESUME = 14,   // the iteration at which the hot two-workgroup loop handed over to the continuation kernel (FwdParams::hot_past)
       ST_WORDS = 16 };
// host report (lqp_boxqp_ctrl.host_report): [ST_WORDS status words as workgroup 0 of the forward's last kernel sees them |
// B info words | B flag words], written straight into pinned host memory by that kernel
enum { RP_LB = 1, RP_UB = 2, RP_TIMEOUT = 4, RP_NOTSPD = 8 };
// per-check counters (uint32 x 4): not-optimal, arrivals, wants-rho, ratio-trigger
// (NOTOPT and ARRIVE share one aligned 64-bit word: the two-workgroup loop adds to and reads both with ONE atomic)
enum { CT_NOTOPT = 0, CT_ARRIVE = 1, CT_WANTS = 2, CT_TRIG = 3, CT_WORDS = 4 };
// verbose=True (reference :289-294: the largest primal and dual error of the batch at every check): non-negative floats order like
// their bit patterns, so one atomicMax per problem and value does it (first lap of the ring only: the trace holds `ring` checks)
template <typename T>
__device__ __forceinline__ void trace_check(unsigned int* __restrict__ vtrace, const int it, const int check_solved, const int ring,
                                            const T pri, const T dua) {
    if (!vtrace) return;
    const int c = it / check_solved;
    if (c >= ring) return;
    atomicMax(vtrace + 2 * c, __float_as_uint((float)pri));
    atomicMax(vtrace + 2 * c + 1, __float_as_uint((float)dua));
}
// per-problem scalars
enum { SC_RHO = 0, SC_PNORM = 1, SC_RATIO = 2, SC_WANTS = 3, SC_PRI = 4, SC_DUA = 5, SC_WORDS = 8 };   // PRI/DUA: errors of the last check

constexpr int XCHG_NPMAX = 4;                                // workgroups per QP of the shared loop, at most
constexpr int XCHG_WORDS = 2 * XCHG_NPMAX * SPD_MAXK * LQP_NB;      // exchange granules per QP of the shared loop: [parity][part][element]
// behind the granules of all QPs, per QP: [0..4) step granules of the resident sweep's workgroups | [4..8) the XCD ids its
// workgroups announce | [8..12) the same for the shared loop (XCHG_TAIL words; zeroed by the setup kernel)
constexpr int XCHG_TAIL = 12;
constexpr int DNX_WORDS = 2 * 256 * 2;   // dense LU-tier loop (lqp_dense.hpp): [parity][row <= 256][one (f32) or two (f64) words]
// the XCD this workgroup runs on (0..7)
__device__ __forceinline__ unsigned int my_xcd() { return (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu; }

template <typename T> struct FwdParams {
    int B, n, m, N, Np, K, ldq;          // ldq: leading dim of Qs
    int Ks, sym_rl, sym_rl_hot;          // symmetric-inverse path: 64-blocks of n, LDS-resident blocks of the loop
                                         // (continuation kernel / 512-thread first launch)
    unsigned long long* dbg;             // optional cycle counters (8 per problem), debug only
    int dbg_qpass;                       // LQP_DBG_QPASS=1: the caller's debug buffer has a second half of B x 8 words for the stamps of the sweep's pass over Q
    unsigned long long* dbg_setup;       // the same words for k_fwd_setup's stamps (LQP_DBG_SETUP=1: then only those)
    int spd;                             // 1: symmetric-inverse path (no KKT matrix M is assembled)
    int qs_lazy;                         // 1: the scaled matrix Qs is not stored; its readers compute (D_i Q_ij) D_j
    int eq_in_loop;                      // 1: k_admm_loop_split applies the equality correction to its register blocks (no k_spd_end)
    int hot_past;                        // 1: the persistent two-workgroup loop runs on PAST the iterations at which the reference may adapt
                                         //    rho (:237) as long as the counters of the check before say "nothing to update", leaves at one
                                         //    that does and names it in status[ST_RESUME]; the continuation launch starts there
    int hot_resume;                      // 1: this launch of the persistent two-workgroup loop starts at status[ST_RESUME] (behind a rho event
                                         //    the library handled between two launches: k_rho_update + gated refactorisation)
    int split_seg;                       // 1: k_admm_loop_split launched once per check segment for a batch LARGER than half the CUs (the
                                         //    pairs take their turns on the chip): workgroups 16 g + x and 16 g + 8 + x share problem 8 g + x
                                         //    (same XCD, neighbours in its dispatch queue), no verdict inside the kernel
    int seg_prev_slot;                   // ... counter slot of the last check before it0 (-1: none)
    int rho_late;                        // 1: rho = ||Qs||_F / sqrt(n) from the sums k_spd_begin leaves, added by k_spd_resident
    int prep_fused;                      // 3: no k_spd_prep at all -- the resident sweep reads Q itself (maxima, verdict, tiles; k_fwd_setup defers what needs D to it); 1 / 2: k_spd_prep ran BEFORE the setup kernel (2: one-workgroup tier, k_spd_inverse finishes the blocks) -- one pass over Q for the column maxima, the
                                         //    symmetry verdict and the UNSCALED blocks; the resident sweep scales them as it loads
                                         //    them and takes ||Qs||_F (-> rho) from its own tiles
    int ar_iter, ar_max, ring;           // adaptive-rho schedule and counter-ring length, for the in-kernel events
    // inputs
    const T *Q, *p, *A, *b, *lb, *ub, *rho_in, *beta_in;
    const int* bound_flags_in;           // optional {any_lb, any_ub} of a LARGER batch this call is a shard of (device), or null
    int xcd_local;                       // 1: workgroups that share a matrix and find themselves on ONE XCD exchange through its L2
    int* host_report;                    // optional pinned host memory, ST_WORDS + 2 B ints (see RP_*), or null
    int zero_words;                      // ints from `status` on (status block + counter ring) that workgroup 0 of k_fwd_setup zeroes
    // outputs
    T *x, *z, *u, *lams, *nus, *rho_out;
    // workspace
    T* Qs;            // B * n * ldq (scale) or unused
    T* M;             // B * Np * Np
    T* packed;        // B * K(K+1) * 4096
    T* vecs;          // B * vstride
    T* scal;          // B * SC_WORDS
    int* piv;         // B * Np
    int* dest;        // B * Np
    int* info;        // B
    int* bflags;      // B: RP_LB | RP_UB of problem b (k_fwd_setup)
    int* status;      // ST_WORDS
    unsigned int* counters;   // ring of CT_WORDS per check
    unsigned int* vtrace;     // nullptr, or [ring][2]: bit patterns of the largest primal / dual error over the batch per check (verbose)
    unsigned long long* xchg; // B * XCHG_WORDS granules: partial-product exchange of the two-workgroup loop (or null)
    unsigned long long* dnx;  // B * dnx_words granules: x-parts of the dense LU-tier loops, lqp_dense.hpp (or null)
    int dnx_words;            // granules per problem (DNX_WORDS, or what the W-workgroup form needs: 2 parities x n elements)
    size_t vstride;
    // controls
    int scale, rho_mode, beta_mode, check_solved, adaptive_rho;
    T eps_abs, eps_rel, rho_value, rho_min, rho_max, ar_tol, ar_inv_tol, ar_thr, beta_value;
};

// vector block of problem b: [ps | lbs | ubs | D | z | u | x | As (m*n) | bs | E | nu | cv | Tm (m*n) | s0]
// (cv, Tm, s0: constant term c, T = G S^-1 and S^-1 b of the symmetric-inverse path, lqp_spd.hpp)
template <typename T> struct VecView {
    T *ps, *lbs, *ubs, *D, *z, *u, *x, *As, *bs, *E, *nu, *cv, *Tm, *s0;
    __device__ VecView(T* base, int n, int m) {
        ps = base; lbs = ps + n; ubs = lbs + n; D = ubs + n; z = D + n; u = z + n; x = u + n;
        As = x + n; bs = As + (size_t)m * n; E = bs + m; nu = E + m;
        cv = nu + m; Tm = cv + n; s0 = Tm + (size_t)m * n;
    }
};
__host__ __device__ inline size_t vec_stride(int n, int m) { return (size_t)round_up(8 * n + 2 * m * n + 4 * m, 8); }

// ---------------------------------------------------------------------------
// setup: norms, auto-scaling, rho, KKT assembly, state init
// (lqp_py/solve_box_qp_admm_torch.py:124-131, 160-212, 221-223)
// LDS: red[NW * n] | d[n] | sel[8] | scratch[NW]
// ---------------------------------------------------------------------------
// (n <= 1024: one slab of column maxima per wave; above, the waves merge their maxima into 4 slabs in 4 rounds -- 16
//  slabs of 2048 doubles would not fit)
__host__ __device__ inline int setup_slabs(int n) { return n > 1024 ? 4 : LQP_NW; }
template <typename T> __host__ __device__ inline int setup_lds_bytes(int n) {
    return (round_up(setup_slabs(n) * n, 8) + round_up(n, 8) + 8 + LQP_NW + 8) * (int)sizeof(T);
}

// M = [[Qs + rho I, As^T], [As, 0]]  (solve_box_qp_admm_torch.py:206-212, 252).
// copy_q == false: the top-left block already holds Qs (written by the scaling pass), only
// rho is added on the diagonal.
template <typename T>
__device__ __forceinline__ void assemble_kkt_rows(const FwdParams<T>& P, const int b, const T* __restrict__ Qs, const int ldq,
                                                  const VecView<T>& V, const T rho, const bool copy_q,
                                                  const T* __restrict__ dsc = nullptr) {
    // dsc != nullptr: `Qs` is the UNSCALED matrix and dsc the scaling vector (FwdParams::qs_lazy on the LU path: the scaled matrix
    // is not kept for the refactorisations, they form (D_i Q_ij) D_j again -- the expression of the setup pass, the same bits)
    const int n = P.n, m = P.m, Np = P.Np;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T* M = P.M + (size_t)b * Np * Np;
    typedef V4<T> vec;
    const bool vec_ok = (ldq % 4 == 0) && ((((uintptr_t)Qs) % sizeof(vec)) == 0) && (n % 4 == 0 || dsc == nullptr);
    if (copy_q) {
        for (int i = w; i < n; i += LQP_NW) {
            const T* q = Qs + (size_t)i * ldq;
            T* mr = M + (size_t)i * Np;
            const T di = dsc ? dsc[i] : T(1);
            if (vec_ok) {
                for (int j = lane * 4; j < n; j += 256) {
                    if (j + 3 < n) {
                        vec v = *(const vec*)(q + j);
                        if (dsc) {
                            const vec dj = *(const vec*)(dsc + j);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v.v[e] = (di * v.v[e]) * dj.v[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (j + e == i) v.v[e] += rho;
                        *(vec*)(mr + j) = v;
                    } else {
                        for (int e = 0; e < 4 && j + e < n; ++e)
                            mr[j + e] = (dsc ? (di * q[j + e]) * dsc[j + e] : q[j + e]) + (i == j + e ? rho : T(0));
                    }
                }
            } else {
                for (int j = lane; j < n; j += 64) mr[j] = (dsc ? (di * q[j]) * dsc[j] : q[j]) + (i == j ? rho : T(0));
            }
        }
    } else {
        for (int i = threadIdx.x; i < n; i += LQP_NT) M[(size_t)i * Np + i] += rho;
    }
    for (int i = w; i < n; i += LQP_NW) {
        T* mr = M + (size_t)i * Np;
        for (int r = lane; r < m; r += 64) mr[n + r] = V.As[(size_t)r * n + i];
    }
    for (int r = w; r < m; r += LQP_NW) {
        T* mr = M + (size_t)(n + r) * Np;
        for (int j = lane; j < n; j += 64) mr[j] = V.As[(size_t)r * n + j];
        for (int c = lane; c < m; c += 64) mr[n + c] = T(0);
    }
}

// ---- the two passes of the scaling over Q (16-B row pieces).  NQ: column quads of 256 per lane, RIF: rows in
//      flight per wave.  Loads are UNCONDITIONAL (lanes past the row end re-read column 0 and their values are masked
//      out): a per-lane `if (j < n)` around a load becomes a branch with s_waitcnt vmcnt(0) right behind it, i.e. one
//      1-KB load in flight per wave.  All RIF rows' loads are issued before any is used. ----
template <typename T, int NQ, int RIF>
__device__ __forceinline__ void setup_colmax(const T* __restrict__ Q, const int n, T* __restrict__ red) {
    const int row0 = 0, row1 = n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T cm[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) cm[q][e] = T(0);
    const int nq = (n + 255) / 256;               // column quads of 256 that exist (uniform)
    int jq[NQ]; bool okq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int j = (lane + 64 * q) * 4; okq[q] = j < n; jq[q] = okq[q] ? j : 0; }
    for (int i0 = row0 + w; i0 < row1; i0 += RIF * LQP_NW) {
        V4<T> v[RIF][NQ];
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
            const int i = i0 + rr * LQP_NW;
            const T* qr = Q + (size_t)(i < row1 ? i : i0) * n;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nq) v[rr][q] = *(const V4<T>*)(qr + jq[q]);
        }
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q < nq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cm[q][e] = tmax(cm[q][e], okq[q] ? tabs(v[rr][q].v[e]) : T(0));
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int j = (lane + 64 * q) * 4;
        if (j < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(size_t)w * n + j + e] = cm[q][e];
        }
    }
}
// Qs = (D_i Q_ij) D_j (and the top-left KKT block when with_m); returns this thread's share of ||Qs||_F^2
template <typename T, int NQ, int RIF>
__device__ __forceinline__ T setup_scale(const T* __restrict__ Q, const int n, const T* __restrict__ d, T* __restrict__ Qw,
                                         const int ldq, T* __restrict__ Mw, const int Np, const bool with_m) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nq = (n + 255) / 256;
    int jq[NQ]; bool okq[NQ]; V4<T> dj[NQ];
    T fro2 = T(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int j = (lane + 64 * q) * 4;
        okq[q] = j < n; jq[q] = okq[q] ? j : 0;
        dj[q] = *(const V4<T>*)(d + jq[q]);
    }
    for (int i0 = w; i0 < n; i0 += RIF * LQP_NW) {
        V4<T> v[RIF][NQ];
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
            const int i = i0 + rr * LQP_NW;
            const T* qr = Q + (size_t)(i < n ? i : i0) * n;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nq) v[rr][q] = *(const V4<T>*)(qr + jq[q]);
        }
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
            const int i = i0 + rr * LQP_NW;
            if (i < n) {
                T* qo = Qw + (size_t)i * ldq;
                T* mo = Mw + (size_t)i * Np;
                const T di = d[i];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q < nq && okq[q]) {
                        V4<T> o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o.v[e] = (di * v[rr][q].v[e]) * dj[q].v[e]; fro2 += o.v[e] * o.v[e]; }
                        if (Qw) *(V4<T>*)(qo + jq[q]) = o;
                        if (with_m) *(V4<T>*)(mo + jq[q]) = o;
                    }
                }
            }
        }
    }
    return fro2;
}

// The same two passes for rows that are not 16-B pieces (n % 4 != 0, e.g. the hard distribution's n = 250): NC column groups
// of 64 per lane, one element each, RIF rows in flight; unconditional loads at clamped columns as above.  (The plain loop
// with a per-lane `if (j < n)` had one 512-B load in flight per wave: 64 MB in 33 us at n = 250 in float64.)
template <typename T, int NC, int RIF>
__device__ __forceinline__ void setup_colmax_s(const T* __restrict__ Q, const int n, T* __restrict__ red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nq = (n + 63) / 64;
    T cm[NC];
    int jq[NC]; bool okq[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) { const int j = lane + 64 * q; okq[q] = j < n; jq[q] = okq[q] ? j : 0; cm[q] = T(0); }
    for (int i0 = w; i0 < n; i0 += RIF * LQP_NW) {
        T v[RIF][NC];
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
            const int i = i0 + rr * LQP_NW;
            const T* qr = Q + (size_t)(i < n ? i : i0) * n;
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (q < nq) v[rr][q] = qr[jq[q]];
        }
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr)
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (q < nq) cm[q] = tmax(cm[q], okq[q] ? tabs(v[rr][q]) : T(0));
    }
#pragma unroll
    for (int q = 0; q < NC; ++q)
        if (okq[q]) red[(size_t)w * n + lane + 64 * q] = cm[q];
}
template <typename T, int NC, int RIF>
__device__ __forceinline__ T setup_scale_s(const T* __restrict__ Q, const int n, const T* __restrict__ d, T* __restrict__ Qw,
                                           const int ldq, T* __restrict__ Mw, const int Np, const bool with_m) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nq = (n + 63) / 64;
    int jq[NC]; bool okq[NC]; T dj[NC];
    T fro2 = T(0);
#pragma unroll
    for (int q = 0; q < NC; ++q) { const int j = lane + 64 * q; okq[q] = j < n; jq[q] = okq[q] ? j : 0; dj[q] = d[jq[q]]; }
    for (int i0 = w; i0 < n; i0 += RIF * LQP_NW) {
        T v[RIF][NC];
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
            const int i = i0 + rr * LQP_NW;
            const T* qr = Q + (size_t)(i < n ? i : i0) * n;
#pragma unroll
            for (int q = 0; q < NC; ++q)
                if (q < nq) v[rr][q] = qr[jq[q]];
        }
#pragma unroll
        for (int rr = 0; rr < RIF; ++rr) {
            const int i = i0 + rr * LQP_NW;
            if (i < n) {
                T* qo = Qw + (size_t)i * ldq;
                T* mo = Mw + (size_t)i * Np;
                const T di = d[i];
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    if (q < nq && okq[q]) {
                        const T o = (di * v[rr][q]) * dj[q];
                        fro2 += o * o;
                        if (Qw) qo[jq[q]] = o;
                        if (with_m) mo[jq[q]] = o;
                    }
                }
            }
        }
    }
    return fro2;
}

// scratch of k_spd_prep for problem b (the Qs area: unused while the scaled matrix is not stored):
// [SPD_NP][64 Ks] column maxima | [SPD_NP] ints: symmetry verdicts
template <typename T>
__device__ __forceinline__ T* prep_scratch(const FwdParams<T>& P, const int b) { return P.Qs + (size_t)b * P.n * P.ldq; }

// The auto-scaling vector (reference :163-175) from the column maxima of |Q|: red[ww * n + j], ww < nred (LDS; reused as the
// sort buffer) -> d[0..n) (LDS).  Zero guard, D = sqrt(1 / max), beta from the 10 % / 90 % quantiles of D, the blend with the
// mean.  A workgroup of NT threads (k_fwd_setup: 1024; the resident sweep, which takes the maxima from its own pass over Q:
// 512 -- for n <= 512 every thread holds the same element in both and the sums are taken in the same order: the same bits).
// sel: 8 elements, scratch: >= NT / 64 (LDS).
template <typename T, int NT>
__device__ __forceinline__ void wg_scaling_vector(const FwdParams<T>& P, const int b, T* __restrict__ red, const int nred,
                                                  T* __restrict__ d, T* __restrict__ sel, T* __restrict__ scratch) {
    const int tid = threadIdx.x, n = P.n;
    T part = T(0);
    for (int j = tid; j < n; j += NT) {
        T v = red[j];
        for (int ww = 1; ww < nred; ++ww) v = tmax(v, red[(size_t)ww * n + j]);
        d[j] = v;
        part += v;
    }
    // ---- zero guard (:164-168) ----
    const T mean_norm = wg_sum_nw<NT / 64>(part, scratch) / T(n);
    const T floor_v = tmax(mean_norm, T(1e-6));
    part = T(0);
    for (int j = tid; j < n; j += NT) {
        T v = d[j];
        if (v <= T(0)) v = tmax(v, floor_v);
        v = tsqrt(T(1) / v);          // D = sqrt(1 / Q_norm) (:170)
        d[j] = v;
        part += v;
    }
    const T dmean = wg_sum_nw<NT / 64>(part, scratch) / T(n);     // also makes d[] visible
    // ---- beta = 1 - q10(D) / q90(D), linear-interpolated quantiles (:171-174) ----
    T beta = P.beta_mode == 2 ? P.beta_in[b] : P.beta_value;
    if (P.beta_mode == 0) {
        const T pos0 = T(0.10) * T(n - 1), pos1 = T(0.90) * T(n - 1);
        const int lo0 = (int)tfloor(pos0), hi0 = (int)tceil(pos0);
        const int lo1 = (int)tfloor(pos1), hi1 = (int)tceil(pos1);
        // the four order statistics behind the two quantiles: bitonic sort of D, padded with +inf to a power of two.
        // One element per thread while that fits: partners closer than 64 are reached by a wave shuffle, only the
        // far ones (6 of the 45 rounds at n = 500) go through LDS and a barrier.  (Counting every element's rank
        // costs n^2 comparisons: 10 us at n = 500 even over all 1024 threads; every round through LDS: 12 us.)
        T* sb = red;                                      // the column maxima are no longer needed
        int N2 = 1;
        while (N2 < n) N2 <<= 1;
        __syncthreads();
        if (N2 <= NT) {
            T v = tid < n ? d[tid] : T(INFINITY);
            for (int k = 2; k <= N2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    T other;
                    if (j >= 64) {
                        if (tid < N2) sb[tid] = v;
                        __syncthreads();
                        other = tid < N2 ? sb[tid ^ j] : v;
                        __syncthreads();
                    } else {
                        other = __shfl_xor(v, j);
                    }
                    const bool keep_min = ((tid & j) == 0) == ((tid & k) == 0);
                    v = keep_min ? tmin(v, other) : tmax(v, other);
                }
            }
            if (tid < N2) sb[tid] = v;
            __syncthreads();
        } else {
            for (int i = tid; i < N2; i += NT) sb[i] = i < n ? d[i] : T(INFINITY);
            __syncthreads();
            for (int k = 2; k <= N2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (N2 >> 1); t += NT) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                        const T x0 = sb[i], x1 = sb[l];
                        if ((x0 > x1) == ((i & k) == 0)) { sb[i] = x1; sb[l] = x0; }
                    }
                    __syncthreads();
                }
            }
        }
        if (tid == 0) { sel[0] = sb[lo0]; sel[1] = sb[hi0]; sel[2] = sb[lo1]; sel[3] = sb[hi1]; }
        __syncthreads();
        const T w0 = pos0 - tfloor(pos0), w1 = pos1 - tfloor(pos1);
        // torch lerp: w < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
        const T q10 = (w0 < T(0.5)) ? sel[0] + w0 * (sel[1] - sel[0]) : sel[1] - (sel[1] - sel[0]) * (T(1) - w0);
        const T q90 = (w1 < T(0.5)) ? sel[2] + w1 * (sel[3] - sel[2]) : sel[3] - (sel[3] - sel[2]) * (T(1) - w1);
        beta = T(1) - q10 / q90;
    }
    __syncthreads();
    for (int j = tid; j < n; j += NT) {
        const T v = (T(1) - beta) * d[j] + beta * dmean;     // (:175)
        d[j] = v;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_fwd_setup(const FwdParams<T> P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T* red = (T*)smem;
    T* d = red + (size_t)round_up(setup_slabs(n) * n, 8);
    T* sel = d + round_up(n, 8);
    T* scratch = sel + 8;
    const T* Q = P.Q + (size_t)b * n * n;
    const T* p = P.p + (size_t)b * n;
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T* scal = P.scal + (size_t)b * SC_WORDS;
    if (tid == 0) P.info[b] = 0;
    // status block + counter ring start from zero.  Nobody else in this launch touches them (the bound flags are per
    // problem, see below), every later launch is ordered behind this one: no separate fill launch (3.6 us + a boundary).
    if (b == 0)
        for (int i = tid; i < P.zero_words; i += LQP_NT) P.status[i] = 0;
    unsigned long long tst = clock64();
#define SETUP_STAMP(i) do { if (P.dbg_setup && tid == 0) { const unsigned long long t_ = clock64(); P.dbg_setup[(size_t)b * 8 + (i)] = t_ - tst; tst = t_; } } while (0)
    if (P.dnx) {                           // granules of the dense LU-tier loop (lqp_dense.hpp): tags start from zero
        unsigned long long* dq = P.dnx + (size_t)b * P.dnx_words;
        for (int i = tid; i < P.dnx_words; i += LQP_NT) dq[i] = 0ull;
    }
    if (P.xchg) {                          // exchange granules of the two-workgroup loop: tags start from zero
        unsigned long long* xq = P.xchg + (size_t)b * XCHG_WORDS;
        for (int i = tid; i < XCHG_WORDS; i += LQP_NT) xq[i] = 0ull;
        if (tid < XCHG_TAIL) P.xchg[(size_t)P.B * XCHG_WORDS + XCHG_TAIL * b + tid] = 0ull;      // step granules of the resident sweep, XCD announcements
    }

    // the small vectors are requested now and used after the pass over Q (n <= 1024: one element per thread; a load
    // issued where it is needed costs a full memory latency each time, ~2 us, with nothing to hide it behind)
    const T* lb = P.lb + (size_t)b * n;
    const T* ub = P.ub + (size_t)b * n;
    const T p0 = tid < n ? p[tid] : T(0), lb0 = tid < n ? lb[tid] : T(0), ub0 = tid < n ? ub[tid] : T(0);
    const T a0 = (m > 0 && tid < n) ? P.A[(size_t)b * m * n + tid] : T(0);          // first equality row
    SETUP_STAMP(0);
    const T* Qs = Q;
    int ldq = n;
    T fro2 = T(0);
    bool q_in_m = false;          // top-left KKT block already written by the scaling pass
    // prep_fused == 3: the resident sweep makes the pass over Q itself (column maxima, symmetry verdict, tiles straight into
    // its registers) and with it everything that needs D: the vector, ps, As / E / bs, lbs / ubs (wg_deferred_*).  What is
    // left here: the zeroes, ||p||_inf, a given rho, the bound flags.
    const bool defer = P.prep_fused == 3;
    if (defer) {
    } else if (P.scale) {
        // ---- column max of |Q| (:163): wave w sweeps rows w, w+16, ...; 16 B per lane per load ----
        const bool qvec = (n % 4 == 0) && n <= 1024 && ((((uintptr_t)Q) % sizeof(V4<T>)) == 0);
        int nred = LQP_NW;
        if (P.prep_fused) {
            // k_spd_prep has been over Q: the maxima of its SPD_NP workgroups' shares, and their symmetry verdicts
            const T* cmp = prep_scratch(P, b);
            for (int j = tid; j < n; j += LQP_NT) red[j] = tmax(cmp[j], cmp[P.Ks * LQP_NB + j]);
            if (tid == 0 && (((const int*)(cmp + 2 * P.Ks * LQP_NB))[0] | ((const int*)(cmp + 2 * P.Ks * LQP_NB))[1]) != 0)
                P.info[b] = P.Ks * 64 + 2;           // not symmetric: the LU path takes it (status word: set by the loop kernel)
            nred = 1;
        } else if (qvec) {
            // (several workgroups per QP for this pass: no faster -- 128 MB in 38 us either way, the HBM rate)
            // (up to 256 columns: one 16-B piece per lane and row -- more rows in flight instead; float64 at n = 250 had ONE 2-KB row in
            //  flight per wave: 64 MB in 33 us)
            if (n <= 256) setup_colmax<T, 1, sizeof(T) == 8 ? 4 : 8>(Q, n, red);
            else setup_colmax<T, 4, sizeof(T) == 8 ? 1 : 4>(Q, n, red);        // (8 rows in flight with 2 column quads: no faster; float64: 4 rows in flight = 128 VGPRs of loads, 315 spilled registers)
        } else if (n <= 256) {
            setup_colmax_s<T, 4, 4>(Q, n, red);
        } else if (n <= 512) {
            setup_colmax_s<T, 8, 2>(Q, n, red);
        } else {
            // (also the HBM-resident tier, 1024 < n <= 2048: two passes of 1024 columns, the 16 waves merging their maxima
            //  into setup_slabs(n) = 4 slabs in 4 rounds; up to 1024 columns: one pass, one slab per wave, no round trip)
            nred = setup_slabs(n);
            for (int c0 = 0; c0 < n; c0 += 1024) {
                T cm[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) cm[q] = T(0);
                // (unconditional loads, eight in flight per wave -- lanes past the row end re-read its first column and are masked: a guarded load
                //  is a branch with a full wait behind it, one 256-B load in flight per wave: 9 MB in 0.6 ms at n = 1500)
                for (int i = w; i < n; i += LQP_NW) {
                    const T* qr = Q + (size_t)i * n + c0;
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int h = 0; h < 16; h += 8) {
                            T v[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) { const int j = lane + 64 * (h + q); v[q] = qr[(c0 + j < n) ? j : 0]; }
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int j = lane + 64 * (h + q);
                                cm[h + q] = tmax(cm[h + q], (c0 + j < n) ? tabs(v[q]) : T(0));
                            }
                        }
                    } else {                                      // (float64: the unconditional form spills registers in this kernel)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int j = lane + 64 * q;
                            if (c0 + j < n) cm[q] = tmax(cm[q], tabs(qr[j]));
                        }
                    }
                }
                for (int round = 0; round < LQP_NW / nred; ++round) {
                    if (w / nred == round) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int j = c0 + lane + 64 * q;
                            if (j < n) {
                                T* slot = red + (size_t)(w % nred) * n + j;
                                *slot = round == 0 ? cm[q] : tmax(*slot, cm[q]);
                            }
                        }
                    }
                    if (nred < LQP_NW) __syncthreads();
                }
            }
        }
        __syncthreads();
        SETUP_STAMP(1);
        wg_scaling_vector<T, LQP_NT>(P, b, red, nred, d, sel, scratch);
        SETUP_STAMP(2);
        for (int j = tid; j < n; j += LQP_NT) V.D[j] = d[j];
        SETUP_STAMP(3);
        // ---- Qs = (D_i Q_ij) D_j, its Frobenius norm (:176, :201), and the top-left KKT block ----
        ldq = P.ldq;
        // (symmetric path: the scaled matrix is not stored, its readers scale Q as they load it)
        T* Qw = P.qs_lazy ? nullptr : P.Qs + (size_t)b * n * ldq;
        T* Mw = P.M + (size_t)b * Np * Np;
        if (P.qs_lazy && P.spd && (P.rho_late || (P.prep_fused && P.rho_mode != 0) || P.prep_fused == 2)) {
            // nothing to store here and the norm (when rho is derived from it) is taken by k_spd_begin / the resident
            // sweep: no second pass over Q
        } else if (qvec) {
            // (every thread meets its elements in the same order whatever the shape of the pass: the norm keeps its bits)
            if (n <= 256) fro2 += setup_scale<T, 1, sizeof(T) == 8 ? 4 : 8>(Q, n, d, Qw, ldq, Mw, Np, !P.spd);
            else fro2 += setup_scale<T, 4, sizeof(T) == 8 ? 1 : 4>(Q, n, d, Qw, ldq, Mw, Np, !P.spd);
        } else if (n <= 256) {
            fro2 += setup_scale_s<T, 4, 4>(Q, n, d, Qw, ldq, Mw, Np, !P.spd);
        } else if (n <= 512) {
            fro2 += setup_scale_s<T, 8, 2>(Q, n, d, Qw, ldq, Mw, Np, !P.spd);
        } else {
            for (int i = w; i < n; i += LQP_NW) {
                const T* qr = Q + (size_t)i * n;
                T* qo = Qw + (size_t)i * ldq;
                T* mo = Mw + (size_t)i * Np;
                const T di = d[i];
                // (eight / four loads in flight; every lane meets its elements in the order it always did: the norm keeps its bits)
                constexpr int SQ = sizeof(T) == 8 ? 4 : 8;
                for (int j0 = lane; j0 < n; j0 += 64 * SQ) {
                    T qv[SQ];
#pragma unroll
                    for (int q = 0; q < SQ; ++q) { const int j = j0 + 64 * q; qv[q] = qr[j < n ? j : lane]; }
#pragma unroll
                    for (int q = 0; q < SQ; ++q) {
                        const int j = j0 + 64 * q;
                        if (j < n) {
                            const T v = (di * qv[q]) * d[j];
                            if (Qw) qo[j] = v;
                            if (!P.spd) mo[j] = v;
                            fro2 += v * v;
                        }
                    }
                }
            }
        }
        q_in_m = true;
        Qs = Qw;
        for (int i = tid; i < n; i += LQP_NT) V.ps[i] = d[i] * (i == tid ? p0 : p[i]);       // (:177)
    } else {
        for (int i = tid; i < n; i += LQP_NT) { V.D[i] = T(1); V.ps[i] = i == tid ? p0 : p[i]; }
        if (P.rho_mode == 0 && !(P.rho_late && P.spd)) {
            for (int i = w; i < n; i += LQP_NW) {
                const T* qr = Q + (size_t)i * n;
                for (int j = lane; j < n; j += 64) { const T v = qr[j]; fro2 += v * v; }
            }
        }
    }
    SETUP_STAMP(4);
    // ---- ||p||_inf on the unscaled p (:127) ----
    {
        T pm = tabs(p0);
        for (int i = tid + LQP_NT; i < n; i += LQP_NT) pm = tmax(pm, tabs(p[i]));
        pm = wg_max(pm, scratch);
        if (tid == 0) scal[SC_PNORM] = pm;
    }
    // ---- rho (:140, :157-158, :200-203) ----
    T rho;
    if (P.rho_mode == 0) {
        const T fro = tsqrt(wg_sum(fro2, scratch));
        rho = fro / T(sqrt((double)n));
        rho = tmin(tmax(rho, P.rho_min), P.rho_max);
    } else if (P.rho_mode == 1) {
        rho = P.rho_value;
    } else {
        rho = P.rho_in[b];
    }
    if (tid == 0) { scal[SC_RHO] = rho; scal[SC_RATIO] = T(1); scal[SC_WANTS] = T(0); }

    // ---- equality block: A D, row normalisation E (:179-190) ----
    if (m > 0 && !defer) {
        const T* A = P.A + (size_t)b * m * n;
        const T* bb = P.b + (size_t)b * m;
        if (P.scale && m >= 4 && m <= 64 && setup_slabs(n) * n >= 64) {
            // a row per wave (rows w, w + 16, ...), its norm a wave maximum, ONE barrier for the mean of the norms (a row per
            // round of the whole workgroup was three barriers a row: 42 us at m = 16; kept below four rows, where a row on one wave is slower).  The maxima are exact and the mean adds
            // them in row order as it always did: same bits.
            T* rn = red;                                   // (the column maxima are done with)
            for (int r = w; r < m; r += LQP_NW) {
                T am = T(0);
                for (int j = lane; j < n; j += 64) {
                    const T v = ((r == 0 && j == tid) ? a0 : A[(size_t)r * n + j]) * d[j];
                    am = tmax(am, tabs(v));
                }
                am = wave_max(am);
                if (lane == 0) rn[r] = am;
            }
            __syncthreads();
            T esum = T(0);
            for (int r = 0; r < m; ++r) esum += rn[r];
            const T floor_a = tmax(esum / T(m), T(1e-6));
            for (int r = w; r < m; r += LQP_NW) {
                T an = rn[r];
                if (an <= T(0)) an = tmax(an, floor_a);
                const T e = T(1) / an;
                for (int j = lane; j < n; j += 64) {
                    const T v = ((r == 0 && j == tid) ? a0 : A[(size_t)r * n + j]) * d[j];
                    V.As[(size_t)r * n + j] = e * v;
                }
                if (lane == 0) { V.E[r] = e; V.bs[r] = e * bb[r]; }
            }
            __syncthreads();
        } else if (P.scale) {
            T esum = T(0);
            for (int r = 0; r < m; ++r) {
                T am = T(0);
                for (int j = tid; j < n; j += LQP_NT) {
                    const T v = ((r == 0 && j == tid) ? a0 : A[(size_t)r * n + j]) * d[j];
                    V.As[(size_t)r * n + j] = v;
                    am = tmax(am, tabs(v));
                }
                am = wg_max(am, scratch);
                if (tid == 0) V.E[r] = am;      // row norm for now
                esum += am;
            }
            __syncthreads();
            const T floor_a = tmax(esum / T(m), T(1e-6));
            for (int r = 0; r < m; ++r) {
                T an = V.E[r];
                if (an <= T(0)) an = tmax(an, floor_a);
                const T e = T(1) / an;
                for (int j = tid; j < n; j += LQP_NT) V.As[(size_t)r * n + j] = e * V.As[(size_t)r * n + j];
                __syncthreads();
                if (tid == 0) { V.E[r] = e; V.bs[r] = e * bb[r]; }
            }
        } else {
            for (int t = tid; t < m * n; t += LQP_NT) V.As[t] = A[t];
            for (int r = tid; r < m; r += LQP_NT) { V.E[r] = T(1); V.bs[r] = bb[r]; }
        }
    }
    SETUP_STAMP(5);
    // ---- bounds (:192-194) and state ----
    // any finite bound in the batch?  (:129-130: a HOST decision in the reference -- it selects the rho = 0 shortcut and
    // the clamps.  Here the clamps always run (an infinite bound is an exact no-op) and the answer goes back with the
    // status: one word per problem here, OR-ed over the batch by the forward's last kernel (fwd_finish); the host compares
    // it with what it assumed when it chose the schedule.)
    {
        bool flb = false, fub = false;
        for (int i = tid; i < n; i += LQP_NT) {
            flb |= (i == tid ? lb0 : lb[i]) > -T(INFINITY);
            fub |= (i == tid ? ub0 : ub[i]) < T(INFINITY);
        }
        const int wg_lb = __syncthreads_or(flb ? 1 : 0), wg_ub = __syncthreads_or(fub ? 1 : 0);
        if (tid == 0) P.bflags[b] = (wg_lb ? RP_LB : 0) | (wg_ub ? RP_UB : 0);
    }
    for (int i = tid; i < n; i += LQP_NT) {
        if (!defer) {
            const T di = P.scale ? V.D[i] : T(1);                // (:192-194; +-inf / D stays +-inf)
            V.lbs[i] = (i == tid ? lb0 : lb[i]) / di;
            V.ubs[i] = (i == tid ? ub0 : ub[i]) / di;
        }
        V.z[i] = T(0); V.u[i] = T(0); V.x[i] = T(0);
    }
    for (int r = tid; r < m; r += LQP_NT) V.nu[r] = T(0);
    __syncthreads();
    SETUP_STAMP(6);
    if (!P.spd) assemble_kkt_rows(P, b, Qs, ldq, V, rho, !q_in_m);
}

// ---------------------------------------------------------------------------
// LU and pack kernels (gated: *gate == 0 -> nothing to do)
// ---------------------------------------------------------------------------
template <typename T, int PB, bool MFMA, int NT>
__global__ __launch_bounds__(NT) void k_lu_factor(T* __restrict__ Mall, const int N, const int ld,
                                                      const size_t mstride, int* __restrict__ piv,
                                                      const int pstride, int* __restrict__ info,
                                                      const int* __restrict__ gate,
                                                      unsigned long long* __restrict__ dbg,
                                                      const int* __restrict__ Nvec) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) info[b] = 0;
    __syncthreads();
    const int Nb = Nvec ? Nvec[b] : N;            // per-problem size (reduced backward systems)
    wg_lu_factor<T, PB, MFMA, NT>(Mall + (size_t)b * mstride, Nb, ld, piv + (size_t)b * pstride, info + b, smem,
                              dbg ? dbg + (size_t)b * 4 : nullptr);
}

// two workgroups per matrix (lqp_lu2.hpp; N <= 512, 2 B workgroups resident): workgroups b and b + B share matrix b
template <typename T, int PB>
__global__ __launch_bounds__(LU2_NT) void k_lu_factor2(T* __restrict__ Mall, const int N, const int ld, const size_t mstride,
                                                       int* __restrict__ piv, const int pstride, int* __restrict__ info,
                                                       const int* __restrict__ gate, const int* __restrict__ Nvec,
                                                       unsigned long long* __restrict__ scr, const size_t scr_stride,
                                                       const unsigned int epoch, unsigned long long* __restrict__ dbg,
                                                       const int B, const int xlocal_ok) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    int b, me;
    if (!shared_map((int)blockIdx.x, B, 2, b, me)) return;
    const int Nb = Nvec ? Nvec[b] : N;
    if (threadIdx.x == 0 && me == 0) info[b] = 0;
    __syncthreads();
    wg_lu_factor2<T, PB>(Mall + (size_t)b * mstride, Nb, ld, piv + (size_t)b * pstride, info + b, smem, me,
                         scr + (size_t)b * scr_stride, epoch, xlocal_ok != 0, (dbg && me == 0) ? dbg + (size_t)b * 16 : nullptr);
}

// 1024 < N <= 2048: two panel rows per thread; to 4096 in float32: four (lqp_lu_big.hpp)
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_lu_factor_big(T* __restrict__ Mall, const int N, const int ld,
                                                          const size_t mstride, int* __restrict__ piv,
                                                          const int pstride, int* __restrict__ info,
                                                          const int* __restrict__ gate, const int* __restrict__ Nvec) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) info[b] = 0;
    __syncthreads();
    // (N is the launch's size, the LDS layout follows it: a smaller Nvec[b] runs on the same form)
    if constexpr (sizeof(T) == 4) {
        if (N > 2048) {           // 2048 < N <= 4096: four rows per thread, panels of 4 columns (2 * 4 * N floats of LDS)
            wg_lu_factor_big<T, 4, true, 4>(Mall + (size_t)b * mstride, Nvec ? Nvec[b] : N, ld, piv + (size_t)b * pstride, info + b, smem);
            return;
        }
    }
    wg_lu_factor_big<T, lu_big_panel<T>(), true, 2>(Mall + (size_t)b * mstride, Nvec ? Nvec[b] : N, ld,
                                                               piv + (size_t)b * pstride, info + b, smem);
}

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_pack(const T* __restrict__ LUall, const int N, const int ld,
                                                 const size_t mstride, const int* __restrict__ piv,
                                                 const int pstride, T* __restrict__ packed,
                                                 const size_t pkstride, int* __restrict__ dest,
                                                 const int dstride, const int vec_ok,
                                                 const int* __restrict__ gate, const int* __restrict__ Nvec) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    const int b = blockIdx.x;
    // gridDim.y == 2: two workgroups per factor (L half / U half) -- at B <= 128 half the CUs are idle
    const int part = gridDim.y == 2 ? 1 + (int)blockIdx.y : 0;
    wg_pack_factor<T>(LUall + (size_t)b * mstride, Nvec ? Nvec[b] : N, ld, piv + (size_t)b * pstride,
                      packed + (size_t)b * pkstride, dest + (size_t)b * dstride, smem, vec_ok != 0, part);
}

// ---------------------------------------------------------------------------
// ADMM loop: iterations [it0, it1) for every problem (:235-313)
// LDS: v[Np] | tmp[64] | z[n] | u[n] | ps[n] | lb[n] | ub[n] | D[n] | bs[m] | red[NW*8] | dest[Np] (int)
// ---------------------------------------------------------------------------
template <typename T> __host__ __device__ inline int loop_lds_bytes(int n, int m, int Np, bool resident) {
    return (resident ? LQP_RLDS * LQP_BLK * (int)sizeof(T) : 0) +
           (Np + 64 + 6 * n + m + LQP_NW * 8 + 8) * (int)sizeof(T) + Np * 4;
}
// residency applies when the stream is long enough and the ring can stay cyclic over the streamed tail
template <int NT> __host__ __device__ inline bool loop_resident_ok(int K, size_t elem) {
    const int S = K * (K + 1), R0 = resident_total<NT>();
    return elem == 4 && S >= R0 + LQP_PF && ((S - R0) % LQP_PF) == 0;
}

template <typename T, int NV, int NWV = LQP_NW>
__device__ __forceinline__ void wg_max_n(T (&v)[NV], T* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = wave_max(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[w * 8 + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        T r = red[q];
#pragma unroll
        for (int ww = 1; ww < NWV; ++ww) r = tmax(r, red[ww * 8 + q]);
        v[q] = r;
    }
}

// bounded spin on a device-scope counter (persistent mode only)
__device__ __forceinline__ bool grid_wait(unsigned int* ctr, const unsigned int target, int* status) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {    // 2 s: give up
                __hip_atomic_store(status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return true;
}

// ---------------------------------------------------------------------------
// symmetric-inverse path (lqp_spd.hpp): factorisation kernels
// ---------------------------------------------------------------------------
// standalone SPD inverse (test / utility entry lqp_spd_inverse_batched): dense (B,n,n) in, dense inverse out
template <int SEL = 1>          // 1: n <= 512 (wg_spd_sweep), 2: above (wg_spd_sweep_big): one sweep per instance, as k_spd_inverse
__global__ __launch_bounds__(LQP_NT) void k_spd_inverse_dense(const float* __restrict__ Kin, float* __restrict__ out,
                                                              float* __restrict__ Hs_all, int* __restrict__ info,
                                                              const int n, const int Ks, float* __restrict__ Yg_all) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* Hs = Hs_all + (size_t)b * sym_blocks(Ks) * LQP_BLK;
    if (tid == 0) info[b] = 0;
    wg_sym_init(Hs, Kin + (size_t)b * n * n, n, n, Ks, 0.f);
    __syncthreads();
    if constexpr (SEL == 2) wg_spd_sweep_big(Hs, Ks, info + b, smem, Yg_all + (size_t)b * (Ks - 1) * LQP_BLK);
    else wg_spd_sweep(Hs, Ks, info + b, smem);
    __syncthreads();
    float* o = out + (size_t)b * n * n;
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    for (int j = 0; j < Ks; ++j)
        for (int i = j; i < Ks; ++i) {
            const V4<float> h = *(const V4<float>*)(Hs + (size_t)sym_idx(i, j, Ks) * LQP_BLK + tid * 4);
            const int gr = i * 64 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gc = j * 64 + c4 + e;
                if (gr < n && gc < n) {
                    o[(size_t)gr * n + gc] = -h.v[e];
                    if (i != j) o[(size_t)gc * n + gr] = -h.v[e];
                }
            }
        }
}

// Si = S^-1 (S = Sm is symmetric positive definite, m <= 16: Gauss-Jordan without pivoting, Sm is destroyed): one thread per entry
// of [Sm | Si] -- 2 m^2 <= 512 of them -- and three barriers per column.  One thread used to walk through all of it: ~4 m^3
// dependent LDS round trips, ~0.25 ms per forward at m = 16 (inside the loop kernel's prologue).  The same operations on the same
// values per entry: the same bits.  Every thread of the workgroup calls; returns 1 when a pivot was not positive.
template <typename T, typename Barrier>
__device__ __forceinline__ int wg_gj_inverse_spd(T* __restrict__ Sm, T* __restrict__ Si, const int m, Barrier barrier) {
    const int tid = threadIdx.x, m2 = 2 * m;
    for (int i = tid; i < m * m; i += blockDim.x) Si[i] = (i / m == i % m) ? T(1) : T(0);
    const int rr = tid / m2, jj = tid - rr * m2;
    const bool mine = tid < m * m2;
    T* const ent = !mine ? Sm : (jj < m ? Sm + rr * m + jj : Si + rr * m + (jj - m));
    int bad = 0;
    barrier();
    for (int c = 0; c < m; ++c) {
        const T d = Sm[c * m + c];
        const T f = mine ? Sm[rr * m + c] : T(0);
        T* const pivot_row = jj < m ? Sm + c * m + jj : Si + c * m + (jj - m);
        if (!(d > T(0))) bad = 1;
        const T inv = d > T(0) ? T(1) / d : T(0);
        barrier();
        if (mine && rr == c) *ent *= inv;
        barrier();
        if (mine && rr != c) *ent -= f * *pivot_row;
        barrier();
    }
    return bad;
}

// equality rows: G = K^-1 A^T, S = A G, T = G S^-1, c = T b, s0 = S^-1 b, Hs += T G^T  (m <= SPD_MAXM)
// LDS: v | ylds | part[NW][Nps] | G[m][Nps] | Tl[m][Nps] | Sm[m*m] | Si[m*m]
__host__ __device__ inline int eqc_lds_bytes(int m, int Ks) {
    const int Nps = Ks * LQP_NB;
    return (Nps + sym_blocks(Ks) * 64 + LQP_NW * Nps + 2 * m * Nps + 2 * m * m + 8) * 4;
}
__device__ __forceinline__ void wg_eq_correct(const FwdParams<float>& P, const int b, char* smem) {
    const int n = P.n, m = P.m, Ks = P.Ks, Nps = Ks * LQP_NB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* v = (float*)smem;
    float* ylds = v + Nps;
    float* part = ylds + sym_blocks(Ks) * 64;
    float* G = part + (size_t)LQP_NW * Nps;
    float* Tl = G + (size_t)m * Nps;
    float* Sm = Tl + (size_t)m * Nps;
    float* Si = Sm + m * m;
    VecView<float> V(P.vecs + (size_t)b * P.vstride, n, m);
    float* Hs = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
    BlockStream<float, LQP_NT> st;
    SymResident rr;
    // ---- G[q] = K^-1 a_q = -(Hs a_q) ----
    for (int q = 0; q < m; ++q) {
        for (int i = tid; i < Nps; i += LQP_NT) v[i] = i < n ? V.As[(size_t)q * n + i] : 0.f;
        __syncthreads();
        wg_sym_gemv<false>(st, rr, nullptr, 0, Hs, Ks, Nps, v, ylds, part);
        __syncthreads();
        for (int e = tid; e < Nps; e += LQP_NT) G[(size_t)q * Nps + e] = -sym_combine(e, Ks, Nps, ylds, part);
        __syncthreads();
    }
    // ---- S = A G (m x m), one wave per entry ----
    for (int t = w; t < m * m; t += LQP_NW) {
        const int q = t / m, q2 = t - q * m;
        float acc = 0.f;
        for (int i = lane; i < n; i += 64) acc += V.As[(size_t)q * n + i] * G[(size_t)q2 * Nps + i];
        acc = wave_sum(acc);
        if (lane == 0) Sm[t] = acc;
    }
    __syncthreads();
    // ---- S^-1 by Gauss-Jordan (S is SPD: no pivoting), m <= 16 ----
    {
        const int bad = wg_gj_inverse_spd(Sm, Si, m, [] { __syncthreads(); });
        if (bad && tid == 0) { if (P.info[b] == 0) P.info[b] = P.Ks * 64 + 1; P.status[ST_NOTSPD] = 1; }   // A rank deficient: not this path
    }
    __syncthreads();
    // ---- T = G S^-1, c = T b, s0 = S^-1 b ----
    for (int t = tid; t < m * Nps; t += LQP_NT) {
        const int q = t / Nps, e = t - q * Nps;
        float acc = 0.f;
        for (int q2 = 0; q2 < m; ++q2) acc += G[(size_t)q2 * Nps + e] * Si[q2 * m + q];
        Tl[t] = acc;
        if (e < n) V.Tm[(size_t)q * n + e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < n; e += LQP_NT) {
        float acc = 0.f;
        for (int q = 0; q < m; ++q) acc += Tl[(size_t)q * Nps + e] * V.bs[q];
        V.cv[e] = acc;
    }
    for (int q = tid; q < m; q += LQP_NT) {
        float acc = 0.f;
        for (int q2 = 0; q2 < m; ++q2) acc += Si[q * m + q2] * V.bs[q2];
        V.s0[q] = acc;
    }
    // ---- Hs += T G^T on every lower block (8 blocks' loads in flight at a time: one after the other the
    //      36 load -> update -> store round trips were pure latency) ----
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    const int S = sym_blocks(Ks);
    int bi = 0, bj = 0;                                      // block (bi, bj) of stream position s0
    for (int s0 = 0; s0 < S; s0 += 8) {
        V4<float> h[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) h[u] = *(const V4<float>*)(Hs + (size_t)(s0 + u) * LQP_BLK + tid * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < S) {
                for (int q = 0; q < m; ++q) {
                    const float t = Tl[(size_t)q * Nps + bi * 64 + r];
                    const V4<float> g = *(const V4<float>*)(G + (size_t)q * Nps + bj * 64 + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[u].v[e] += t * g.v[e];
                }
                *(V4<float>*)(Hs + (size_t)(s0 + u) * LQP_BLK + tid * 4) = h[u];
                if (++bi == Ks) { ++bj; bi = bj; }
            }
        }
    }
}

// (re)factorisation of the symmetric path for problem b: Hs = -(Qs + rho I)^-1, then the equality correction.
// check_sym: also verify that Qs is symmetric to rounding (the sweep only ever reads its lower triangle); a
// matrix that is not goes to the LU path like one that is not positive definite.
// SEL: 0 = any size (the continuation kernel's in-kernel refactorisation), 1 = n <= 512 only, 2 = above only: one sweep per
// instance of k_spd_inverse -- with both inlined the register allocator spilled 236 registers
template <int SEL = 0>
__device__ __forceinline__ void wg_spd_factor(const FwdParams<float>& P, const int b, const float rho, char* smem,
                                              const bool check_sym) {
    float* Hs = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
    const bool lazy = P.scale && P.qs_lazy;
    const float* Qs = (P.scale && !lazy) ? (P.Qs + (size_t)b * P.n * P.ldq) : (P.Q + (size_t)b * P.n * P.n);
    const int ldq = (P.scale && !lazy) ? P.ldq : P.n;
    const float* dsc = lazy ? VecView<float>(P.vecs + (size_t)b * P.vstride, P.n, P.m).D : nullptr;
    if (check_sym && P.prep_fused && P.Ks <= SPD_MAXK) {
        // First factorisation behind k_spd_prep (the one-workgroup tier: more problems than half the CUs).  The UNSCALED
        // lower blocks are in the packed area already and the symmetry verdict went to the setup kernel: what is left is
        // element-wise -- (D_r v) D_c (the operations of sym_scale4: the same bits as the scaled copy), ||Qs||_F^2 on the way
        // (-> rho = clamp(||Qs||_F / sqrt(n)), :200-203, when rho is not given), rho on the diagonal.  Three passes over Q
        // less per solve (norm, symmetry check, block build: 0.6 of 7.9 ms at B = 1024, n = 500).
        const int n = P.n, Ks = P.Ks, tid = threadIdx.x, r = tid >> 4, c4 = (tid & 15) * 4;
        float* Dl = (float*)smem;                                    // [64 Ks] scaling vector, 1 on the padding
        float* red = Dl + 64 * Ks;
        for (int i = tid; i < 64 * Ks; i += LQP_NT) Dl[i] = i < n ? dsc[i] : 1.f;
        __syncthreads();
        const float* src = Hs + (size_t)(Ks & 1) * sym_blocks(Ks) * LQP_BLK;     // (where k_spd_prep leaves them: spd_half(P, b, Ks & 1))
        float fs = 0.f;
        for (int j = 0; j < Ks; ++j)
            for (int i = j; i < Ks; ++i) {
                const size_t off = (size_t)sym_idx(i, j, Ks) * LQP_BLK + tid * 4;
                V4<float> v = *(const V4<float>*)(src + off);
                const int gr = i * 64 + r, gc = j * 64 + c4;
                const float dr = Dl[gr];
                float t2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v.v[e] = (dr * v.v[e]) * Dl[gc + e];
                    t2 += (gr < n && gc + e < n) ? v.v[e] * v.v[e] : 0.f;      // (the identity on the padding is left out)
                }
                fs += i == j ? t2 : 2.f * t2;
                *(V4<float>*)(Hs + off) = v;
            }
        float rho_here = rho;
        if (P.rho_mode == 0) {
            const float fro = sqrtf(wg_sum(fs, red));
            rho_here = fro / (float)sqrt((double)n);
            rho_here = tmin(tmax(rho_here, P.rho_min), P.rho_max);
            if (tid == 0) P.scal[(size_t)b * SC_WORDS + SC_RHO] = rho_here;
        }
        __syncthreads();
        for (int i = 0; i < Ks; ++i) {                               // rho on the diagonal (own elements: written above by this thread)
            const int gr = i * 64 + r;
            if (gr < n && r >= c4 && r < c4 + 4) Hs[(size_t)sym_idx(i, i, Ks) * LQP_BLK + tid * 4 + (r - c4)] += rho_here;
        }
        __syncthreads();
    } else {
        if (check_sym) {
            float* red = (float*)smem;
            const float asym = wg_sym_asymmetry(Qs, ldq, P.n, P.Ks, red, dsc);
            if (threadIdx.x == 0 && asym > 0.f) { P.info[b] = P.Ks * 64 + 2; P.status[ST_NOTSPD] = 1; }
            __syncthreads();
        }
        wg_sym_init(Hs, Qs, ldq, P.n, P.Ks, rho, dsc);
        __syncthreads();
    }
    // (one sweep for both ways in: the blocks are in place either way)
    if (SEL == 2 || (SEL == 0 && P.Ks > SPD_MAXK)) wg_spd_sweep_big(Hs, P.Ks, P.info + b, smem, P.M + (size_t)b * P.Np * P.Np);   // (M: unused on this path)
    else wg_spd_sweep(Hs, P.Ks, P.info + b, smem, P.dbg ? P.dbg + (size_t)b * 8 : nullptr);
    if (threadIdx.x == 0 && P.info[b] != 0) P.status[ST_NOTSPD] = 1;
    if (P.m > 0) {
        __syncthreads();
        wg_eq_correct(P, b, smem);
    }
}
__host__ __device__ inline int spd_factor_lds_bytes(int m, int Ks) {
    const int a = spd_lds_bytes(Ks > SPD_MAXK ? SPD_MAXK : Ks), c = m > 0 ? eqc_lds_bytes(m, Ks) : 0;
    return a > c ? a : c;
}
template <int SEL = 1>          // 1: n <= 512 (wg_spd_sweep), 2: 512 < n <= 1024 (wg_spd_sweep_big)
__global__ __launch_bounds__(LQP_NT) void k_spd_inverse(const FwdParams<float> P, const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    const int b = blockIdx.x;
    wg_spd_factor<SEL>(P, b, P.scal[(size_t)b * SC_WORDS + SC_RHO], smem, gate == nullptr);
}

// ---- the same factorisation spread over launches: with fewer problems than half the CUs, SPD_NP workgroups share
//      one matrix.  begin (symmetry check, Qs + rho I -> blocks) | one launch per pivot step, out of place between
//      the two halves of the problem's packed area (K(K+1) >= 2 sym_blocks(Ks) blocks) | end (equality correction).
//      The parity of the first buffer is chosen so that the last step writes the half the loop reads. ----
constexpr int SPD_NP = 2;
__device__ __forceinline__ float* spd_half(const FwdParams<float>& P, const int b, const int which) {
    return P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK + (size_t)which * sym_blocks(P.Ks) * LQP_BLK;
}
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ __launch_bounds__(LQP_NT) void k_spd_begin(const FwdParams<float> P, const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    int b, part;
    if (!shared_map((int)blockIdx.x, P.B, SPD_NP, b, part)) return;
    const bool lazy = P.scale && P.qs_lazy;
    const float* Qs = (P.scale && !lazy) ? (P.Qs + (size_t)b * P.n * P.ldq) : (P.Q + (size_t)b * P.n * P.n);
    const float* dsc = lazy ? VecView<float>(P.vecs + (size_t)b * P.vstride, P.n, P.m).D : nullptr;
    // rho_late (first factorisation only): this workgroup's share of ||Qs||_F^2 goes behind the sweep's exchange buffer,
    // the blocks are written without rho
    const bool late = gate == nullptr && P.rho_late;
    float* fro_out = late ? (P.M + (size_t)b * P.Np * P.Np + (size_t)2 * P.Ks * LQP_BLK + part) : nullptr;
    // (K <= 8: the launches after this one go back and forth between the two halves and end in half 0; above: in place)
    const float asym = wg_sym_check_init<SPD_NP>(spd_half(P, b, P.Ks > SPD_MAXK ? 0 : (P.Ks & 1)), Qs, (P.scale && !lazy) ? P.ldq : P.n, P.n, P.Ks,
                                                 late ? 0.f : P.scal[(size_t)b * SC_WORDS + SC_RHO], (float*)smem,
                                                 gate == nullptr, part, dsc, fro_out);
    if (threadIdx.x == 0 && asym > 0.f) { P.info[b] = P.Ks * 64 + 2; P.status[ST_NOTSPD] = 1; }
}
// Above 512 rows with rho = clamp(||Qs||_F / sqrt(n)) (FwdParams::rho_late): k_spd_begin has built the blocks WITHOUT rho and left
// the two workgroups' shares of ||Qs||_F^2 behind the panel area -- the setup kernel then makes one pass over Q (the column maxima)
// instead of two.  Here rho is formed (part 0 + part 1, the order fixed), stored, and added to the K diagonal tiles before the
// sweep's first launch reads them.
template <int LQP_ANY = 0>
__global__ __launch_bounds__(LQP_NT) void k_spd_rho_big(const FwdParams<float> P) {
    const int b = blockIdx.x, tid = threadIdx.x, n = P.n, Ks = P.Ks;
    const float* fr = P.M + (size_t)b * P.Np * P.Np + (size_t)2 * Ks * LQP_BLK;
    float rho = sqrtf(fr[0] + fr[1]) / (float)sqrt((double)n);
    rho = tmin(tmax(rho, P.rho_min), P.rho_max);
    if (tid == 0) P.scal[(size_t)b * SC_WORDS + SC_RHO] = rho;
    float* Hs = spd_half(P, b, 0);
    const int j = tid >> 6, t = tid & 63;
    if (j < Ks && j * 64 + t < n) Hs[(size_t)sym_idx(j, j, Ks) * LQP_BLK + t * 64 + t] += rho;
}
// the first factorisation's k_spd_begin, moved IN FRONT of the setup kernel (FwdParams::prep_fused): the column maxima the
// scaling starts from come out of the same pass over Q that checks its symmetry and builds the blocks -- unscaled; the
// resident sweep scales them as it loads them.  One pass over Q per solve instead of two.
static_assert(SPD_NP == 2, "k_fwd_setup reads the two halves k_spd_prep leaves");
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ __launch_bounds__(LQP_NT) void k_spd_prep(const FwdParams<float> P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    int b, part;
    if (!shared_map((int)blockIdx.x, P.B, SPD_NP, b, part)) return;
    float* sc = prep_scratch(P, b);
    const float asym = wg_sym_prep<SPD_NP>(spd_half(P, b, P.Ks & 1), P.Q + (size_t)b * P.n * P.n, P.n, P.n, P.Ks, (float*)smem,
                                           part, sc + (size_t)part * P.Ks * LQP_NB);
    if (threadIdx.x == 0) ((int*)(sc + (size_t)SPD_NP * P.Ks * LQP_NB))[part] = asym > 0.f ? 1 : 0;
}
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ __launch_bounds__(LQP_NT) void k_spd_step(const FwdParams<float> P, const int* __restrict__ gate, const int k,
                                                      const int pivot_tasks) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    int b, part;                                                   // (the two workgroups of a matrix on one XCD: the panel is shared in its L2)
    if (!shared_map((int)blockIdx.x, P.B, SPD_NP, b, part)) return;
    // W, W^T of the next pivot block travel between the launches in the (unused on this path) KKT matrix area
    wg_spd_sweep<SPD_NP>(spd_half(P, b, (P.Ks - k) & 1), P.Ks, P.info + b, smem, nullptr, spd_half(P, b, (P.Ks - k - 1) & 1),
                         k, k + 1, part, P.M + (size_t)b * P.Np * P.Np, pivot_tasks);
}
// 512 < n <= 1024, few problems: one launch per PHASE of a pivot step (1: pivot block + Y phase, 2: tile updates), two
// workgroups per matrix, in place in half 0; the panel scratch is the (unused on this path) KKT matrix area.
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ __launch_bounds__(LQP_NT) void k_spd_big_step(const FwdParams<float> P, const int* __restrict__ gate, const int k,
                                                          const int phases) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    int b, part;
    if (!shared_map((int)blockIdx.x, P.B, SPD_NP, b, part)) return;
    wg_spd_sweep_big<SPD_NP>(spd_half(P, b, 0), P.Ks, P.info + b, smem, P.M + (size_t)b * P.Np * P.Np, k, k + 1, phases, part);
}
// ---- what k_fwd_setup leaves to the resident sweep under prep_fused == 3 (the same operations on the same values as
//      there: the same bits).  d: the scaling vector in LDS. ----
// One wave each (they run beside the pivot block of the sweep's first step).
__device__ __forceinline__ void wave_deferred_vectors(const FwdParams<float>& P, const int b, const float* __restrict__ d) {
    const int n = P.n, lane = threadIdx.x & 63;
    VecView<float> V(P.vecs + (size_t)b * P.vstride, n, P.m);
    const float* p = P.p + (size_t)b * n;
    const float* lb = P.lb + (size_t)b * n;
    const float* ub = P.ub + (size_t)b * n;
    for (int i0 = 0; i0 < n; i0 += 256) {               // (twelve requests in flight)
        float pv[4], lv[4], uv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u + lane, ic = i < n ? i : n - 1;
            pv[u] = p[ic]; lv[u] = lb[ic]; uv[u] = ub[ic];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u + lane;
            if (i < n) {
                const float di = d[i];
                V.D[i] = di;
                V.ps[i] = di * pv[u];                // (:177)
                V.lbs[i] = lv[u] / di;               // (:192-194; +-inf / D stays +-inf)
                V.ubs[i] = uv[u] / di;
            }
        }
    }
}
// equality block: A D, row normalisation E (:179-190)
__device__ __forceinline__ void wave_deferred_eq_rows(const FwdParams<float>& P, const int b, const float* __restrict__ d) {
    const int n = P.n, m = P.m, lane = threadIdx.x & 63;
    if (m <= 0) return;
    VecView<float> V(P.vecs + (size_t)b * P.vstride, n, m);
    const float* A = P.A + (size_t)b * m * n;
    const float* bb = P.b + (size_t)b * m;
    float esum = 0.f, mine = 0.f;                           // lane r keeps the norm of row r (m <= SPD_MAXM <= 64)
    for (int r = 0; r < m; ++r) {
        float am = 0.f;
        for (int j = lane; j < n; j += 64) {
            const float v = A[(size_t)r * n + j] * d[j];
            V.As[(size_t)r * n + j] = v;
            am = tmax(am, tabs(v));
        }
        am = wave_max(am);
        if (lane == r) mine = am;
        esum += am;
    }
    const float floor_a = tmax(esum / (float)m, 1e-6f);
    for (int r = 0; r < m; ++r) {
        float an = __shfl(mine, r);
        if (an <= 0.f) an = tmax(an, floor_a);
        const float e = 1.f / an;
        for (int j = lane; j < n; j += 64) V.As[(size_t)r * n + j] = e * V.As[(size_t)r * n + j];      // (this lane's own stores)
        if (lane == 0) { V.E[r] = e; V.bs[r] = e * bb[r]; }
    }
}
// the hooks of wg_spd_sweep_resident_v2 (RsNoScaling)
struct RsSetupHooks {
    const FwdParams<float>& P;
    int b, part, np;
    __device__ __forceinline__ void scaling(float* red, float* d, float* work) const {
        wg_scaling_vector<float, RS_NT>(P, b, red, 1, d, work, work + 8);
    }
    __device__ __forceinline__ void deferred(const float* d) const {
        if (part == 0) wave_deferred_vectors(P, b, d);
        if (part == np - 1) wave_deferred_eq_rows(P, b, d);
    }
};

// all pivot steps in ONE launch, the matrix resident in the registers of its two workgroups (lqp_spd.hpp).  Reads the
// blocks k_spd_begin built (half Ks & 1 of the packed area), leaves -(Qs + rho I)^-1 in half 0, where the loop reads it.
// Exchange buffer: the (unused on this path) KKT-matrix area; step flags: behind the loop's exchange granules.
template <int KS, int NP = 2, bool F16 = false>
__global__ __launch_bounds__(RS_NT) void k_spd_resident(const FwdParams<float> P, const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    int b, part;                                                   // (the NP workgroups of a problem on one XCD: shared_map)
    if (!shared_map((int)blockIdx.x, P.B, NP, b, part)) return;
    unsigned int* fl = (unsigned int*)(P.xchg + (size_t)P.B * XCHG_WORDS + (size_t)XCHG_TAIL * b);
    const unsigned int epoch = 32u * (unsigned int)P.status[ST_NFACTOR];
    RsLateRho lr;
    const bool fused = gate == nullptr && P.prep_fused;     // unscaled blocks from k_spd_prep (first factorisation only)
    const bool qpass = fused && P.prep_fused == 3;          // ... or no blocks at all: the sweep reads Q itself
    lr.on = (gate == nullptr && P.rho_late && !fused) ? 1 : 0;
    lr.n = P.n; lr.rho_min = P.rho_min; lr.rho_max = P.rho_max;
    lr.rho_out = part == 0 ? P.scal + (size_t)b * SC_WORDS + SC_RHO : nullptr;
    lr.dsc = (fused && !qpass) ? VecView<float>(P.vecs + (size_t)b * P.vstride, P.n, P.m).D : nullptr;
    lr.q = qpass ? P.Q + (size_t)b * P.n * P.n : nullptr;
    lr.cmx = qpass ? prep_scratch(P, b) : nullptr;
    lr.dbg_tid = 64 * (P.dbg_qpass >> 8);
    lr.qdbg = (qpass && P.dbg && (P.dbg_qpass & 1) && part == 0) ? P.dbg + (size_t)(P.B + b) * 8 : nullptr;      // (LQP_DBG_QPASS=1: the debug buffer holds 2 B x 8 words)
    lr.fro_self = (fused && P.rho_mode == 0) ? 1 : 0;
    lr.rho_given = (fused && P.rho_mode != 0) ? P.scal[(size_t)b * SC_WORDS + SC_RHO] : 0.f;
    lr.xcd_local = P.xcd_local;
    const RsSetupHooks hooks{P, b, part, NP};
    wg_spd_sweep_resident_v2<KS, NP, RsSetupHooks, F16>(spd_half(P, b, KS & 1), spd_half(P, b, 0), P.M + (size_t)b * P.Np * P.Np, fl, epoch, part,
                                 P.info + b, P.status + ST_TIMEOUT, smem, lr,
                                 (P.dbg && part == 0) ? P.dbg + (size_t)b * 8 : nullptr, hooks);
}
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ __launch_bounds__(LQP_NT) void k_spd_end(const FwdParams<float> P, const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    if (gate && *gate == 0) return;
    const int b = blockIdx.x;
    if (threadIdx.x == 0 && P.info[b] != 0) P.status[ST_NOTSPD] = 1;
    if (P.m > 0) wg_eq_correct(P, b, smem);
}

// equality duals of the last x-update.  LU path: the tail of the solve vector.  Symmetric path:
// nu = S^-1 (G^T w - b) = T^T w - s0, left in LDS by the loop
template <typename T, int NT, bool SYM>
__device__ __forceinline__ void loop_store_nu(const VecView<T>& V, const T* __restrict__ v, const T* __restrict__ nus_l,
                                              const int n, const int m) {
    const T* src = SYM ? nus_l : v + n;          // symmetric path: computed while v still held w (check / last iteration)
    for (int r = threadIdx.x; r < m; r += NT) V.nu[r] = src[r];
}
// LDS of the loop on the symmetric path; rl = LDS-resident blocks
__host__ __device__ inline int sym_loop_lds_bytes(int n, int m, int Ks, int rl, int nw = LQP_NW) {
    const int Nps = Ks * LQP_NB;
    return (rl * LQP_BLK + 3 * Nps + sym_blocks(Ks) * 64 + nw * Nps + 6 * n + 2 * m + nw * 8 + 8) * 4 + 64;
}
// how many blocks of the symmetric stream stay in LDS (after the LQP_RREG register blocks)
__host__ __device__ inline int sym_resident_lds_blocks(int n, int m, int Ks, int nreg = LQP_RREG, int nw = LQP_NW) {
    const int S = sym_blocks(Ks);
    int rl = S - nreg;
    if (rl < 0) rl = 0;
    while (rl > 0 && sym_loop_lds_bytes(n, m, Ks, rl, nw) > 160 * 1024) --rl;
    return rl;
}

// ---- epilogue: undo scaling, duals (:316-327), report ----
// Runs as the END of the forward's last kernel: k_fwd_epilogue, or the continuation loop kernel itself (`persistent & 4`:
// one launch and its boundary less per solve).  Also ORs the per-problem bound flags into the status block and, when the
// caller gave pinned host memory, reports there directly: every workgroup its own info / flag words, workgroup 0 the status
// block -- no device-to-host copy behind the solve (a 1.4 us blit kernel + a boundary).  Words other workgroups of the SAME
// launch may still write (ST_NOTSPD, ST_TIMEOUT of an in-kernel refactorisation) are therefore also kept per problem
// (RP_NOTSPD / RP_TIMEOUT in the writer's own flag word).
template <typename T>
__device__ __forceinline__ void fwd_finish(const FwdParams<T>& P, const int b) {
    const int n = P.n, m = P.m, tid = threadIdx.x, nt = blockDim.x;
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    const T rho = P.scal[(size_t)b * SC_WORDS + SC_RHO];
    // a failed factorisation (singular KKT matrix / Q + rho I not positive definite) or a barrier timeout must not
    // leave plausible-looking numbers behind: callers that did not wait for the status see NaN
    const int info_b = P.info[b];
    const int notspd = __hip_atomic_load(P.status + ST_NOTSPD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int tmo = __hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool bad = info_b != 0 || notspd != 0 || tmo != 0;
    const T poison = bad ? T(__builtin_nanf("")) : T(0);
    for (int i = tid; i < n; i += nt) {
        const T d = V.D[i];
        const T xo = d * V.x[i] + poison, zo = d * V.z[i] + poison, uo = V.u[i] / d + poison;
        P.x[(size_t)b * n + i] = xo;
        P.z[(size_t)b * n + i] = zo;
        P.u[(size_t)b * n + i] = uo;
        const T y = uo * rho;
        P.lams[(size_t)b * 2 * n + i] = (-y > T(0)) ? -y : T(0);
        P.lams[(size_t)b * 2 * n + n + i] = (y > T(0)) ? y : T(0);
    }
    for (int r = tid; r < m; r += nt) P.nus[(size_t)b * m + r] = V.nu[r] * V.E[r] + poison;
    if (tid == 0) P.rho_out[b] = rho;
    const int mine = P.bflags[b] | (tmo ? RP_TIMEOUT : 0) | ((notspd && P.spd && info_b != 0) ? RP_NOTSPD : 0);
    if (P.host_report && tid == 0) {
        __hip_atomic_store(P.host_report + ST_WORDS + b, info_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(P.host_report + ST_WORDS + P.B + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (b == 0) {
        // the bound flags of the batch (the per-problem words were written by an earlier launch)
        int f = 0;
        for (int i = tid; i < P.B; i += nt) f |= P.bflags[i];
        const int any_lb = __syncthreads_or(f & RP_LB) ? 1 : 0, any_ub = __syncthreads_or(f & RP_UB) ? 1 : 0;
        if (tid < ST_WORDS) {
            int v = __hip_atomic_load(P.status + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == ST_ANY_LB) v = (any_lb || (P.bound_flags_in && P.bound_flags_in[0] != 0)) ? 1 : 0;
            if (tid == ST_ANY_UB) v = (any_ub || (P.bound_flags_in && P.bound_flags_in[1] != 0)) ? 1 : 0;
            if (tid == ST_ANY_LB || tid == ST_ANY_UB) P.status[tid] = v;
            if (P.host_report) __hip_atomic_store(P.host_report + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// ---------------------------------------------------------------------------
// The ADMM loop (:235-313).  RES: resident head of the factor stream (registers + LDS); SYM: symmetric-inverse
// x-update instead of the cached triangular solves; NT: threads.
// TAIL = continuation launch of the speculative (no host sync) schedule.  It has its own name in traces /
// profiles, exits at once when the first launch converged and, when `persistent & 2`, does the adaptive-rho
// steps (:237-256) in-kernel: global decision from the counters of the last check, masked rho update and
// refactorisation by this workgroup -- LU path: at its first iteration (LU + re-pack); symmetric path: at every
// event inside [it0, it1), segment by segment.  Only this cold variant carries the factorisation code; the first
// (hot) launch stays lean.
// ---------------------------------------------------------------------------
// NP == 2 (symmetric path above 512 rows, first launch, 2 B workgroups resident: k_admm_loop_np2): workgroups `part` = 0 / 1 of a
// problem stream one RANGE of whole block columns of H each -- [0, jc) and [jc, Ks), about half of the blocks: twice the registers
// and LDS under the same matrix, half the stream per CU -- and exchange their partial products every iteration as tagged 8-byte
// granules (the hand-off of k_admm_loop_split: two buffers by iteration parity, tag = iteration + 1, the area zeroed by the setup
// kernel), added in the fixed order part 0 + part 1: both hold bit-identical iterates and run the element-wise update and the
// checks redundantly; part 0 alone reports to the counters and writes state.
template <typename T, bool RES, bool TAIL, int NT, bool SYM, int NP = 1>
__device__ __forceinline__ void admm_loop_body_from(const FwdParams<T>& P, int it0, const int it1, int ctr_base, int prev_slot,
                                                    const int persistent, char* smem);
template <typename T, bool RES, bool TAIL, int NT, bool SYM = false, int NP = 1>
__device__ __forceinline__ void admm_loop_body(const FwdParams<T>& P, const int it0, const int it1,
                                               const int ctr_base,       // counter slot of check it0 / check
                                               const int prev_slot,      // slot of the last check before it0, -1: none / known not done
                                               const int persistent, char* smem) {
    // every problem stopped at an earlier check -> nothing to do (break at :312)
    if (__hip_atomic_load(P.status + ST_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    admm_loop_body_from<T, RES, TAIL, NT, SYM, NP>(P, it0, it1, ctr_base, prev_slot, persistent, smem);
}
template <typename T, bool RES, bool TAIL, int NT, bool SYM, int NP>
__device__ __forceinline__ void admm_loop_body_from(const FwdParams<T>& P, int it0, const int it1, int ctr_base, int prev_slot,
                                                    const int persistent, char* smem) {
    static_assert(NP == 1 || (NP == 2 && SYM && RES && !TAIL && NT == 1024 && sizeof(T) == 4), "two workgroups per problem: the hot symmetric loop only");
    int b = blockIdx.x, part_id = 0;
    if constexpr (NP == 2) { if (!shared_map((int)blockIdx.x, P.B, 2, b, part_id)) return; }
    const bool lead = part_id == 0;                          // (the workgroup that reports and writes state)
    const int n = P.n, m = P.m, N = P.N, Np = P.Np, K = P.K;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if constexpr (TAIL) {
        // persistent & 8: the hot two-workgroup loop ran past the first possible rho event (FwdParams::hot_past) and names the
        // iteration at which it stopped -- an event that changes something, or the end of its range
        if (persistent & 8) {
            const int r = __builtin_amdgcn_readfirstlane(P.status[ST_RESUME]);
            if (r > 0) {
                it0 = r;
                ctr_base = ((r + P.check_solved - 1) / P.check_solved) % P.ring;
                prev_slot = ((r - 1) / P.check_solved) % P.ring;
            }
        }
    }
    if (prev_slot >= 0) {
        if (__hip_atomic_load(P.counters + (size_t)prev_slot * CT_WORDS + CT_NOTOPT, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (b == 0 && tid == 0 && lead) {
                P.status[ST_FINAL_ITER] = it0 - 1;
                __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    if (it0 >= it1) return;
    // LDS carve.  LU path:  [resident blocks] v tmp | z u ps lb ub D bs red | dest
    //            SYM path: [resident blocks] v xs ylds cvl part[NW][Nps] | z u ps lb ub D bs red
    const int Nps = SYM ? P.Ks * LQP_NB : Np;                // padded length of the solve vector
    int rl = SYM ? (NT == 512 ? P.sym_rl_hot : P.sym_rl) : LQP_RLDS;
    // NP == 2: this workgroup's range of the stream -- block columns [rj0, rj0 + rwc), rsn blocks from stream index rs0.  Only ITS
    // row-sum slots and column sums are kept (the layout the host sized holds them for the whole stream): what that frees holds
    // more blocks of H -- three / two at Ks = 16.
    int rj0 = 0, rs0 = 0, rsn = -1, rwc = P.Ks;
    int ylds_slots = SYM ? sym_blocks(P.Ks) : 0, part_stride = Nps;
    if constexpr (NP == 2) {
        const int S_ = sym_blocks(P.Ks);
        int jc = 0, acc = 0;
        while (jc < P.Ks && 2 * acc < S_) { acc += P.Ks - jc; ++jc; }      // columns [0, jc): the first to reach half of the blocks
        rj0 = part_id == 0 ? 0 : jc; rs0 = part_id == 0 ? 0 : acc; rsn = part_id == 0 ? acc : S_ - acc;
        rwc = part_id == 0 ? jc : P.Ks - jc;
        const int freed = (S_ - rsn) * 64 + (NT / 64) * (Nps - rwc * 64);      // floats
        ylds_slots = rsn; part_stride = rwc * 64;
        rl += freed / LQP_BLK;
    }
    T* lds_res = (T*)smem;                                   // resident blocks (RES only), 16-KB aligned chunks
    T* v = lds_res + (RES ? (size_t)rl * LQP_BLK : 0);
    T* tmp = v + Nps;                                        // LU: 64 scratch; SYM: xs (the new x)
    T* xs = tmp;
    T* ylds = xs + Nps;                                      // SYM: one 64-slot per block of the stream
    T* cvl = ylds + (size_t)ylds_slots * 64;
    T* part = cvl + Nps;
    T* z = SYM ? part + (size_t)(NT / 64) * part_stride : tmp + 64;
    T* u = z + n;
    T* ps = u + n;
    T* lb = ps + n;
    T* ub = lb + n;
    T* D = ub + n;
    T* bs = D + n;
    T* red = bs + (SYM ? 2 * m : m);                        // SYM: nu of the current check sits behind bs
    int* dest = (int*)(red + (NT / 64) * 8 + 8);             // LU only

    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T* scal = P.scal + (size_t)b * SC_WORDS;
    const T* packed = P.packed + (size_t)b * packed_blocks(K) * LQP_BLK;
    // The continuation launch of the symmetric path walks through SEGMENTS [seg0, seg1) that end at the
    // adaptive-rho events (:237-256) and refactorises in between, inside the kernel; every other build runs
    // exactly one segment [it0, it1).
    int seg0 = it0;
    const int it_end = it1;
  while (true) {
    int seg1 = it_end;
    if constexpr (TAIL && SYM) {
        if (P.adaptive_rho && (persistent & 2)) {
            const int nxt = (seg0 / P.ar_iter + 1) * P.ar_iter;
            if (nxt < P.ar_max && nxt < seg1) seg1 = nxt;
            if (seg0 > 0 && seg0 % P.ar_iter == 0 && seg0 < P.ar_max) {
                unsigned int* ctl = P.counters + (size_t)(((seg0 - 1) / P.check_solved) % P.ring) * CT_WORDS;
                if (__hip_atomic_load(ctl + CT_WANTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 &&
                    __hip_atomic_load(ctl + CT_TRIG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {      // uniform over the whole grid
                    T rho_ = scal[SC_RHO];
                    if (scal[SC_WANTS] != T(0)) rho_ = rho_ * scal[SC_RATIO];
                    rho_ = tmin(tmax(rho_, P.rho_min), P.rho_max);
                    __syncthreads();
                    if (tid == 0) scal[SC_RHO] = rho_;
                    if (b == 0 && tid == 0) { P.status[ST_NFACTOR] += 1; P.status[ST_RHO_UPDATED] = 1; }
                    wg_spd_factor(P, b, rho_, smem, false);
                    __syncthreads();
                }
            }
        }
    }
    if constexpr (TAIL && !SYM && NT == 1024) {
        // The LU path's continuation launch walks through the adaptive-rho events (:237-256) the same way (round 5: one launch
        // per possible event before -- nine at the defaults, each ~5 us of nothing when the solve was over): the decision from the
        // counters of the last check (complete: every check of a persistent launch ends in a device-wide wait), the masked rho
        // update, KKT re-assembly, pivoted LU and pack in-kernel.
        if (P.adaptive_rho && (persistent & 2)) {
            const int nxt = (seg0 / P.ar_iter + 1) * P.ar_iter;
            if (nxt < P.ar_max && nxt < seg1) seg1 = nxt;
            if (seg0 > 0 && seg0 % P.ar_iter == 0 && seg0 < P.ar_max) {
                unsigned int* ctl = P.counters + (size_t)(((seg0 - 1) / P.check_solved) % P.ring) * CT_WORDS;
                if (__hip_atomic_load(ctl + CT_WANTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 &&
                    __hip_atomic_load(ctl + CT_TRIG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {          // uniform over the whole grid
                    constexpr int kPB = sizeof(T) == 4 ? 16 : 8;
                    T rho_ = scal[SC_RHO];
                    if (scal[SC_WANTS] != T(0)) rho_ = rho_ * scal[SC_RATIO];
                    rho_ = tmin(tmax(rho_, P.rho_min), P.rho_max);
                    __syncthreads();
                    if (tid == 0) scal[SC_RHO] = rho_;
                    if (b == 0 && tid == 0) { P.status[ST_NFACTOR] += 1; P.status[ST_RHO_UPDATED] = 1; }
                    const bool lazy_ = P.scale && P.qs_lazy;
                    const T* Qs_ = (P.scale && !lazy_) ? (P.Qs + (size_t)b * n * P.ldq) : (P.Q + (size_t)b * n * n);
                    assemble_kkt_rows(P, b, Qs_, (P.scale && !lazy_) ? P.ldq : n, V, rho_, true, lazy_ ? V.D : nullptr);
                    __syncthreads();
                    wg_lu_factor<T, kPB, sizeof(T) == 4, NT>(P.M + (size_t)b * Np * Np, N, Np, P.piv + (size_t)b * Np,
                                                                P.info + b, smem, nullptr);
                    __syncthreads();
                    wg_pack_factor<T>(P.M + (size_t)b * Np * Np, N, Np, P.piv + (size_t)b * Np,
                                      P.packed + (size_t)b * packed_blocks(K) * LQP_BLK, P.dest + (size_t)b * Np, smem, true);
                    __syncthreads();
                }
            }
        }
    }
    const int it0 = seg0, it1 = seg1;                        // (shadow the launch bounds inside the segment)
    const T rho = scal[SC_RHO];
    const T pnorm = scal[SC_PNORM];
    const int S = SYM ? sym_blocks(P.Ks) : K * (K + 1);
    const bool cyclic = true;                 // (wg_packed_solve pads the stream to a multiple of the ring depth)

    BlockStream<T, NT> st;
    ResidentRegs<T, NT> rr;
    const T* packed_r = packed + (size_t)rs0 * LQP_BLK;
    unsigned long long* const xq = NP == 2 ? P.xchg + (size_t)b * XCHG_WORDS : nullptr;
    // (LDS-resident blocks actually used: the layout is sized for the whole stream, a range may be shorter than registers + rl)
    int rl_use = rl;
    if constexpr (NP == 2) { const int room = rsn - resident_regs<NT>(); rl_use = room < 0 ? 0 : (room < rl ? room : rl); }
    if constexpr (SYM) {
        if constexpr (RES) {
            const int Sr_ = NP == 2 ? rsn : S;
            sym_resident_load<NT>(rr, lds_res, packed_r, Sr_, rl_use);
            sym_prime<NT>(st, packed_r, (Sr_ < resident_regs<NT>() ? Sr_ : resident_regs<NT>()) + rl_use, Sr_);
        }
    } else if constexpr (RES) {
        resident_load<T, NT>(rr, lds_res, packed);
        stream_prime_from<T, NT>(st, packed, resident_total<NT>(), S);
    } else {
        stream_prime<T, NT>(st, packed, S);
    }

    for (int i = tid; i < n; i += NT) {
        z[i] = V.z[i]; u[i] = V.u[i]; ps[i] = V.ps[i]; lb[i] = V.lbs[i]; ub[i] = V.ubs[i]; D[i] = V.D[i];
    }
    for (int r = tid; r < m; r += NT) bs[r] = V.bs[r];
    if constexpr (SYM) {
        for (int i = tid; i < Nps; i += NT) cvl[i] = (i < n && m > 0) ? V.cv[i] : T(0);
        if constexpr (NP == 2) {       // (every slot / column sum of the range is written by every product; zeroed once all the same)
            for (int i = tid; i < ylds_slots * 64; i += NT) ylds[i] = T(0);
            for (int i = tid; i < (NT / 64) * part_stride; i += NT) part[i] = T(0);
        }
    } else {
        const int* gdest = P.dest + (size_t)b * Np;
        for (int i = tid; i < Np; i += NT) dest[i] = gdest[i];
    }
    const T* xv = SYM ? xs : v;                              // where the x-update leaves x
    __syncthreads();

    unsigned long long dbt[4] = {0, 0, 0, 0}, dt0 = 0;      // debug: cycles in rhs / product / combine / update+check
    const bool dbg_on = SYM && P.dbg != nullptr;
    int slot = TAIL ? ((it0 + P.check_solved - 1) / P.check_solved) % P.ring : ctr_base;
    for (int it = it0; it < it1; ++it) {
        if (dbg_on) dt0 = clock64();
        const bool check = (it % P.check_solved) == 0;
        T mx[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) mx[q] = T(0);
        // ||Q x / D||_inf of the check (:299) without touching Q: the x-update solved (Qs + rho I) x + As^T nu = w
        // exactly (to the solve's rounding), so Qs x = w - rho x - As^T nu.  It only feeds a tolerance SCALE.
        if constexpr (SYM) {
            // ---- symmetric path: product | barrier | ONE fused element-wise pass (combine, z/u update, residual
            //      norms, next right-hand side) | barrier.  v holds w = -p + rho (z - u) on entry. ----
            if (it == it0) {
                for (int i = tid; i < Nps; i += NT) v[i] = (i < n) ? -ps[i] + rho * (z[i] - u[i]) : T(0);
                wg_barrier_lds();
            }
            if (dbg_on) { const unsigned long long t = clock64(); dbt[0] += t - dt0; dt0 = t; }
            if constexpr (NP == 2)         // (the range's slots and column sums: base pointers moved so that the walk's own indices fit)
                wg_sym_gemv<RES, NT>(st, rr, lds_res, rl_use, packed_r, P.Ks, part_stride, v, ylds - (size_t)rs0 * 64, part - (size_t)rj0 * 64,
                                     rj0, rs0, rsn);
            else wg_sym_gemv<RES, NT>(st, rr, lds_res, rl, packed, P.Ks, Nps, v, ylds, part);
            wg_barrier_lds();
            if (dbg_on) { const unsigned long long t = clock64(); dbt[1] += t - dt0; dt0 = t; }
            T* nus_l = bs + m;
            if ((check || it + 1 == it1) && m > 0) {         // nu = T^T w - s0 (one wave per row) while v is still w
                for (int r = w; r < m; r += (NT / 64)) {
                    T acc = T(0);
                    for (int i = lane; i < n; i += 64) acc += V.Tm[(size_t)r * n + i] * v[i];
                    acc = wave_sum(acc);
                    if (lane == 0) nus_l[r] = acc - V.s0[r];
                }
                wg_barrier_lds();
            }
            if (dbg_on) { const unsigned long long t = clock64(); dbt[2] += t - dt0; dt0 = t; }
            for (int i = tid; i < Nps; i += NT) {
                T ysum;
                if constexpr (NP == 2) {
                    // the partial product of this workgroup's columns: row sums of the blocks (bi, j), j in its range, j <= bi; column sums
                    // where column bi is its own (the terms of sym_combine that are not zero for this range, in sym_combine's order)
                    const int bi = i >> 6, r_ = i & 63;
                    ysum = T(0);
                    const int jhi = bi < rj0 + rwc - 1 ? bi : rj0 + rwc - 1;
                    for (int j = rj0; j <= jhi; ++j) ysum += ylds[(size_t)(sym_idx(bi, j, P.Ks) - rs0) * 64 + r_];
                    if (bi >= rj0 && bi < rj0 + rwc) {
#pragma unroll
                        for (int ww = 0; ww < NT / 64; ++ww) ysum += part[(size_t)ww * part_stride + (i - rj0 * 64)];
                    }
                } else {
                    ysum = sym_combine<NT>(i, P.Ks, Nps, ylds, part);
                }
                if constexpr (NP == 2) {
                    // this workgroup's partial out, the partner's in (one granule per element, thread i = element i)
                    const unsigned int tag = (unsigned int)(it + 1);
                    unsigned long long* base = xq + (size_t)(it & 1) * (2 * SPD_BIGK * LQP_NB);
                    __hip_atomic_store(base + (size_t)part_id * (SPD_BIGK * LQP_NB) + i,
                                       ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned int, ysum),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long* src = base + (size_t)(1 - part_id) * (SPD_BIGK * LQP_NB) + i;
                    unsigned long long g = 0;
                    unsigned int spins = 0;
                    unsigned long long t0 = 0;
                    bool lost = false;
                    for (;;) {
                        g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned int)(g >> 32) == tag) break;
                        if ((++spins & 1023u) == 0) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz
                            if (t0 == 0) t0 = now;
                            else if (now - t0 > 50000000ULL ||
                                     __hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {      // 0.5 s: give up, flagged
                                __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                lost = true;
                                break;
                            }
                        }
                    }
                    // (a partner that never showed up must not leave plausible numbers: NaN from here on, and the time-out word)
                    const T other = lost ? T(__builtin_nanf("")) : __builtin_bit_cast(float, (unsigned int)g);
                    ysum = part_id == 0 ? ysum + other : other + ysum;
                }
                const T xi = cvl[i] - ysum;
                xs[i] = xi;
                T wn = T(0);
                if (i < n) {
                    const T zp = z[i];
                    const T ui = u[i];
                    T zn = xi + ui;
                    zn = tmin(tmax(zn, lb[i]), ub[i]);     // (:273-276; an infinite bound is a no-op)
                    const T r = xi - zn;
                    const T s = rho * (zn - zp);
                    const T un = ui + r;
                    z[i] = zn;
                    u[i] = un;
                    if (check) {
                        const T di = D[i];
                        mx[0] = tmax(mx[0], tabs(di * r));
                        mx[1] = tmax(mx[1], tabs(di * s));
                        mx[2] = tmax(mx[2], tabs(di * xi));
                        mx[3] = tmax(mx[3], tabs(di * zn));
                        mx[4] = tmax(mx[4], tabs((rho * di) * un));
                        T qx = v[i] - rho * xi;
                        for (int q = 0; q < m; ++q) qx -= V.As[(size_t)q * n + i] * nus_l[q];
                        mx[5] = tmax(mx[5], tabs(qx / di));
                    }
                    wn = -ps[i] + rho * (zn - un);           // next iteration's right-hand side
                }
                v[i] = wn;
            }
        } else {
            // ---- rhs = [-p + rho (z - u); b], scattered to its pivoted position (:259-262) ----
            for (int i = tid; i < Np; i += NT) {
                T val = T(0);
                if (i < n) val = -ps[i] + rho * (z[i] - u[i]);
                else if (i < N) val = bs[i - n];
                v[dest[i]] = val;
            }
            wg_barrier_lds();
            // ---- x-update: cached triangular solves (:267) ----
            if constexpr (RES) {
                wg_packed_solve_resident<T, NT>(st, rr, lds_res, packed, K, v, tmp, true);
            } else {
                wg_packed_solve<T, NT>(st, packed, K, v, tmp, cyclic);
                if (!cyclic && it + 1 < it1) stream_prime<T, NT>(st, packed, S);
            }
            // ---- z-update, residuals, dual (:271-282) ----
            const T* nul = v + n;                            // nu is the tail of the solution
            for (int i = tid; i < n; i += NT) {
                const T xi = v[i];
                const T zp = z[i];
                const T ui = u[i];
                T zn = xi + ui;
                zn = tmin(tmax(zn, lb[i]), ub[i]);
                const T r = xi - zn;
                const T s = rho * (zn - zp);
                const T un = ui + r;
                z[i] = zn;
                u[i] = un;
                if (check) {
                    const T di = D[i];
                    mx[0] = tmax(mx[0], tabs(di * r));
                    mx[1] = tmax(mx[1], tabs(di * s));
                    mx[2] = tmax(mx[2], tabs(di * xi));
                    mx[3] = tmax(mx[3], tabs(di * zn));
                    mx[4] = tmax(mx[4], tabs((rho * di) * un));
                    T qx = -ps[i] + rho * (zp - ui) - rho * xi;
                    for (int q = 0; q < m; ++q) qx -= V.As[(size_t)q * n + i] * nul[q];
                    mx[5] = tmax(mx[5], tabs(qx / di));
                }
            }
        }
        if (check) {
            T mv[6] = {mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]};
            wg_max_n<T, 6, NT / 64>(mv, red);
            const T tiny = T(1e-16);
            const T pri_scale = tmax(tmax(mv[2], mv[3]), tiny);
            const T tol_p = P.eps_abs + P.eps_rel * pri_scale;
            const T dua_scale = tmax(tmax(tmax(mv[4], mv[5]), pnorm), tiny);
            const T tol_d = P.eps_abs + P.eps_rel * dua_scale;
            const bool solved = (mv[0] < tol_p) && (mv[1] < tol_d);
            const bool wants = (mv[0] > tmax(tol_p, P.ar_thr)) || (mv[1] > tmax(tol_d, P.ar_thr));
            // ratio for the next adaptive-rho step (:239-245)
            const T num = tmax(mv[0] / pri_scale, tiny);
            const T den = tmax(mv[1] / dua_scale, tiny);
            const T ratio = tsqrt(num / den);
            const bool trig = (ratio > P.ar_tol) || (ratio < P.ar_inv_tol);
            unsigned int* ct = P.counters + (size_t)slot * CT_WORDS;
            if (tid == 0 && lead) {
                scal[SC_RATIO] = ratio;
                scal[SC_WANTS] = wants ? T(1) : T(0);
                scal[SC_PRI] = mv[0];                        // primal / dual error of this check (the NumPy twin returns them)
                scal[SC_DUA] = mv[1];
                trace_check(P.vtrace, it, P.check_solved, P.ring, mv[0], mv[1]);
                unsigned int r1 = 0, r2 = 0;
                if (wants) r1 = atomicAdd(ct + CT_WANTS, 1u);
                if (trig) r2 = atomicAdd(ct + CT_TRIG, 1u);
                if (persistent & 1) {
                    // the arrival and this problem's verdict in ONE 64-bit add (NOTOPT and ARRIVE share an aligned word), behind
                    // the RETURNED adds above: whoever sees the last arrival sees every counter of the check.  Nothing else
                    // is handed over here -- no release fence (an L2 write-back, microseconds per check)
                    asm volatile("s_waitcnt vmcnt(0)" :: "v"(r1), "v"(r2) : "memory");
                    __hip_atomic_fetch_add((unsigned long long*)(ct + CT_NOTOPT), (solved ? 0ull : 1ull) | (1ull << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (!solved) {
                    atomicAdd(ct + CT_NOTOPT, 1u);
                }
            }
            ++slot;
            if (persistent & 1) {
                // all workgroups resident: device-wide "all optimal?" (torch.all at :312)
                grid_wait(ct + CT_ARRIVE, NP == 2 ? (unsigned int)P.B : gridDim.x, P.status);      // (one arrival per problem)
                const unsigned int notopt = __hip_atomic_load(ct + CT_NOTOPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int tmo = __hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (notopt == 0 || tmo) {
                    if (b == 0 && tid == 0 && lead) {
                        P.status[ST_FINAL_ITER] = it;
                        __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // leave the loop with the state of iteration `it`
                    __syncthreads();
                    if (lead) {
                        for (int i = tid; i < n; i += NT) { V.z[i] = z[i]; V.u[i] = u[i]; V.x[i] = xv[i]; }
                        loop_store_nu<T, NT, SYM>(V, v, bs + m, n, m);
                    }
                    if (dbg_on && tid == 0 && lead) {
                        dbt[3] += clock64() - dt0;
                        for (int q = 0; q < 4; ++q) P.dbg[(size_t)b * 8 + q] += dbt[q];
                    }
                    return;
                }
            }
        }
        wg_barrier_lds();
        if (dbg_on) { const unsigned long long t = clock64(); dbt[3] += t - dt0; }
    }
    if (dbg_on && tid == 0 && lead)
        for (int q = 0; q < 4; ++q) P.dbg[(size_t)b * 8 + q] += dbt[q];
    // ---- save state for the next launch / the epilogue ----
    if (lead) {
        for (int i = tid; i < n; i += NT) { V.z[i] = z[i]; V.u[i] = u[i]; V.x[i] = xv[i]; }
        loop_store_nu<T, NT, SYM>(V, v, bs + m, n, m);
    }
    if (!TAIL || seg1 >= it_end) break;
    seg0 = seg1;
    __syncthreads();
  }
}
// the first launch of the symmetric path above 512 rows on TWO workgroups per problem (admm_loop_body_from, NP == 2)
template <int LQP_ANY = 0>
__global__ __launch_bounds__(1024) void k_admm_loop_np2(const FwdParams<float> P, const int it0, const int it1, const int ctr_base,
                                                        const int prev_slot, const int persistent) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    admm_loop_body<float, true, false, 1024, true, 2>(P, it0, it1, ctr_base, prev_slot, persistent, smem);
}

// persistent & 4 (continuation launches): this is the forward's LAST launch -- it ends with the epilogue (fwd_finish)
template <typename T, bool RES, bool TAIL, int NT, bool SYM = false>
__global__ __launch_bounds__(NT) void k_admm_loop(const FwdParams<T> P, const int it0, const int it1, const int ctr_base,
                                                      const int prev_slot, const int persistent) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    admm_loop_body<T, RES, TAIL, NT, SYM>(P, it0, it1, ctr_base, prev_slot, persistent, smem);
    if constexpr (TAIL) {
        if (persistent & 4) {
            __syncthreads();                   // (the state the loop left in global memory: written by other threads)
            fwd_finish(P, blockIdx.x);
        }
    }
}

// ---------------------------------------------------------------------------
// The same loop with TWO workgroups per QP (symmetric x-update, f32, 2 B <= #CUs, Ks >= SPLIT_MINK): workgroups b
// and b + B hold one half of the blocks of H each, ALL of them on chip for the whole launch (lqp_spd.hpp,
// wg_sym_gemv_split), and exchange their partial products every iteration:
//   thread e < Nps combines its element of this workgroup's partial, publishes it as ONE 8-byte granule
//   {tag, value} (agent-scope relaxed atomic store = sc1 write-through store), polls the partner's granule of the
//   same element (sc1 loads, L1 bypassed) until the tag matches, and adds the two partials in the fixed order
//   part 0 + part 1 -- both workgroups then hold bit-identical iterates and run the element-wise update and the
//   checks redundantly (part 0 alone reports to the counters / writes state).
// Two granule buffers alternate by iteration parity: a workgroup can be at most one exchange ahead of its
// partner, so a granule is never overwritten before it was read.  The area is zeroed by k_fwd_setup of the same
// forward (tags start at 1).  Spins are bounded: a timeout sets ST_TIMEOUT and the kernel still drains.
//
// The global stop (torch.all(is_optimal), :312) does not block the loop.  At a check, part 0 adds {not optimal?,
// arrival} to the check's counter word with ONE 64-bit atomic, both workgroups keep a snapshot of the iterate and go
// on iterating; in the following iterations part 0 reads the counter word while its product runs (the load's latency
// is hidden), and once all B arrivals are in, the verdict travels to the partner in the top bits of that iteration's
// granule tags.  "All optimal" -> both restore the snapshot, part 0 stores it and the kernel exits with the
// reference's iteration count (the 1-3 speculative iterations are dropped); otherwise the snapshot is forgotten.
// A verdict still open at the next check or at the last iteration of the launch is waited for (bounded spin).
//
// LDS (floats; every offset but the last three arrays is a compile-time constant):
//   [rl blocks] v yrow cvl part[NW][Nps] z u ps lb ub D xs sz su sx (Nps each) red[NW*8+8] flags[8] | bs nus snu (m each)
// ---------------------------------------------------------------------------
template <int NT, int NP = 2> __host__ __device__ constexpr int split_loop_lds_floats(int Ks) {
    return split_lds_blocks<NT, NP>(Ks) * LQP_BLK + (3 + NT / 64 + 10) * Ks * LQP_NB + (NT / 64) * 8 + 8 + 8;
}
template <int NT, int NP = 2> __host__ __device__ inline int split_loop_lds_bytes(int Ks, int m) {
    // + the equality block: As, G, T (m x Nps each), S, S^-1 (m x m), s0, b, nu, nu snapshot
    return (split_loop_lds_floats<NT, NP>(Ks) + 3 * m * Ks * LQP_NB + 2 * m * m + 4 * m + 8) * 4;
}

// ---------------------------------------------------------------------------
// The hot loop for SMALL problems (symmetric path, n <= 128: BASELINE configs[1], n = 100): the 1024-thread kernel above
// spends 8.4 k cycles per iteration there -- a 16-wave static walk with per-block LDS slots and partial-sum slices, sized
// for 36 blocks, around THREE blocks of work.  Here: 256 threads, the whole (unpacked, full) matrix -H in registers --
// thread t holds the 64 entries of row t >> 1, columns 64 (t & 1) .. -- the product is 64 FMAs per thread against
// broadcast reads of w and ONE lane-pair add; element e of every vector lives in thread e's registers (threads 0..127).
// Same iteration, same check (:285-313, blocking device-wide stop) and same exit state as admm_loop_body; first (hot)
// launch only, continuation launches (adaptive-rho events, a counter ring turn) run the general kernel.
// LDS: w[128] | y[128] | nus[m] | red[4 * 8 + 8]
// ---------------------------------------------------------------------------
__host__ __device__ inline int small_loop_lds_bytes(int m) { return (128 + 128 + (m > 0 ? m : 1) + 4 * 8 + 8 + 8) * 4; }

template <int LQP_ANY = 0>
__global__ __launch_bounds__(256) void k_admm_loop_small(const FwdParams<float> P, const int it0, const int it1,
                                                         const int ctr_base) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    typedef float T;
    constexpr int NT = 256;
    const int b = blockIdx.x, n = P.n, m = P.m, Ks = P.Ks;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (__hip_atomic_load(P.status + ST_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (it0 >= it1) return;
    T* const wv = (T*)smem;
    T* const yv = wv + 128;
    T* const nus_l = yv + 128;
    T* const red = nus_l + (m > 0 ? m : 1);
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T* scal = P.scal + (size_t)b * SC_WORDS;
    const T* packed = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
    const T rho = scal[SC_RHO];
    const T pnorm = scal[SC_PNORM];

    // ---- the full matrix: row r, columns 64 h .. 64 h + 63 (block (i, j) of the packed lower triangle, or the transpose
    //      of block (j, i)) ----
    const int r = tid >> 1, h = tid & 1;
    const int bi = r >> 6, rr = r & 63;
    const bool row_live = bi < Ks && h < Ks;
    T hreg[64];
    if (row_live) {
        if (bi >= h) {
            const T* src = packed + (size_t)sym_idx(bi, h, Ks) * LQP_BLK + rr * 64;
#pragma unroll
            for (int c = 0; c < 64; c += 4) {
                const V4<T> q = *(const V4<T>*)(src + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) hreg[c + e] = q.v[e];
            }
        } else {
            const T* src = packed + (size_t)sym_idx(h, bi, Ks) * LQP_BLK + rr;
#pragma unroll
            for (int c = 0; c < 64; ++c) hreg[c] = src[c * 64];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 64; ++c) hreg[c] = T(0);
    }
    // ---- element e = tid of every vector (threads 0..127) ----
    const int e = tid;
    const bool live = e < n;
    T zi = live ? V.z[e] : T(0), ui = live ? V.u[e] : T(0);
    const T psi = live ? V.ps[e] : T(0), lbi = live ? V.lbs[e] : T(0), ubi = live ? V.ubs[e] : T(0);
    const T di = live ? V.D[e] : T(1), cvi = (live && m > 0) ? V.cv[e] : T(0);
    T xi = T(0);
    if (tid < 128) { wv[tid] = live ? -psi + rho * (zi - ui) : T(0); }
    __syncthreads();

    int slot = ctr_base;
    for (int it = it0; it < it1; ++it) {
        const bool check = (it % P.check_solved) == 0;
        // ---- x-update: y = (-H) w, x = c - y ----
        {
            const T* wp = wv + 64 * h;
            T a0 = T(0), a1 = T(0);
#pragma unroll
            for (int c = 0; c < 64; c += 8) {
                const V4<T> q0 = *(const V4<T>*)(wp + c), q1 = *(const V4<T>*)(wp + c + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { a0 += hreg[c + k] * q0.v[k]; a1 += hreg[c + 4 + k] * q1.v[k]; }
            }
            T acc = a0 + a1;
            acc += dpp<0xB1>(acc);                               // the two halves of a row sit in adjacent lanes
            if (h == 0 && r < 128) yv[r] = acc;
        }
        if ((check || it + 1 == it1) && m > 0) {                 // nu = T^T w - s0 (one wave per row) while wv still holds w
            for (int q = w; q < m; q += NT / 64) {
                T acc = T(0);
                for (int i = lane; i < n; i += 64) acc += V.Tm[(size_t)q * n + i] * wv[i];
                acc = wave_sum(acc);
                if (lane == 0) nus_l[q] = acc - V.s0[q];
            }
        }
        __syncthreads();
        T mx[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) mx[q] = T(0);
        if (live) {
            const T wi = wv[e];
            xi = cvi - yv[e];
            const T zp = zi;
            T zn = xi + ui;
            zn = tmin(tmax(zn, lbi), ubi);                       // (:273-276; an infinite bound is a no-op)
            const T rr_ = xi - zn;
            const T ss = rho * (zn - zp);
            const T un = ui + rr_;
            zi = zn;
            ui = un;
            if (check) {
                mx[0] = tabs(di * rr_);
                mx[1] = tabs(di * ss);
                mx[2] = tabs(di * xi);
                mx[3] = tabs(di * zn);
                mx[4] = tabs((rho * di) * un);
                T qx = wi - rho * xi;                            // Qs x = w - rho x - As^T nu (see admm_loop_body)
                for (int q = 0; q < m; ++q) qx -= V.As[(size_t)q * n + e] * nus_l[q];
                mx[5] = tabs(qx / di);
            }
        }
        __syncthreads();                                         // (everybody has read w and y)
        if (tid < 128) wv[tid] = live ? -psi + rho * (zi - ui) : T(0);     // next right-hand side
        if (check) {
            T mv[6] = {mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]};
            wg_max_n<T, 6, NT / 64>(mv, red);
            const T tiny = T(1e-16);
            const T pri_scale = tmax(tmax(mv[2], mv[3]), tiny);
            const T tol_p = P.eps_abs + P.eps_rel * pri_scale;
            const T dua_scale = tmax(tmax(tmax(mv[4], mv[5]), pnorm), tiny);
            const T tol_d = P.eps_abs + P.eps_rel * dua_scale;
            const bool solved = (mv[0] < tol_p) && (mv[1] < tol_d);
            const bool wants = (mv[0] > tmax(tol_p, P.ar_thr)) || (mv[1] > tmax(tol_d, P.ar_thr));
            const T num = tmax(mv[0] / pri_scale, tiny);
            const T den = tmax(mv[1] / dua_scale, tiny);
            const T ratio = tsqrt(num / den);
            const bool trig = (ratio > P.ar_tol) || (ratio < P.ar_inv_tol);
            unsigned int* ct = P.counters + (size_t)slot * CT_WORDS;
            if (tid == 0) {
                scal[SC_RATIO] = ratio;
                scal[SC_WANTS] = wants ? T(1) : T(0);
                scal[SC_PRI] = mv[0];
                scal[SC_DUA] = mv[1];
                trace_check(P.vtrace, it, P.check_solved, P.ring, mv[0], mv[1]);
                unsigned int r1 = 0, r2 = 0;
                if (wants) r1 = atomicAdd(ct + CT_WANTS, 1u);
                if (trig) r2 = atomicAdd(ct + CT_TRIG, 1u);
                asm volatile("s_waitcnt vmcnt(0)" :: "v"(r1), "v"(r2) : "memory");
                __hip_atomic_fetch_add((unsigned long long*)(ct + CT_NOTOPT), (solved ? 0ull : 1ull) | (1ull << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ++slot;
            grid_wait(ct + CT_ARRIVE, gridDim.x, P.status);      // device-wide "all optimal?" (torch.all at :312)
            const unsigned int notopt = __hip_atomic_load(ct + CT_NOTOPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int tmo = __hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (notopt == 0 || tmo) {
                if (b == 0 && tid == 0) {
                    P.status[ST_FINAL_ITER] = it;
                    __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
        }
        __syncthreads();
    }
    // ---- state for the continuation launch / the epilogue ----
    if (live) { V.z[e] = zi; V.u[e] = ui; V.x[e] = xi; }
    __syncthreads();
    for (int q = tid; q < m; q += NT) V.nu[q] = nus_l[q];
}

// NP = 4 (batches up to a quarter of the CUs): one column pair per workgroup, every partial product published once and
// fetched by the three others; the sum runs over the parts in their order on every workgroup (identical iterates).
template <int KS, int NT, bool DBG = false, int NP = 2>
__global__ __launch_bounds__(NT) void k_admm_loop_split(const FwdParams<float> P, const int it0_in, const int it1,
                                                        const int ctr_base_in) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    typedef float T;
    int it0 = it0_in, ctr_base = ctr_base_in;
    if (P.hot_resume) {            // (a later round of the hot loop: it goes on where the round before stopped)
        const int r = __builtin_amdgcn_readfirstlane(P.status[ST_RESUME]);
        if (r <= 0) return;
        it0 = r;
        ctr_base = ((r + P.check_solved - 1) / P.check_solved) % P.ring;
    }
    constexpr int NWV = NT / 64, Ks = KS, Nps = KS * LQP_NB, rl = split_lds_blocks<NT, NP>(KS);
    constexpr int XPART = SPD_MAXK * LQP_NB, XPAR = NP * XPART;      // granules of one part / of one parity of the exchange
    // (split_seg: NP == 2)
    int b = 8 * ((int)blockIdx.x >> 4) + ((int)blockIdx.x & 7), part_id = ((int)blockIdx.x >> 3) & 1;      // (split_seg: shared_map with NP = 2)
    if (P.split_seg ? b >= P.B : !shared_map((int)blockIdx.x, P.B, NP, b, part_id)) return;
    const int n = P.n, m = P.m;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (__hip_atomic_load(P.status + ST_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (P.split_seg && P.seg_prev_slot >= 0) {
        // every problem was optimal at the last check (a launch enqueued ahead of that knowledge): as admm_loop_body
        if (__hip_atomic_load(P.counters + (size_t)P.seg_prev_slot * CT_WORDS + CT_NOTOPT, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (blockIdx.x == 0 && tid == 0) {
                P.status[ST_FINAL_ITER] = it0 - 1;
                __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    if (it0 >= it1) return;
    T* const lds_res = (T*)smem;
    T* const v = lds_res + (size_t)rl * LQP_BLK;
    T* const yrow = v + Nps;
    T* const cvl = yrow + Nps;
    T* const part = cvl + Nps;
    T* const z = part + (size_t)NWV * Nps;
    T* const u = z + Nps;
    T* const ps = u + Nps;
    T* const lb = ps + Nps;
    T* const ub = lb + Nps;
    T* const D = ub + Nps;
    T* const xs = D + Nps;
    T* const sz = xs + Nps;                                  // snapshot of (z, u, x) at the last check
    T* const su = sz + Nps;
    T* const sx = su + Nps;
    T* const red = sx + Nps;
    int* const flags = (int*)(red + NWV * 8 + 8);             // [0] exchange timed out (sticky), [1] verdict of this iteration
    T* const Gl = (T*)(flags + 8);                            // equality block: G = K^-1 As^T, T = G S^-1 (m x Nps each)
    T* const Tl = Gl + (size_t)m * Nps;
    T* const Asl = Tl + (size_t)m * Nps;                      // the scaled equality rows (read at every check)
    T* const Sm = Asl + (size_t)m * Nps;
    T* const Si = Sm + m * m;
    T* const s0l = Si + m * m;
    T* const bs = s0l + m;
    T* const nus_l = bs + m;
    T* const snu = nus_l + m;

    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T* scal = P.scal + (size_t)b * SC_WORDS;
    const T* packed = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
    unsigned long long* xq = P.xchg + (size_t)b * XCHG_WORDS;
    const T rho = scal[SC_RHO];
    const T pnorm = scal[SC_PNORM];

    SplitResident<NT> rr;
    // (one instantiation of the block helpers per part: which blocks a workgroup holds is a compile-time fact)
#define LQP_BY_PART(CALL)                                                                   \
    do {                                                                                    \
        if (part_id == 0) { constexpr int PARTC = 0; CALL; }                                \
        else if (part_id == 1) { constexpr int PARTC = 1; CALL; }                           \
        else if constexpr (NP > 2) {                                                        \
            if (part_id == 2) { constexpr int PARTC = 2; CALL; }                            \
            else { constexpr int PARTC = 3; CALL; }                                         \
        }                                                                                   \
    } while (0)
    LQP_BY_PART((split_resident_load<KS, PARTC, NT, NP>(rr, lds_res, packed)));
    for (int i = tid; i < Nps; i += NT) {
        const bool in = i < n;
        z[i] = in ? V.z[i] : T(0); u[i] = in ? V.u[i] : T(0); ps[i] = in ? V.ps[i] : T(0);
        lb[i] = in ? V.lbs[i] : T(0); ub[i] = in ? V.ubs[i] : T(0); D[i] = in ? V.D[i] : T(1);
        cvl[i] = (in && m > 0) ? V.cv[i] : T(0);
        yrow[i] = T(0);                                       // rows / columns of the partner stay zero
    }
    for (int r = tid; r < m; r += NT) bs[r] = V.bs[r];
    for (int i = tid; i < m * Nps; i += NT) { const int q = i / Nps, e = i - q * Nps; Asl[i] = e < n ? V.As[(size_t)q * n + e] : T(0); }
    for (int i = tid; i < NWV * Nps; i += NT) part[i] = T(0);
    if (tid < 8) flags[tid] = 0;
    // Which XCD are the workgroups of this QP on?  Each announces its id (write-through store, at once) and reads the
    // others' here, a whole load phase later.  On ONE XCD its L2 is their point of coherence: the granules are then stored
    // with workgroup scope (they stay in that L2; an sc1 store drops the line and the reader goes to memory for it) and
    // read as before (sc1 loads bypass the reader's L1 only).  Placement is the dispatcher's: never assumed, always asked.
    unsigned long long* const xcw = P.xchg + (size_t)P.B * XCHG_WORDS + (size_t)XCHG_TAIL * b + 8;
    const unsigned int xcd_me = my_xcd();
    // (tagged with the launch's first iteration: a problem sees one launch of this kernel per check segment when the batch takes
    //  turns on the chip, and the dispatcher is free to place every one of them differently)
    const unsigned long long ann = (unsigned long long)(it0 + 1) << 16;
    if (tid == 0) __hip_atomic_store(xcw + part_id, ann | 0x100ull | xcd_me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        int same = P.xcd_local;
        for (int pp = 0; pp < NP && same; ++pp) {
            if (pp == part_id) continue;
            unsigned long long g = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (((g = __hip_atomic_load(xcw + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~0xFFFFull) != ann) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 50000000ULL) { g = ~0ull; break; }      // (0.5 s: the exchange below will flag it)
            }
            same = (unsigned int)(g & 0xFFull) == xcd_me;
        }
        flags[2] = same;
    }
    __syncthreads();
    const bool xlocal = flags[2] != 0;
    auto xstore = [&](unsigned long long* ptr, const unsigned long long val) {
        if (xlocal) __hip_atomic_store(ptr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(ptr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- equality constraints: H <- H + T G^T with G = K^-1 As^T, S = As G, T = G S^-1 (what wg_eq_correct does to the
    //      blocks in global memory, three passes over them in a launch of its own) applied to the blocks in REGISTERS:
    //      m products with the blocks this kernel holds anyway, the partner's half through the exchange granules.  The
    //      corrected blocks only go back to global memory if a continuation launch needs them (end of the kernel). ----
    const bool eq_here = m > 0 && P.eq_in_loop;
    const int moff = eq_here ? m : 0;                         // the iterations' exchanges follow these m in buffer parity
    if (m > 0 && !eq_here) {
        for (int i = tid; i < m * Nps; i += NT) { const int q = i / Nps, e = i - q * Nps; Tl[i] = e < n ? V.Tm[(size_t)q * n + e] : T(0); }
        for (int r = tid; r < m; r += NT) s0l[r] = V.s0[r];
    }
    if (eq_here) {
        if (part_id == 0 && tid == 0 && P.info[b] != 0) P.status[ST_NOTSPD] = 1;      // (k_spd_end did not run)
        for (int q = 0; q < m; ++q) {
            LQP_BY_PART((wg_sym_gemv_split<KS, PARTC, NT, NP>(rr, lds_res, Nps, Asl + (size_t)q * Nps, yrow, part)));
            wg_barrier_lds();
            if (tid < Nps) {
                const int i = tid;
                const T own = split_combine<NT>(i, Nps, yrow, part);
                const unsigned int tag = 0x20000000u + (unsigned int)q;
                unsigned long long* base = xq + (size_t)(q & 1) * XPAR;
                xstore(base + (size_t)part_id * XPART + i,
                       ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned int, own));
                T y = T(0);
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {                 // same order on every workgroup
                    T term = own;
                    if (pp != part_id) {
                        const unsigned long long* src = base + (size_t)pp * XPART + i;
                        unsigned long long g = 0;
                        if (!flags[0]) {
                            unsigned int spins = 0;
                            unsigned long long t0 = 0;
                            for (;;) {
                                g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (((unsigned int)(g >> 32) & 0x3FFFFFFFu) == tag) break;
                                if ((++spins & 1023u) == 0) {
                                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz
                                    if (t0 == 0) t0 = now;
                                    else if (now - t0 > 50000000ULL) {                                   // 0.5 s: give up
                                        __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        flags[0] = 1;
                                        break;
                                    }
                                }
                            }
                        }
                        term = __builtin_bit_cast(float, (unsigned int)g);
                    }
                    y = pp == 0 ? term : y + term;
                }
                Gl[(size_t)q * Nps + i] = -y;
            }
            wg_barrier_lds();
        }
        // S = As G (m x m), one wave per entry
        for (int t = w; t < m * m; t += NWV) {
            const int q = t / m, q2 = t - q * m;
            T acc = T(0);
            for (int i = lane; i < n; i += 64) acc += Asl[(size_t)q * Nps + i] * Gl[(size_t)q2 * Nps + i];
            acc = wave_sum(acc);
            if (lane == 0) Sm[t] = acc;
        }
        wg_barrier_lds();
        {                                                      // S^-1 by Gauss-Jordan (S is SPD: no pivoting), m <= 16
            const int bad = wg_gj_inverse_spd(Sm, Si, m, [] { wg_barrier_lds(); });
            if (bad && tid == 0) { if (P.info[b] == 0) P.info[b] = Ks * 64 + 1; P.status[ST_NOTSPD] = 1; }   // A rank deficient
        }
        wg_barrier_lds();
        // T = G S^-1, c = T b, s0 = S^-1 b (both workgroups hold them; the copies in global memory are for later launches)
        for (int t = tid; t < m * Nps; t += NT) {
            const int q = t / Nps, e = t - q * Nps;
            T acc = T(0);
            for (int q2 = 0; q2 < m; ++q2) acc += Gl[(size_t)q2 * Nps + e] * Si[q2 * m + q];
            Tl[t] = acc;
            if (part_id == 0 && e < n) V.Tm[(size_t)q * n + e] = acc;
        }
        for (int q = tid; q < m; q += NT) {
            T acc = T(0);
            for (int q2 = 0; q2 < m; ++q2) acc += Si[q * m + q2] * bs[q2];
            s0l[q] = acc;
            if (part_id == 0) V.s0[q] = acc;
        }
        wg_barrier_lds();
        for (int e = tid; e < Nps; e += NT) {
            T acc = T(0);
            for (int q = 0; q < m; ++q) acc += Tl[(size_t)q * Nps + e] * bs[q];
            cvl[e] = e < n ? acc : T(0);
            if (part_id == 0 && e < n) V.cv[e] = acc;
        }
        // the blocks: thread t holds EPT consecutive elements of row t / LPR of every block
        LQP_BY_PART((split_eq_update<KS, PARTC, NT, NP>(rr, lds_res, Gl, Tl, m, Nps)));
        wg_barrier_lds();
    }
    for (int i = tid; i < Nps; i += NT) v[i] = (i < n) ? -ps[i] + rho * (z[i] - u[i]) : T(0);
    wg_barrier_lds();

    int slot = ctr_base;
    bool pending = false;                                     // a check's verdict is still open (uniform)
    int pend_it = 0;
    const unsigned long long* pend_word = nullptr;
    unsigned long long dbt[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0;  // debug: cycles of wave 0 (part 0) per phase
    const bool dbg_on = DBG && P.dbg != nullptr && part_id == 0;
    // 64-bit word {low: problems not optimal, high: arrivals} of a check -> 0 unknown, 1 all optimal, 2 go on
    auto verdict_of = [&](const unsigned long long cw) -> int {
        if ((unsigned int)(cw >> 32) < (unsigned int)P.B) return 0;
        return (unsigned int)cw == 0u ? 1 : 2;
    };
    auto wait_verdict = [&]() -> int {                        // (one thread) bounded spin, 2 s
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const int vd = verdict_of(__hip_atomic_load(pend_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (vd) return vd;
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {
                __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 1;
            }
        }
    };
    auto leave_with_snapshot = [&]() {                        // every problem was optimal at iteration pend_it
        if (blockIdx.x == 0 && tid == 0) {
            P.status[ST_FINAL_ITER] = pend_it;
            __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (part_id == 0) {
            for (int i = tid; i < n; i += NT) { V.z[i] = sz[i]; V.u[i] = su[i]; V.x[i] = sx[i]; }
            for (int r = tid; r < m; r += NT) V.nu[r] = snu[r];
        }
    };

    // One iteration.  COLD = std::true_type: a check iteration or the last one of the launch (nu, the six norms, the
    // counters, the snapshot, a blocking wait for an open verdict); std::false_type: everything else -- the hot
    // variant carries none of that code, and the hot iterations run in an inner loop of their own below, so the
    // register allocator keeps the resident blocks (and everything else the product needs) out of scratch there.
    // Returns 1 when the workgroup is done (all problems were optimal at the last check).
    auto iterate = [&](auto cold_tag, const int it, const bool check) -> int {
        constexpr bool COLD = decltype(cold_tag)::value;
        if (dbg_on) dt0 = clock64();
        // ---- part 0: look at the open verdict while the product runs ----
        unsigned long long cw = 0;
        const bool look = pending && part_id == 0 && tid == 0;
        if (look) cw = __hip_atomic_load(pend_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        LQP_BY_PART((wg_sym_gemv_split<KS, PARTC, NT, NP>(rr, lds_res, Nps, v, yrow, part)));
        if (look) {
            int vd = verdict_of(cw);
            if constexpr (COLD) { if (vd == 0) vd = wait_verdict(); }
            flags[1] = vd;
        }
        wg_barrier_lds();
        if (dbg_on) { const unsigned long long t = clock64(); dbt[0] += t - dt0; dt0 = t; }
        int verdict = (pending && part_id == 0) ? flags[1] : 0;
        if constexpr (COLD) {
            if (m > 0) {                                      // nu = T^T w - s0 while v is still w
                for (int r = w; r < m; r += NWV) {
                    T acc = T(0);
                    for (int i = lane; i < n; i += 64) acc += Tl[(size_t)r * Nps + i] * v[i];
                    acc = wave_sum(acc);
                    if (lane == 0) nus_l[r] = acc - s0l[r];
                }
                wg_barrier_lds();
            }
        }
        T xi = T(0);
        if (tid < Nps) {
            const int i = tid;
            const T own = split_combine<NT>(i, Nps, yrow, part);
            // ---- exchange: publish this element's partial (part 0: with the verdict), fetch the partner's ----
            const unsigned int tag = (unsigned int)(it + 1);
            unsigned long long* base = xq + (size_t)((it + moff) & 1) * XPAR;
            xstore(base + (size_t)part_id * XPART + i,
                   ((unsigned long long)(tag | ((unsigned int)verdict << 30)) << 32) |
                       (unsigned long long)__builtin_bit_cast(unsigned int, own));
            if (dbg_on) { const unsigned long long t = clock64(); dbt[1] += t - dt0; dt0 = t; }
            T y = T(0);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {                     // same order on every workgroup
                T term = own;
                if (pp != part_id) {
                    const unsigned long long* src = base + (size_t)pp * XPART + i;
                    unsigned long long g = 0;
                    if (!(verdict == 1 && part_id == 0) && !flags[0]) {      // (part 0 leaving: nothing to fetch)
                        unsigned int spins = 0;
                        unsigned long long t0 = 0;
                        for (;;) {
                            g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (((unsigned int)(g >> 32) & 0x3FFFFFFFu) == tag) break;
                            if ((++spins & 1023u) == 0) {
                                const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz
                                if (t0 == 0) t0 = now;
                                else if (now - t0 > 50000000ULL) {                                   // 0.5 s: give up
                                    __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    flags[0] = 1;
                                    break;
                                }
                            }
                        }
                    }
                    if (pp == 0 && pending) {                     // the verdict travels in part 0's tags
                        verdict = (int)((unsigned int)(g >> 62));
                        if (tid == 0) flags[1] = verdict;
                    }
                    term = __builtin_bit_cast(float, (unsigned int)g);
                }
                y = pp == 0 ? term : y + term;
            }
            if (dbg_on) { const unsigned long long t = clock64(); dbt[2] += t - dt0; dt0 = t; }
            xi = cvl[i] - y;
        }
        if (pending) {
            // the verdict is uniform over both workgroups: part 0 read it before the exchange, part 1 found it in the tags
            if (part_id != 0) { wg_barrier_lds(); verdict = flags[1]; }
            if (verdict == 1) return 1;
            if (verdict == 2) pending = false;
        }
        T mx[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) mx[q] = T(0);
        if (tid < Nps) {
            const int i = tid;
            xs[i] = xi;
            const T zp = z[i];
            const T ui = u[i];
            T zn = xi + ui;
            zn = tmin(tmax(zn, lb[i]), ub[i]);
            const T r = xi - zn;
            const T s = rho * (zn - zp);
            const T un = ui + r;
            const bool in = i < n;
            if (in) { z[i] = zn; u[i] = un; }
            if constexpr (COLD) {
                if (check) {
                    sz[i] = zn; su[i] = un; sx[i] = xi;      // snapshot for a late "all optimal"
                    if (in) {
                        const T di = D[i];
                        mx[0] = tabs(di * r);
                        mx[1] = tabs(di * s);
                        mx[2] = tabs(di * xi);
                        mx[3] = tabs(di * zn);
                        mx[4] = tabs((rho * di) * un);
                        T qx = v[i] - rho * xi;
                        for (int q = 0; q < m; ++q) qx -= Asl[(size_t)q * Nps + i] * nus_l[q];
                        mx[5] = tabs(qx / di);
                    }
                }
            }
            v[i] = in ? -ps[i] + rho * (zn - un) : T(0);     // next iteration's right-hand side
        }
        if (dbg_on) { const unsigned long long t = clock64(); dbt[3] += t - dt0; dt0 = t; }
        if constexpr (COLD) {
            if (check) {
                for (int r = tid; r < m; r += NT) snu[r] = nus_l[r];
                // six inf-norms: per wave by DPP, then ONE thread folds the 16 wave results
#pragma unroll
                for (int q = 0; q < 6; ++q) mx[q] = wave_max(mx[q]);
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) red[w * 8 + q] = mx[q];
                }
                __syncthreads();
                if (part_id == 0 && tid == 0) {               // (the partner computed the very same numbers)
                    T mv[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) mv[q] = red[q];
#pragma unroll 1
                    for (int ww = 1; ww < NWV; ++ww) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) mv[q] = tmax(mv[q], red[ww * 8 + q]);
                    }
                    const T tiny = T(1e-16);
                    const T pri_scale = tmax(tmax(mv[2], mv[3]), tiny);
                    const T tol_p = P.eps_abs + P.eps_rel * pri_scale;
                    const T dua_scale = tmax(tmax(tmax(mv[4], mv[5]), pnorm), tiny);
                    const T tol_d = P.eps_abs + P.eps_rel * dua_scale;
                    const bool solved = (mv[0] < tol_p) && (mv[1] < tol_d);
                    const bool wants = (mv[0] > tmax(tol_p, P.ar_thr)) || (mv[1] > tmax(tol_d, P.ar_thr));
                    const T num = tmax(mv[0] / pri_scale, tiny);
                    const T den = tmax(mv[1] / dua_scale, tiny);
                    const T ratio = tsqrt(num / den);
                    const bool trig = (ratio > P.ar_tol) || (ratio < P.ar_inv_tol);
                    unsigned int* ct = P.counters + (size_t)slot * CT_WORDS;
                    scal[SC_RATIO] = ratio;
                    scal[SC_WANTS] = wants ? T(1) : T(0);
                    scal[SC_PRI] = mv[0];
                    scal[SC_DUA] = mv[1];
                    trace_check(P.vtrace, it, P.check_solved, P.ring, mv[0], mv[1]);
                    if (wants) atomicAdd(ct + CT_WANTS, 1u);
                    if (trig) atomicAdd(ct + CT_TRIG, 1u);
                    atomicAdd((unsigned long long*)ct, (1ull << 32) | (solved ? 0ull : 1ull));   // {not optimal, arrival}
                }
                pend_word = (const unsigned long long*)(P.counters + (size_t)slot * CT_WORDS);
                pend_it = it;
                pending = !P.split_seg;        // (one launch per check segment: the verdict is the next launch's / k_check_done's)
                ++slot;
            }
        }
        if (dbg_on) { const unsigned long long t = clock64(); dbt[4] += t - dt0; dt0 = t; }
        wg_barrier_lds();
        if (dbg_on) { const unsigned long long t = clock64(); dbt[5] += t - dt0; }
        return 0;
    };

    int left = 0;
    int it = it0;
    for (; it < it1 && !left;) {
        if (P.hot_past && P.adaptive_rho && it > it0 && it % P.ar_iter == 0 && it < P.ar_max) {
            // An iteration at which the reference may adapt rho (:237-246).  Whether anything changes is decided by the check
            // BEFORE it -- any(do_rho_update) and the ratio test, both over the whole batch: the counters of that check, complete
            // once every problem has arrived.  Nothing to update: run on in this kernel (a solve that never adapts rho stays on the
            // register-resident loop: 3 us per iteration instead of the continuation kernel's 14).  Otherwise leave; the
            // continuation launch takes over at this iteration (status[ST_RESUME]) and begins with the event.
            if (tid == 0) {
                const unsigned int* ce = P.counters + (size_t)(((it - 1) / P.check_solved) % P.ring) * CT_WORDS;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned long long cw;
                while ((unsigned int)((cw = __hip_atomic_load((const unsigned long long*)ce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) < (unsigned int)P.B) {
                    __builtin_amdgcn_s_sleep(4);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ULL) {      // 2 s: give up, results are flagged
                        __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                const bool fire = (unsigned int)cw != 0u &&
                                  __hip_atomic_load(ce + CT_WANTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 &&
                                  __hip_atomic_load(ce + CT_TRIG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
                flags[3] = fire ? 1 : 0;
            }
            __syncthreads();
            if (flags[3]) break;
        }
        const bool check = (it % P.check_solved) == 0;
        if (check || it + 1 == it1) {
            left = iterate(std::true_type(), it, check);
            ++it;
        } else {
            int e = (it / P.check_solved + 1) * P.check_solved;      // next special iteration: a check or the last one
            if (e > it1 - 1) e = it1 - 1;
#pragma unroll 1
            for (; it < e; ++it) {
                left = iterate(std::false_type(), it, false);
                if (left) break;
            }
        }
    }
    if (left) {
        leave_with_snapshot();
        if (dbg_on && tid == 0)
            for (int q = 0; q < 6; ++q) P.dbg[(size_t)b * 8 + q] += dbt[q];
        return;
    }
    if (dbg_on && tid == 0)
        for (int q = 0; q < 6; ++q) P.dbg[(size_t)b * 8 + q] += dbt[q];
    // ---- end of the launch: a verdict still open is for the state we hold (the check ran in the last iteration) ----
    if (pending && blockIdx.x == 0 && tid == 0) {
        if (wait_verdict() == 1 && !__hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            P.status[ST_FINAL_ITER] = pend_it;
            __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (P.hot_past && blockIdx.x == 0 && tid == 0) P.status[ST_RESUME] = it;      // (where the continuation launch goes on)
    if (part_id == 0) {
        for (int i = tid; i < n; i += NT) { V.z[i] = z[i]; V.u[i] = u[i]; V.x[i] = xs[i]; }
        for (int r = tid; r < m; r += NT) V.nu[r] = nus_l[r];
    }
    // the loop goes on in a continuation launch, which reads the blocks from global memory: there they still lack the
    // equality correction
    if (eq_here) {
        T* packed_w = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
        LQP_BY_PART((split_resident_store<KS, PARTC, NT, NP>(rr, lds_res, packed_w)));
#undef LQP_BY_PART
    }
}

// all problems optimal at the check held in `slot` (iteration `it_check`)?  -> DONE
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ void k_check_done(int* status, const unsigned int* counters, const int slot, const int it_check) {
    if (threadIdx.x == 0 && status[ST_DONE] == 0 && counters[(size_t)slot * CT_WORDS + CT_NOTOPT] == 0) {
        status[ST_FINAL_ITER] = it_check;
        status[ST_DONE] = 1;
    }
}

// ---------------------------------------------------------------------------
// adaptive rho (:237-256): global decision from the counters of the last
// check, masked per-problem update, KKT re-assembly; sets the gate for the
// LU + pack launches that follow.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_rho_update(const FwdParams<T> P, int last_slot, const int max_it = 0) {
    const int b = blockIdx.x, n = P.n, m = P.m;
    if (__hip_atomic_load(P.status + ST_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (b == 0 && threadIdx.x == 0) P.status[ST_GATE] = 0;
        return;
    }
    if (last_slot < 0) {
        // the event iteration is where the hot loop stopped (FwdParams::hot_past): an event only if it left FOR one
        const int r = P.status[ST_RESUME];
        if (!(r > 0 && r % P.ar_iter == 0 && r < P.ar_max && r < max_it)) {
            if (b == 0 && threadIdx.x == 0) P.status[ST_GATE] = 0;
            return;
        }
        last_slot = ((r - 1) / P.check_solved) % P.ring;
    }
    const unsigned int* ct = P.counters + (size_t)last_slot * CT_WORDS;
    const bool all_done = ct[CT_NOTOPT] == 0;
    const bool fire = !all_done && ct[CT_WANTS] > 0 && ct[CT_TRIG] > 0;
    if (b == 0 && threadIdx.x == 0) {
        P.status[ST_GATE] = fire ? 1 : 0;
        if (fire) { P.status[ST_NFACTOR] += 1; P.status[ST_RHO_UPDATED] = 1; }
    }
    if (!fire) return;
    T* scal = P.scal + (size_t)b * SC_WORDS;
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T rho = scal[SC_RHO];
    if (scal[SC_WANTS] != T(0)) rho = rho * scal[SC_RATIO];
    rho = tmin(tmax(rho, P.rho_min), P.rho_max);
    __syncthreads();
    if (threadIdx.x == 0) scal[SC_RHO] = rho;
    if (P.spd) return;
    const bool lazy = P.scale && P.qs_lazy;
    const T* Qs = (P.scale && !lazy) ? (P.Qs + (size_t)b * n * P.ldq) : (P.Q + (size_t)b * n * n);
    assemble_kkt_rows(P, b, Qs, (P.scale && !lazy) ? P.ldq : n, V, rho, true, lazy ? V.D : nullptr);
}

// primal / dual error of the last convergence check, per problem (lqp_boxqp_last_residuals)
// the check trace of a verbose solve (trace_check): bit patterns -> float32 in the caller's buffer
template <int LQP_ANY = 0>
__global__ void k_copy_trace(const unsigned int* __restrict__ vtrace, float* __restrict__ out, const int words) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) out[i] = __uint_as_float(vtrace[i]);
}
template <typename T>
__global__ void k_copy_residuals(const T* __restrict__ scal, T* __restrict__ pri, T* __restrict__ dua, const int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        if (pri) pri[b] = scal[(size_t)b * SC_WORDS + SC_PRI];
        if (dua) dua[b] = scal[(size_t)b * SC_WORDS + SC_DUA];
    }
}

// ---------------------------------------------------------------------------
// epilogue: undo scaling, duals (:316-327)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fwd_epilogue(const FwdParams<T> P) {
    fwd_finish(P, blockIdx.x);
}

// ---------------------------------------------------------------------------
// fixed-point backward (solve_box_qp_admm_torch.py:349-432)
// ---------------------------------------------------------------------------
template <typename T> struct BwdParams {
    int B, n, m, N, Np, K;
    const T *g, *x, *u, *lams, *nus, *Q, *A, *lb, *ub, *rho_in;
    T rho_value;
    int rho_mode;
    T *dQ, *dp, *dA, *db, *dlb, *dub;
    T* M;        // B * Np * Np
    T* packed;
    T* rhs;      // B * Np  (rhs, then the solution d)
    int *piv, *dest, *info;
    int* fidx;   // B * n : indices of the free variables (reduced system), in order
    int* nred;   // B     : size of the reduced system = #free + m
    int reduced; // 1: solve on the free set only
    int chol;    // 1: the reduced system is solved through a blocked Cholesky of Q_FF (f32, symmetric Q)
    T* rhs2;     // B * Np: residual / correction of the one refinement step of the LU form (or null)
    int refine;  // 1: the epilogue adds rhs2 to rhs
    unsigned long long* dbg;   // optional cycle counters (8 per problem), debug only
    int la_maxk;               // largest block count the look-ahead Cholesky may take (LDS of the launch), 0 = off
    T* bsc;                    // Cholesky form: [B][Np] the power-of-two equilibration of the free-set block, S (Q_FF + eps I) S (0 = off)
    int* host_report;          // optional pinned host memory, B ints: the epilogue leaves every problem's info word there
    int early_report;          // 1: Cholesky form -- k_bwd_chol_solve reports right behind its factorisation, not the epilogue
    int reported;              // 1 (phase 2 only): the phase-1 call has stored the info words into host_report already
    int lu_reported;           // 1: LU form -- k_report_info has stored them right behind the factorisation, the epilogue does not
    int phase;                 // Cholesky form in two calls: 1 = free set + Q_FF + its factorisation only (no cotangent needed:
                               // enqueued right behind the forward), 2 = the solves + epilogue on that factor; 0 = everything
    int kkt;                   // 1: the KKT-system backward (backward='kkt', reference :435-584) on the same kernels: the (3n+m)
                               //    system [[Q, G^T diag(lam), A^T], [G, -diag(s), 0], [A, 0, 0]] with G = [-I; I] reduces exactly
                               //    (dlam = diag(1/s) G dx) to [[Q + diag(w), A^T], [A, 0]] [dx; dnu] = [-g; 0],
                               //    w = lam_lo / s_lo + lam_hi / s_hi (both clamped at 1e-8, :450-452): every variable is "free",
                               //    w takes the place of the 1e-8 regulariser, and the epilogue forms dlb / dub from dlam
};

// diagonal weight of the KKT-system backward for variable i (see BwdParams::kkt)
template <typename T>
__device__ __forceinline__ T kkt_weight(const BwdParams<T>& P, const int b, const int i) {
    const int n = P.n;
    const T xi = P.x[(size_t)b * n + i];
    const T slo = tmax(xi - P.lb[(size_t)b * n + i], T(1e-8)), shi = tmax(P.ub[(size_t)b * n + i] - xi, T(1e-8));
    const T llo = tmax(P.lams[(size_t)b * 2 * n + i], T(1e-8)), lhi = tmax(P.lams[(size_t)b * 2 * n + n + i], T(1e-8));
    return llo / slo + lhi / shi;
}

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_bwd_build(const BwdParams<T> P) {
    // mask :360-365, rhs :368-375, non-symmetric system :378-392
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const T rho = (P.rho_mode == 2) ? P.rho_in[b] : P.rho_value;
    const T* x = P.x + (size_t)b * n;
    const T* u = P.u + (size_t)b * n;
    const T* lb = P.lb + (size_t)b * n;
    const T* ub = P.ub + (size_t)b * n;
    const T* g = P.g + (size_t)b * n;
    const T* Q = P.Q + (size_t)b * n * n;
    const T* A = P.A ? P.A + (size_t)b * m * n : nullptr;
    T* M = P.M + (size_t)b * Np * Np;
    T* rhs = P.rhs + (size_t)b * Np;
    if (tid == 0) P.info[b] = 0;
    const bool qvec = (n % 4 == 0) && ((((uintptr_t)Q) % sizeof(V4<T>)) == 0);
#pragma unroll 2
    for (int i = w; i < n; i += LQP_NW) {
        const T s = x[i] + u[i];
        const T keep = (s > ub[i] || s < lb[i]) ? T(0) : T(1);
        const T* qr = Q + (size_t)i * n;
        T* mr = M + (size_t)i * Np;
        if (qvec) {
            for (int j = lane * 4; j < n; j += 256) {
                V4<T> v = *(const V4<T>*)(qr + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    T val = keep * v.v[e];
                    if (j + e == i) val = (val + rho * (T(1) - keep)) + T(1e-8);
                    v.v[e] = val;
                }
                *(V4<T>*)(mr + j) = v;
            }
        } else {
            for (int j = lane; j < n; j += 64) {
                T val = keep * qr[j];
                if (j == i) val = (val + rho * (T(1) - keep)) + T(1e-8);
                mr[j] = val;
            }
        }
        for (int r = lane; r < m; r += 64) mr[n + r] = keep * A[(size_t)r * n + i];
        if (lane == 0) rhs[i] = -(g[i] * keep);
    }
    for (int r = w; r < m; r += LQP_NW) {
        T* mr = M + (size_t)(n + r) * Np;
        for (int j = lane; j < n; j += 64) mr[j] = A[(size_t)r * n + j];
        for (int c = lane; c < m; c += 64) mr[n + c] = (c == r) ? T(1e-8) : T(0);
        if (lane == 0) rhs[n + r] = T(0);
    }
}

// Reduced fixed-point system.  For an active bound (keep_i = 0) row i of the reference's system
// (:378-392) reads (rho + 1e-8) dv_i = 0, so dv_i = 0 exactly and the variable drops out of every
// other row: what is left is [[Q_FF, A_F^T], [A_F, 0]] (+1e-8 I) [dv_F; dnu] = [-g_F; 0] on the
// free set F.  Same solution, (|F|+m)^3 instead of (n+m)^3 work.  LDS: fl[n] (int) | wtot[NW]
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_bwd_build_reduced(const BwdParams<T> P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int* fl = (int*)smem;               // free list
    int* wtot = fl + round_up(n, 8);
    const T* x = P.x + (size_t)b * n;
    const T* u = P.u + (size_t)b * n;
    const T* lb = P.lb + (size_t)b * n;
    const T* ub = P.ub + (size_t)b * n;
    const T* g = P.g + (size_t)b * n;
    const T* Q = P.Q + (size_t)b * n * n;
    const T* A = P.A ? P.A + (size_t)b * m * n : nullptr;
    T* M = P.M + (size_t)b * Np * Np;
    T* rhs = P.rhs + (size_t)b * Np;
    if (tid == 0 && blockIdx.y == 0) P.info[b] = 0;
    // ---- ordered compaction of the free set: one variable per thread and pass (n <= 1024: one pass) ----
    int nf = 0;
    for (int i0 = 0; i0 < n; i0 += LQP_NT) {
        const int i = i0 + tid;
        bool keep = false;
        if (i < n) {
            if (P.kkt) keep = true;
            else {
                const T sxu = x[i] + u[i];
                keep = !(sxu > ub[i] || sxu < lb[i]);
            }
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (i0 > 0) __syncthreads();                  // (wtot of the previous pass has been read by everybody)
        if (lane == 0) wtot[w] = __popcll(bal);
        __syncthreads();
        int base = nf;
#pragma unroll
        for (int ww = 0; ww < LQP_NW; ++ww) {
            const int c = wtot[ww];
            if (ww < w) base += c;
            nf += c;
        }
        if (keep) {
            fl[base + before] = i;
            if (blockIdx.y == 0) P.fidx[(size_t)b * n + base + before] = i;
        }
    }
    if (tid == 0 && blockIdx.y == 0) P.nred[b] = nf + m;
    __syncthreads();
    // ---- reduced matrix and right-hand side (rows interleaved over the gridDim.y workgroups) ----
    const int wy = blockIdx.y, ny = gridDim.y;
    // (two rows per wave at a time, eight gathers of each in flight before the first store -- clamped indices, no guarded load:
    //  one gather, its wait and its store per trip were ~5 memory round trips per row, 56 us at n = 500)
    constexpr int GQ = 8;
    for (int a0 = (w * ny + wy) * 2; a0 < nf; a0 += LQP_NW * ny * 2) {
        const int a1 = a0 + 1 < nf ? a0 + 1 : a0;
        const int i0 = fl[a0], i1 = fl[a1];
        const T* q0 = Q + (size_t)i0 * n;
        const T* q1 = Q + (size_t)i1 * n;
        const T dg0 = P.kkt ? kkt_weight(P, b, i0) : T(1e-8), dg1 = P.kkt ? kkt_weight(P, b, i1) : T(1e-8);
        for (int c0 = 0; c0 < nf; c0 += 64 * GQ) {
            T v0[GQ], v1[GQ];
#pragma unroll
            for (int q = 0; q < GQ; ++q) {
                const int c = c0 + lane + 64 * q;
                const int col = fl[c < nf ? c : nf - 1];
                v0[q] = q0[col];
                v1[q] = q1[col];
            }
#pragma unroll
            for (int q = 0; q < GQ; ++q) {
                const int c = c0 + lane + 64 * q;
                if (c < nf) {
                    M[(size_t)a0 * Np + c] = c == a0 ? v0[q] + dg0 : v0[q];
                    if (a1 != a0) M[(size_t)a1 * Np + c] = c == a1 ? v1[q] + dg1 : v1[q];
                }
            }
        }
        for (int r = lane; r < m; r += 64) {
            M[(size_t)a0 * Np + nf + r] = A[(size_t)r * n + i0];
            if (a1 != a0) M[(size_t)a1 * Np + nf + r] = A[(size_t)r * n + i1];
        }
        if (lane == 0 && P.g) { rhs[a0] = -g[i0]; if (a1 != a0) rhs[a1] = -g[i1]; }      // (no cotangent yet: phase 1, k_bwd_gather_rhs later)
    }
    for (int r = w * ny + wy; r < m; r += LQP_NW * ny) {
        T* mr = M + (size_t)(nf + r) * Np;
        for (int c = lane; c < nf; c += 64) mr[c] = A[(size_t)r * n + fl[c]];
        for (int c = lane; c < m; c += 64) mr[nf + c] = (c == r && !P.kkt) ? T(1e-8) : T(0);
        if (lane == 0) rhs[nf + r] = T(0);
    }
}

// ---------------------------------------------------------------------------
// Cholesky form of the reduced backward system (f32, Q symmetric, n <= 1024, m <= 16):
//   [[Kf, A_F^T], [A_F, eps I]] [dv_F; dnu] = [-g_F; 0],   Kf = Q_FF + eps I  (eps = 1e-8, :378-392)
//   dv_F = u0 - G dnu,  u0 = Kf^-1 (-g_F),  G = Kf^-1 A_F^T,  (A_F G - eps I) dnu = A_F u0.
// Build: ordered compaction of the free set, Kf as packed lower 64x64 blocks (identity padding), -g_F -> rhs,
// A_F rows -> the (otherwise unused) M buffer.  LDS: fl[n] (int) | wtot[NW]
// ---------------------------------------------------------------------------
// Symmetric equilibration of the backward's free-set block by POWERS OF TWO: s_a = 2^-floor(log2(d_a) / 2), d_a the diagonal entry
// (q_aa + 1e-8, or + the KKT form's weight), so that every diagonal entry of S (Q_FF + eps I) S lies in [1, 4).  Exact in floating
// point -- the float32 Cholesky factorisation sees the same significands, its roundoff is invariant -- and it is what lets the
// float16-pipe tile products scale a 32-row block by ONE factor: the reference's Q is not pre-conditioned in the backward
// (:378-393), and rows 10^3 apart in magnitude inside a block lost up to ten bits there (measured on D Q D, d = 10^U(-1.5, 1.5):
// dp 5e-5 of scale against 5e-7).  The system solved is S K S (S^-1 dv) = S rhs, the equality rows enter as A_F S.
__device__ __forceinline__ float bwd_equil_scale(const float d) {
    if (!(d > 0.f) || !(d < 3.0e38f)) return 1.f;
    const int h = (int)((__float_as_uint(d) >> 23) & 0xFFu) - 127;       // floor(log2 d)
    return __uint_as_float((unsigned int)(127 - (h >> 1)) << 23);
}
template <int LQP_ANY = 0>      // (a template only so that the split build can place its one instance: tools/gen_split_build.py)
__global__ __launch_bounds__(LQP_NT) void k_bwd_build_chol(const BwdParams<float> P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int* fl = (int*)smem;
    int* wtot = fl + round_up(n, 8);
    const float* x = P.x + (size_t)b * n;
    const float* u = P.u + (size_t)b * n;
    const float* lb = P.lb + (size_t)b * n;
    const float* ub = P.ub + (size_t)b * n;
    const float* g = P.g + (size_t)b * n;
    const float* Q = P.Q + (size_t)b * n * n;
    const float* A = P.A ? P.A + (size_t)b * m * n : nullptr;
    const int Kmax = round_up(n, LQP_NB) / LQP_NB, Npm = Kmax * LQP_NB;
    float* Ls = P.packed + (size_t)b * sym_blocks(Kmax) * LQP_BLK;
    float* AF = P.M + (size_t)b * Np * Np;
    float* rhs = P.rhs + (size_t)b * Np;
    if (tid == 0 && blockIdx.y == 0) P.info[b] = 0;
    float* wl = (float*)(wtot + LQP_NW + 8);          // KKT-system backward: the diagonal weights (n floats)
    bool keep = false;
    if (tid < n) {
        if (P.kkt) { keep = true; wl[tid] = kkt_weight(P, b, tid); }
        else {
            const float sxu = x[tid] + u[tid];
            keep = !(sxu > ub[tid] || sxu < lb[tid]);
        }
    }
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[w] = __popcll(bal);
    __syncthreads();
    int base = 0, nf = 0;
#pragma unroll
    for (int ww = 0; ww < LQP_NW; ++ww) {
        const int c = wtot[ww];
        if (ww < w) base += c;
        nf += c;
    }
    if (keep) {
        fl[base + before] = tid;
        if (blockIdx.y == 0) P.fidx[(size_t)b * n + base + before] = tid;
    }
    if (tid == 0 && blockIdx.y == 0) P.nred[b] = nf + m;
    __syncthreads();
    const int Kb = round_up(nf, LQP_NB) / LQP_NB;
    const int r = tid >> 4, c4 = (tid & 15) * 4;
    const int nblk = sym_blocks(Kb);
    // the equilibration (bwd_equil_scale): s of free variable a, 1 on the padding; kept for the solve kernel
    float* const sl = wl + round_up(n, 8);
    for (int a = tid; a < Npm; a += LQP_NT) {
        float sv = 1.f;
        if (P.bsc && a < nf) sv = bwd_equil_scale(Q[(size_t)fl[a] * n + fl[a]] + (P.kkt ? wl[fl[a]] : 1e-8f));
        sl[a] = sv;
        if (P.bsc && blockIdx.y == 0) P.bsc[(size_t)b * Np + a] = sv;
    }
    __syncthreads();
    for (int t = blockIdx.y; t < nblk; t += gridDim.y) {
        int j = 0;
        while (sym_idx(j + 1, j + 1, Kb) <= t && j + 1 < Kb) ++j;      // block column of stream position t
        const int i = j + (t - sym_idx(j, j, Kb));
        const int a = i * 64 + r;
        // (unconditional gathers with clamped indices, then a select: a guarded load costs a branch and a full
        //  s_waitcnt per element)
        const bool rok = a < nf;
        const size_t rowoff = (size_t)fl[rok ? a : 0] * n;
        float q4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = j * 64 + c4 + e;
            q4[e] = Q[rowoff + fl[c < nf ? c : 0]];
        }
        V4<float> v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = j * 64 + c4 + e;
            float val = (a == c) ? 1.f : 0.f;
            if (rok && c < nf) val = ((q4[e] + (a == c ? (P.kkt ? wl[fl[a]] : 1e-8f) : 0.f)) * sl[a]) * sl[c];
            v.v[e] = val;
        }
        *(V4<float>*)(Ls + (size_t)t * LQP_BLK + tid * 4) = v;
    }
    if (blockIdx.y == 0) {
        if (P.phase != 1)          // (phase 1: no cotangent yet -- k_bwd_chol_solve gathers it in phase 2)
            for (int a = tid; a < Npm; a += LQP_NT) rhs[a] = a < nf ? -g[fl[a]] * sl[a] : 0.f;
        for (int q = 0; q < m; ++q)
            for (int a = tid; a < Npm; a += LQP_NT) AF[(size_t)q * Npm + a] = a < nf ? A[(size_t)q * n + fl[a]] * sl[a] : 0.f;
    }
}

// factor + solves + Schur complement of the equality rows; leaves [dv_F; dnu] in rhs like the LU path
// LDS: wg_chol_factor's layout, then v | acc | u0 | G[m][Npm] | t[64] | part[NW*64] | S[m*m] | wv[m] | dn[m]
// (nr: right-hand sides solved together, 2 or 4 -- see k_bwd_chol_solve)
__host__ __device__ inline int bwd_chol_lds_bytes(int n, int m, int nr = 2) {
    const int Kmax = round_up(n, LQP_NB) / LQP_NB, Npm = Kmax * LQP_NB;
    const int a0 = spd_lds_bytes(Kmax > SPD_MAXK ? SPD_MAXK : Kmax);      // (above: wg_chol_factor_big, panel in chunks)
    const int a = (Kmax < SPD_MAXK && chol_la_lds_bytes(Kmax) > a0) ? chol_la_lds_bytes(Kmax) : a0;
    const int c = ((1 + nr + m) * Npm + nr * 64 + nr * LQP_NW * 64 + m * m + 2 * m + 8) * 4;
    return a > c ? a : c;
}
// NRV = 0: the 1 + m solves two right-hand sides at a time (the headline's m = 1: one round); NRV = 4: four at a time, an instance
// of its own for problems with three or more equality rows (m = 16: five rounds of block-column barriers instead of nine) -- its
// registers are not the m <= 2 kernel's problem.  The same arithmetic per right-hand side either way.
// F16 (round 6): the look-ahead factorisation's tile waves on the float16 matrix pipe (wg_chol_factor_la<true>)
template <int NRV = 0, bool F16 = false>
__global__ __launch_bounds__(LQP_NT) void k_bwd_chol_solve(const BwdParams<float> P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Kmax = round_up(n, LQP_NB) / LQP_NB, Npm = Kmax * LQP_NB;
    const int nf = P.nred[b] - m;
    const int Kb = round_up(nf, LQP_NB) / LQP_NB, Nb = Kb * LQP_NB;
    float* Ls = P.packed + (size_t)b * sym_blocks(Kmax) * LQP_BLK;
    const float* AF = P.M + (size_t)b * Np * Np;
    float* rhs = P.rhs + (size_t)b * Np;
    unsigned long long dt0 = P.dbg ? clock64() : 0;
    if (Kb > 0 && P.phase != 2) {
        if (Kb <= P.la_maxk) wg_chol_factor_la<F16>(Ls, Kb, P.info + b, smem, P.dbg ? P.dbg + (size_t)b * 8 : nullptr);
        else if (Kb <= SPD_MAXK) wg_chol_factor(Ls, Kb, P.info + b, smem);
        else wg_chol_factor_big(Ls, Kb, P.info + b, smem);
    }
    // The factorisation is what can fail (Q_FF not positive definite in float32: the caller repeats on the pivoted LU): its info
    // word goes to the caller's pinned host memory NOW -- of the prefactor call when there was one (phase 1; phase 2 then finds
    // P.reported) --, so that a synchronous caller, which polls those words, returns while the solves and the epilogue still run
    // (stream-ordered results) and prepares its next call under them.  What the Schur complement of the equality rows can still
    // find below (a pivot that is not a positive number: with its -1e-8 I that is a NaN, i.e. NaN inputs) is not reported
    // any more: the epilogue poisons that problem's gradients with NaN, which is what the reference's solve hands back for it.
    if (P.host_report && (P.early_report || P.phase == 1) && !P.reported) {      // (a prefactor call always reports)
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(P.host_report + b, __hip_atomic_load(P.info + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (P.phase == 1) return;              // (the factor is in the workspace: lqp_boxqp_backward_fp_prefactor)
    __syncthreads();
    if (P.dbg && tid == 0) { const unsigned long long t = clock64(); P.dbg[(size_t)b * 8 + 0] = t - dt0; dt0 = t; P.dbg[(size_t)b * 8 + 3] = Kb; }
    // right-hand sides [rhs | A_F^T] as rows of X (u0 = X[0], G = X[1 ..]), solved two at a time
    constexpr int NR = NRV == 4 ? 4 : 2;
    float* acc = (float*)smem;
    float* u0 = acc + (size_t)NR * Npm;
    float* G = u0 + Npm;
    float* t = G + (size_t)m * Npm;
    float* part = t + NR * 64;
    float* S = part + NR * LQP_NW * 64;
    float* wv = S + m * m;
    float* dn = wv + m;
    if (P.phase == 2) {                    // the cotangent arrives only now: gathered over the free set of phase 1
        const float* g = P.g + (size_t)b * n;
        const int* fidx = P.fidx + (size_t)b * n;
        const float* bs_ = P.bsc ? P.bsc + (size_t)b * Np : nullptr;
        for (int e = tid; e < Nb; e += LQP_NT) u0[e] = e < nf ? -g[fidx[e]] * (bs_ ? bs_[e] : 1.f) : 0.f;
    } else {
        for (int e = tid; e < Nb; e += LQP_NT) u0[e] = rhs[e];
    }
    for (int q = 0; q < m; ++q)
        for (int e = tid; e < Nb; e += LQP_NT) G[(size_t)q * Npm + e] = AF[(size_t)q * Npm + e];
    __syncthreads();
    if (Kb > 0)
        for (int c0 = 0; c0 < 1 + m; c0 += NR)
            wg_chol_solve_n<NR>(Ls, Kb, u0 + (size_t)c0 * Npm, Npm, (1 + m - c0) < NR ? (1 + m - c0) : NR, acc, t, part,
                                (P.dbg && P.la_maxk == 0) ? P.dbg + (size_t)b * 8 : nullptr);
    if (P.dbg && tid == 0) { const unsigned long long t = clock64(); P.dbg[(size_t)b * 8 + 1] = t - dt0; dt0 = t; }
    if (m > 0) {
        // S = A_F G - eps I,  wv = A_F u0  (one wave per entry)
        for (int e = w; e < m * m + m; e += LQP_NW) {
            const int q = e < m * m ? e / m : e - m * m;
            const float* rv = e < m * m ? G + (size_t)(e - q * m) * Npm : u0;
            float a = 0.f;
            for (int i = lane; i < nf; i += 64) a += AF[(size_t)q * Npm + i] * rv[i];
            a = wave_sum(a);
            if (lane == 0) {
                if (e < m * m) S[e] = a - ((e / m == e % m) ? 1e-8f : 0.f);
                else wv[q] = a;
            }
        }
        __syncthreads();
        // (S - eps I) dnu = wv: Gauss-Jordan with partial pivoting, m <= 16.  One thread per entry of [S | wv] (a single thread walked
        // through all of it before: ~4 m^3 dependent LDS round trips, 0.2 ms of the 0.3 ms kernel at m = 16); the same operations on
        // the same values per entry, so the same bits.  dn[0] doubles as the pivot row, dn[1] as the skip flag of a column step.
        {
            const int rr = tid / (m + 1), jj = tid - rr * (m + 1);          // entry (rr, jj); jj == m: the right-hand side
            const bool mine = tid < m * (m + 1);
            int* pivrow = (int*)dn;
            for (int c = 0; c < m; ++c) {
                if (tid == 0) {
                    int pr = c;
                    float best = tabs(S[c * m + c]);
                    for (int r2 = c + 1; r2 < m; ++r2) if (tabs(S[r2 * m + c]) > best) { best = tabs(S[r2 * m + c]); pr = r2; }
                    const bool skip = !(best > 0.f);
                    if (skip && P.info[b] == 0) P.info[b] = nf + c + 1;
                    pivrow[0] = pr; pivrow[1] = skip ? 1 : 0;
                }
                __syncthreads();
                const int pr = pivrow[0];
                const bool skip = pivrow[1] != 0;
                float* const rowc = jj < m ? S + c * m + jj : wv + c;        // this thread's column of rows c / pr / rr
                float* const rowp = jj < m ? S + pr * m + jj : wv + pr;
                float* const rowr = jj < m ? S + rr * m + jj : wv + rr;
                if (!skip && pr != c && tid <= m) {                          // swap rows c and pr (threads 0 .. m: rr == 0, one column each)
                    const float tmp = *rowc; *rowc = *rowp; *rowp = tmp;
                }
                __syncthreads();
                const float inv = skip ? 0.f : 1.f / S[c * m + c];
                const float f = (mine && !skip) ? S[rr * m + c] : 0.f;       // (read before anybody writes column c)
                __syncthreads();
                if (!skip && tid <= m) *rowc *= inv;
                __syncthreads();
                if (mine && !skip && rr != c) *rowr -= f * *rowc;
                __syncthreads();
            }
            for (int q = tid; q < m; q += LQP_NT) dn[q] = wv[q];
        }
        __syncthreads();
    }
    for (int a = tid; a < nf; a += LQP_NT) {
        float d = u0[a];
        for (int q = 0; q < m; ++q) d -= G[(size_t)q * Npm + a] * dn[q];
        rhs[a] = P.bsc ? d * P.bsc[(size_t)b * Np + a] : d;             // dv = S (the solution of the equilibrated system)
    }
    for (int q = tid; q < m; q += LQP_NT) rhs[nf + q] = dn[q];
    if (P.dbg && tid == 0) P.dbg[(size_t)b * 8 + 2] = clock64() - dt0;
}

// One step of iterative refinement for the LU form of the reduced system: r = rhs - M d with the ORIGINAL entries
// (Q, A gathered again; M itself was factored in place), dot products accumulated in double.  The cached solve uses
// explicitly inverted 64x64 diagonal blocks, which -- like any fp32 LU of this bordered, nearly singular-cornered
// matrix -- leaves 1e-5 ... 1e-4 of error when the exact dv is ~0 (dl_dz in the row space of A); one correction
// solve with the same factor brings it to the fp32 rounding level.  LDS: dvf[n] | dn[m] | fl[n] (int)
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_bwd_residual(const BwdParams<T> P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T* dvf = (T*)smem;                     // dv scattered to all n variables (0 on the active set)
    T* dn = dvf + round_up(n, 8);
    int* fl = (int*)(dn + round_up(m > 0 ? m : 1, 8));
    const int nf = P.nred[b] - m;
    const T* d = P.rhs + (size_t)b * Np;
    T* r = P.rhs2 + (size_t)b * Np;
    const T* Q = P.Q + (size_t)b * n * n;
    const T* A = P.A ? P.A + (size_t)b * m * n : nullptr;
    const T* g = P.g + (size_t)b * n;
    for (int i = tid; i < n; i += LQP_NT) dvf[i] = T(0);
    for (int a = tid; a < nf; a += LQP_NT) fl[a] = P.fidx[(size_t)b * n + a];
    for (int q = tid; q < m; q += LQP_NT) dn[q] = d[nf + q];
    __syncthreads();
    for (int a = tid; a < nf; a += LQP_NT) dvf[fl[a]] = d[a];
    __syncthreads();
    // gridDim.y == 2: two workgroups per problem, alternate groups of rows (at B <= 128 half the CUs are idle).  Four rows per
    // wave at a time, their loads issued together (a row at a time was one memory round trip + reduction per row: 31 in sequence
    // at n = 500); each row's terms are added in the order they always were.
    constexpr int RG = 4;
    const int ny = (int)gridDim.y, me = (int)blockIdx.y;
    for (int a0 = (w * ny + me) * RG; a0 < nf + m; a0 += LQP_NW * ny * RG) {
        const T* row[RG];
        double acc[RG];
#pragma unroll
        for (int u = 0; u < RG; ++u) {
            const int a = min(a0 + u, nf + m - 1);
            row[u] = a >= nf ? A + (size_t)(a - nf) * n : Q + (size_t)fl[a] * n;
            acc[u] = 0.0;
        }
        for (int j = lane; j < n; j += 64) {
            const double dj = (double)dvf[j];
            T v[RG];
#pragma unroll
            for (int u = 0; u < RG; ++u) v[u] = row[u][j];
#pragma unroll
            for (int u = 0; u < RG; ++u) acc[u] += (double)v[u] * dj;
        }
#pragma unroll
        for (int u = 0; u < RG; ++u) acc[u] = wave_sum(acc[u]);
        if (lane < RG && a0 + lane < nf + m) {
            const int a = a0 + lane;
            double mine = readlane_t(acc[0], 0);          // (lane 0's sums: the bits the one-row form produced)
#pragma unroll
            for (int u = 1; u < RG; ++u) { const double t = readlane_t(acc[u], 0); mine = lane == u ? t : mine; }
            double res;
            if (a < nf) {
                const int i = fl[a];
                res = -(double)g[i] - mine - 1e-8 * (double)d[a];
                for (int q = 0; q < m; ++q) res -= (double)A[(size_t)q * n + i] * (double)dn[q];
            } else {
                res = -mine - 1e-8 * (double)dn[a - nf];
            }
            r[a] = (T)res;
        }
    }
    if (me == 0)
        for (int a = nf + m + tid; a < Np; a += LQP_NT) r[a] = T(0);
}

// solve with the packed factor (one rhs per problem, in global memory, in place)
// LDS: v[Np] | tmp[64] | dest[Np]
template <typename T> __host__ __device__ inline int solve_lds_bytes(int Np) { return (Np + 64) * (int)sizeof(T) + Np * 4; }

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_packed_solve(const T* __restrict__ packed_all, const int Nuni, const int Npmax,
                                                         const int Kmax, const int* __restrict__ dest_all,
                                                         T* __restrict__ rhs_all, const int nrhs,
                                                         const size_t rhs_bstride, const int rhs_rstride,
                                                         const int rhs_cstride, const int* __restrict__ Nvec) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = Nvec ? Nvec[b] : Nuni;
    const int K = round_up(N, LQP_NB) / LQP_NB, Np = K * LQP_NB;
    T* v = (T*)smem;
    T* tmp = v + Npmax;
    int* dest = (int*)(tmp + 64);
    const T* packed = packed_all + (size_t)b * packed_blocks(Kmax) * LQP_BLK;
    const int S = K * (K + 1);
    const bool cyclic = true;                 // (wg_packed_solve pads the stream to a multiple of the ring depth)
    BlockStream<T, LQP_NT> st;
    stream_prime(st, packed, S);
    for (int i = tid; i < Np; i += LQP_NT) dest[i] = dest_all[(size_t)b * Npmax + i];
    __syncthreads();
    T* rhs = rhs_all + (size_t)b * rhs_bstride;
    for (int c = 0; c < nrhs; ++c) {
        for (int i = tid; i < Np; i += LQP_NT) v[dest[i]] = (i < N) ? rhs[(size_t)i * rhs_rstride + (size_t)c * rhs_cstride] : T(0);
        wg_barrier_lds();
        wg_packed_solve(st, packed, K, v, tmp, cyclic && (c + 1 < nrhs));
        if (!cyclic && c + 1 < nrhs) stream_prime(st, packed, S);
        for (int i = tid; i < N; i += LQP_NT) rhs[(size_t)i * rhs_rstride + (size_t)c * rhs_cstride] = v[i];
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_bwd_epilogue(const BwdParams<T> P) {
    // gradients :396-430; d = [dv; dnu] is in P.rhs
    extern __shared__ __attribute__((aligned(32))) char smem[];
    const int b = blockIdx.x, n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T* dv = (T*)smem;          // n
    T* xs = dv + n;            // n
    T* dnu = xs + n;           // m
    const T rho = (P.rho_mode == 2) ? P.rho_in[b] : P.rho_value;
    const T* d = P.rhs + (size_t)b * Np;
    const T* x = P.x + (size_t)b * n;
    // singular system / Q_FF not positive definite: gradients come out as NaN, never as plausible garbage
    const T poison = P.info[b] != 0 ? T(__builtin_nanf("")) : T(0);
    if (P.host_report && !(P.chol && P.early_report) && !P.lu_reported && tid == 0 && blockIdx.y == 0)      // (straight into pinned host memory: no device-to-host copy behind the
                                                                       //  call; Cholesky form: k_bwd_chol_solve has reported already)
        __hip_atomic_store(P.host_report + b, P.info[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (P.reduced) {
        const int nf = P.nred[b] - m;
        const int* fl = P.fidx + (size_t)b * n;
        const T* d2 = P.rhs2 + (size_t)b * Np;           // correction of the refinement step (LU form)
        for (int i = tid; i < n; i += LQP_NT) { dv[i] = poison; xs[i] = x[i]; }
        __syncthreads();
        for (int a = tid; a < nf; a += LQP_NT) dv[fl[a]] = d[a] + (P.refine ? d2[a] : T(0)) + poison;
        for (int r = tid; r < m; r += LQP_NT) dnu[r] = d[nf + r] + (P.refine ? d2[nf + r] : T(0)) + poison;
    } else {
        for (int i = tid; i < n; i += LQP_NT) { dv[i] = d[i] + poison; xs[i] = x[i]; }
        for (int r = tid; r < m; r += LQP_NT) dnu[r] = d[n + r] + poison;
    }
    __syncthreads();
    // gridDim.y workgroups share the rows of one problem (dQ is a pure 4 n^2-byte write); the small outputs: slab 0
    const bool first = blockIdx.y == 0;
    if (P.dp && first) for (int i = tid; i < n; i += LQP_NT) P.dp[(size_t)b * n + i] = dv[i];
    if (P.db && first) for (int r = tid; r < m; r += LQP_NT) P.db[(size_t)b * m + r] = -dnu[r];
    if (P.dA && m > 0 && first) {
        const T* nus = P.nus + (size_t)b * m;
        for (int t = tid; t < m * n; t += LQP_NT) {
            const int r = t / n, j = t - r * n;
            P.dA[(size_t)b * m * n + t] = dnu[r] * xs[j] + nus[r] * dv[j];
        }
    }
    const T* Q = P.Q + (size_t)b * n * n;
    const T* A = P.A ? P.A + (size_t)b * m * n : nullptr;
    if (P.kkt && first) {
        // dlam = diag(1 / s) G dx (rows -I | I), dl_dh = -lam dlam, dlb = -dl_dh[:n], dub = dl_dh[n:] (:544, :573-575)
        for (int i = tid; i < n; i += LQP_NT) {
            const T xi = xs[i];
            const T slo = tmax(xi - P.lb[(size_t)b * n + i], T(1e-8)), shi = tmax(P.ub[(size_t)b * n + i] - xi, T(1e-8));
            const T llo = tmax(P.lams[(size_t)b * 2 * n + i], T(1e-8)), lhi = tmax(P.lams[(size_t)b * 2 * n + n + i], T(1e-8));
            if (P.dlb) P.dlb[(size_t)b * n + i] = llo * (-dv[i] / slo);
            if (P.dub) P.dub[(size_t)b * n + i] = -lhi * (dv[i] / shi);
        }
    }
    const bool need_kkt = (P.dlb || P.dub) && !P.kkt;
    T* dQ = P.dQ ? P.dQ + (size_t)b * n * n : nullptr;
    // 16-B stores; a lane's columns are the same in every row, so their x_j and dv_j / 2 stay in registers
    const bool vec = (n % 4 == 0) && n <= 1024 && dQ && ((((uintptr_t)dQ) % sizeof(V4<T>)) == 0);
    V4<T> xj[4], hj[4];
    if (vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane * 4 + 256 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) { xj[q].v[e] = j < n ? xs[j + e] : T(0); hj[q].v[e] = j < n ? T(0.5) * dv[j + e] : T(0); }
        }
    }
    for (int i = w + LQP_NW * blockIdx.y; i < n; i += LQP_NW * gridDim.y) {
        const T hi = T(0.5) * dv[i], xi = xs[i];
        T acc = T(0);
        if (need_kkt) {
            const T* qr = Q + (size_t)i * n;
            for (int j = lane; j < n; j += 64) acc += qr[j] * dv[j];
            acc = wave_sum(acc);
        }
        if (dQ) {
            T* o = dQ + (size_t)i * n;
            if (vec) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = lane * 4 + 256 * q;
                    if (j < n) {
                        V4<T> ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov.v[e] = hi * xj[q].v[e] + hj[q].v[e] * xi;
                        *(V4<T>*)(o + j) = ov;
                    }
                }
            } else {
                for (int j = lane; j < n; j += 64) o[j] = hi * xs[j] + (T(0.5) * dv[j]) * xi;
            }
        }
        if (need_kkt && lane == 0) {
            T kkt = -P.g[(size_t)b * n + i] - acc;
            T at = T(0);
            for (int r = 0; r < m; ++r) at += A[(size_t)r * n + i] * dnu[r];
            if (m > 0) kkt = kkt - at;
            T div = rho * P.u[(size_t)b * n + i];
            if (div == T(0)) div = T(1);
            const T dlam = kkt / div;
            const T* lams = P.lams + (size_t)b * 2 * n;
            if (P.dlb) P.dlb[(size_t)b * n + i] = dlam * lams[i];
            if (P.dub) P.dub[(size_t)b * n + i] = -dlam * lams[n + i];
        }
    }
}

// ---------------------------------------------------------------------------
// KKT solve helpers (lqp_py/solve_qp_eqcon_torch.py:23-25, utils.py:23-32)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_kkt_build(const T* __restrict__ Qall, const T* __restrict__ pall,
                                                      const T* __restrict__ Aall, const T* __restrict__ ball,
                                                      const int n, const int m, const int Np,
                                                      T* __restrict__ Mall, T* __restrict__ rhsall, int* info) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const T* Q = Qall + (size_t)b * n * n;
    T* M = Mall + (size_t)b * Np * Np;
    T* rhs = rhsall + (size_t)b * Np;
    if (tid == 0) info[b] = 0;
    for (int i = w; i < n; i += LQP_NW) {
        const T* qr = Q + (size_t)i * n;
        T* mr = M + (size_t)i * Np;
        for (int j = lane; j < n; j += 64) mr[j] = qr[j];
        for (int r = lane; r < m; r += 64) mr[n + r] = Aall[(size_t)b * m * n + (size_t)r * n + i];
        if (lane == 0) rhs[i] = -pall[(size_t)b * n + i];
    }
    for (int r = w; r < m; r += LQP_NW) {
        T* mr = M + (size_t)(n + r) * Np;
        for (int j = lane; j < n; j += 64) mr[j] = Aall[(size_t)b * m * n + (size_t)r * n + j];
        for (int c = lane; c < m; c += 64) mr[n + c] = T(0);
        if (lane == 0) rhs[n + r] = ball[(size_t)b * m + r];
    }
}

template <typename T>
__global__ void k_kkt_unpack(const T* __restrict__ rhsall, const int n, const int m, const int Np,
                             T* __restrict__ x, T* __restrict__ nus) {
    const int b = blockIdx.x;
    const T* d = rhsall + (size_t)b * Np;
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[(size_t)b * n + i] = d[i];
    if (nus) for (int r = threadIdx.x; r < m; r += blockDim.x) nus[(size_t)b * m + r] = d[n + r];
}

// dQ = 0.5 (dx x^T + x dx^T); dA = dnu x^T + nus dx^T
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_outer_grads(const T* __restrict__ dx, const T* __restrict__ x,
                                                        const T* __restrict__ dnu, const T* __restrict__ nus,
                                                        const int n, const int m, T* __restrict__ dQ, T* __restrict__ dA) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const T* dxb = dx + (size_t)b * n;
    const T* xb = x + (size_t)b * n;
    if (dQ) {
        for (int i = w; i < n; i += LQP_NW) {
            const T di = dxb[i], xi = xb[i];
            T* o = dQ + (size_t)b * n * n + (size_t)i * n;
            for (int j = lane; j < n; j += 64) o[j] = T(0.5) * (di * xb[j] + xi * dxb[j]);
        }
    }
    if (dA && m > 0) {
        for (int t = tid; t < m * n; t += LQP_NT) {
            const int r = t / n, j = t - r * n;
            dA[(size_t)b * m * n + t] = dnu[(size_t)b * m + r] * xb[j] + nus[(size_t)b * m + r] * dxb[j];
        }
    }
}

// strided copies between a user (B,N,N) matrix and the padded workspace
template <typename T>
__global__ void k_copy_matrix(const T* __restrict__ src, const int lds_, const size_t sstride,
                              T* __restrict__ dst, const int ldd, const size_t dstride, const int N) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = w; i < N; i += nw)
        for (int j = lane; j < N; j += 64) dst[(size_t)b * dstride + (size_t)i * ldd + j] = src[(size_t)b * sstride + (size_t)i * lds_ + j];
}
// phase 2 of the LU form: the right-hand side [-g_F; 0] over the free set phase 1 left in the workspace (what k_bwd_build_reduced
// writes itself when it has the cotangent)
template <typename T>
__global__ __launch_bounds__(256) void k_bwd_gather_rhs(const BwdParams<T> P) {
    const int b = blockIdx.x, n = P.n, m = P.m;
    const int nf = P.nred[b] - m;
    const T* g = P.g + (size_t)b * n;
    const int* fl = P.fidx + (size_t)b * n;
    T* rhs = P.rhs + (size_t)b * P.Np;
    for (int a = threadIdx.x; a < nf; a += 256) rhs[a] = -g[fl[a]];
    for (int r = threadIdx.x; r < m; r += 256) rhs[nf + r] = T(0);
}

// the info words of a factorisation into the caller's pinned host memory (LU form of the backward: they are final behind the LU, a
// synchronous caller that polls them returns while pack, solves and epilogue still run)
template <int LQP_ANY = 0>
__global__ void k_report_info(const int* __restrict__ info, int* __restrict__ host_report, const int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) __hip_atomic_store(host_report + i, info[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename TI>
__global__ void k_copy_ints(const TI* __restrict__ src, const int sstride, TI* __restrict__ dst, const int dstride, const int N) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += blockDim.x) dst[(size_t)b * dstride + i] = src[(size_t)b * sstride + i];
}

}  // namespace lqp
