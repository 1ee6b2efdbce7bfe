// The dense tier of the LU path: when the factor of the KKT matrix is small enough for M^-1 to live ON CHIP (registers of two
// workgroups), the x-update of the ADMM loop (lqp_py/solve_box_qp_admm_torch.py:258-268: lu_solve with the cached factor) becomes
// one matrix-vector product with the explicit inverse -- the form the symmetric tier has had since round 1, here for ANY matrix
// the pivoted LU takes: float64, many equality rows, non-symmetric Q.
//
//   k_lu_inverse      X = M^-1 from the packed factor (solve-ordered 64 x 64 panels with pre-inverted diagonal blocks, lqp_trsv.hpp):
//                     a blocked triangular solve with N right-hand sides, every step a 64-deep block product on the matrix cores
//                     (v_mfma_f64_16x16x4 / v_mfma_f32_32x32x2).  Column tiles are independent: any number of workgroups per
//                     matrix, no communication.
//   k_admm_loop_dense the loop: two workgroups per problem, each holds its half of the ROWS of X[0:n, 0:n] in registers for
//                     the whole launch, x-halves cross as tagged 8-byte granules (the hand-off form of k_admm_loop_split).
//
// Why it pays (rocprofv3, profiles/r05_b_hard64_*): the cached-LU loop streams the padded factor -- 960 KB per problem and
// iteration at n = 250, m = 16, float64 -- from beyond the L2 at the rate the fabric delivers (7.7 GB per launch, 1.05 ms of a
// 2.05 ms step); the inverse costs ~4/3 N^3 flops ONCE and the loop then moves nothing.
#pragma once
#include "lqp_boxqp.hpp"

namespace lqp {

// ---------------------------------------------------------------------------------------------------------------------
// X = M^-1 from the packed factor
// ---------------------------------------------------------------------------------------------------------------------
// One workgroup of 256 threads works on a tile of TWG columns of X at a time: 4 waves = (64 / TR row tiles) x (TWG / TC column
// tiles) of one 64-row block; Y (Np x TWG, the right-hand sides turning into the solution) lives in LDS.
// SMALL (float32 only): 16 x 16 tiles of v_mfma_f32_16x16x4 instead of 32 x 32 of v_mfma_f32_32x32x2 -- a workgroup tile of 16 columns,
// so that Y fits the LDS for N up to 2048 (64 columns: N <= 630); float64 is always 16 x 16.
template <typename T, bool SMALL = false> struct InvCfg {
    static constexpr int TR = (sizeof(T) == 4 && !SMALL) ? 32 : 16;       // rows / columns of one matrix-instruction tile
    static constexpr int TC = TR;
    static constexpr int KS = (sizeof(T) == 4 && !SMALL) ? 2 : 4;         // k-depth of one instruction (= lane groups)
    static constexpr int RT = 64 / TR;                        // row tiles per 64-row block: 2 | 4
    static constexpr int CT = 4 / RT;                         // column tiles per workgroup: 2 | 1
    static constexpr int TWG = CT * TC;                       // columns of X per workgroup tile: 64 | 16
    static constexpr int YS = TWG + 1;                        // row stride of Y in LDS: odd, so that the lane groups' rows (16 / 32 apart)
                                                              // start 128 B apart in the banks (an even stride put all of them on the same ones)
};
template <typename T, bool SMALL = false> __host__ __device__ inline int lu_inverse_lds_bytes(int Np) { return Np * InvCfg<T, SMALL>::YS * (int)sizeof(T); }

// acc (one TR x TC tile, C layout) -= A[TR x 64] * Y[64 x TC]:  A = rows r0 .. r0+TR of a 64 x 64 row-major block `blk`;
// Y rows yr0 .. yr0+64, columns yc0 .. of the LDS array (row stride YS)
template <typename T, bool SMALL = false>
struct InvAcc;
template <> struct InvAcc<float, false> {
    static constexpr int NA = 32;                                  // A operands of one 64-deep block product
    f32x16 a;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = 0.f;
    }
    // element (row, col) of register q in lane (li, lg)
    __device__ static __forceinline__ int row(int q, int lg) { return (q & 3) + 8 * (q >> 2) + 4 * lg; }
    __device__ static __forceinline__ void load(float (&op)[NA], const float* __restrict__ blk, const int r0, const int li, const int lg) {
        // (lane group lg takes k = 32 lg + t -- any split of the 64 terms over the instruction's k-slots sums the same product --,
        //  so a lane's operands are 128 contiguous bytes of its row: whole cache lines, four times fewer line fetches than k = 2 t + lg)
        const float* ap = blk + (size_t)(r0 + li) * 64 + 32 * lg;
#pragma unroll
        for (int v = 0; v < NA / 4; ++v) {
            const V4<float> x = *(const V4<float>*)(ap + 4 * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) op[4 * v + e] = x.v[e];
        }
    }
    __device__ __forceinline__ void mac(const float (&op)[NA], const float* __restrict__ Y, const int YS, const int li, const int lg) {
        const float* bp = Y + (size_t)(32 * lg) * YS + li;
        float bq[NA];                                              // (all B operands requested first: read one by one, each matrix
#pragma unroll                                                     //  instruction stood behind an LDS round trip of its own)
        for (int t = 0; t < NA; ++t) bq[t] = bp[(size_t)t * YS];
#pragma unroll
        for (int t = 0; t < NA; ++t) a = __builtin_amdgcn_mfma_f32_32x32x2f32(op[t], bq[t], a, 0, 0, 0);
    }
};
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// (operand / result layout of v_mfma_f32_16x16x4_f32 as measured, tools/microbench/mfma_f32_16x16x4_layout.hip: A[i = l & 15][k = l >> 4],
//  B[k = l >> 4][j = l & 15], D register q <-> row 4 (l >> 4) + q, column l & 15 -- the rows are NOT the float64 instruction's)
template <> struct InvAcc<float, true> {
    static constexpr int NA = 16;
    f32x4_t a;
    __device__ __forceinline__ void zero() { a = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    __device__ static __forceinline__ int row(int q, int lg) { return 4 * lg + q; }
    __device__ static __forceinline__ void load(float (&op)[NA], const float* __restrict__ blk, const int r0, const int li, const int lg) {
        const float* ap = blk + (size_t)(r0 + li) * 64 + 16 * lg;        // (k = 16 lg + t: 64 contiguous bytes of the lane's row)
#pragma unroll
        for (int v = 0; v < NA / 4; ++v) {
            const V4<float> x = *(const V4<float>*)(ap + 4 * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) op[4 * v + e] = x.v[e];
        }
    }
    __device__ __forceinline__ void mac(const float (&op)[NA], const float* __restrict__ Y, const int YS, const int li, const int lg) {
        const float* bp = Y + (size_t)(16 * lg) * YS + li;
        float bq[NA];
#pragma unroll
        for (int t = 0; t < NA; ++t) bq[t] = bp[(size_t)t * YS];
#pragma unroll
        for (int t = 0; t < NA; ++t) a = __builtin_amdgcn_mfma_f32_16x16x4f32(op[t], bq[t], a, 0, 0, 0);
    }
};
template <bool SMALL> struct InvAcc<double, SMALL> {
    static constexpr int NA = 16;
    f64x4 a;
    __device__ __forceinline__ void zero() { a = f64x4{0.0, 0.0, 0.0, 0.0}; }
    __device__ static __forceinline__ int row(int q, int lg) { return 4 * q + lg; }
    __device__ static __forceinline__ void load(double (&op)[NA], const double* __restrict__ blk, const int r0, const int li, const int lg) {
        const double* ap = blk + (size_t)(r0 + li) * 64 + 16 * lg;       // (k = 16 lg + t: see the float32 form)
#pragma unroll
        for (int v = 0; v < NA / 4; ++v) {
            const V4<double> x = *(const V4<double>*)(ap + 4 * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) op[4 * v + e] = x.v[e];
        }
    }
    __device__ __forceinline__ void mac(const double (&op)[NA], const double* __restrict__ Y, const int YS, const int li, const int lg) {
        const double* bp = Y + (size_t)(16 * lg) * YS + li;
        double bq[NA];
#pragma unroll
        for (int t = 0; t < NA; ++t) bq[t] = bp[(size_t)t * YS];
#pragma unroll
        for (int t = 0; t < NA; ++t) a = __builtin_amdgcn_mfma_f64_16x16x4f64(op[t], bq[t], a, 0, 0, 0);
    }
};

// X[0:N, 0:N] (row-major, leading dimension ldx) = M^-1; packed / dest: what k_pack left (dest[r]: position of original
// right-hand-side row r after the row interchanges).  grid = (B, G): workgroup (b, g) takes column tiles g, g + G, ...
// Gx > 0: a ONE-dimensional grid of 8 ceil(B / 8) * G workgroups, one tile each, dealt out so that the tiles of a
// problem run on ONE XCD at about the same time: workgroup id -> XCD id % 8 (the dispatcher's round robin), slot id / 8 on it,
// problem 8 * (slot / Gx) + XCD, tile slot % Gx.  Every tile reads the whole factor (960 KB at N = 266, float64); with (b, g)
// in grid order the tiles of one problem are 128 workgroups apart, six tiles of ALL problems are resident together (16 factors
// per 4-MB L2) and every tile's pass comes from beyond the L2: 1.25 GB per launch at B = 128.  Problem-major on an XCD, five or
// six factors are in flight per L2 and a factor is fetched about once.
template <typename T, bool SMALL = false>
__global__ __launch_bounds__(256) void k_lu_inverse(const T* __restrict__ packed_all, const size_t pkstride, const int Nuni,
                                                    const int Gx, const int* __restrict__ dest_all, const int dstride,
                                                    T* __restrict__ X_all, const size_t xstride, const int ldx,
                                                    const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    typedef InvCfg<T, SMALL> C;
    typedef InvAcc<T, SMALL> Acc;
    if (gate && *gate == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int b = blockIdx.x, tile0 = blockIdx.y, tstep = gridDim.y;
    if (Gx > 0) {                                            // (Gx = G | B << 16; the grid is padded to a multiple of 8 problems)
        const int G = Gx & 0xFFFF, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        b = 8 * (slot / G) + xcd;
        if (b >= (Gx >> 16)) return;
        tile0 = slot % G;
        tstep = G;
    }
    const int N = Nuni;
    const int K = round_up(N, LQP_NB) / LQP_NB, Np = K * LQP_NB;
    const int li = lane % C::TR, lg = lane / C::TR;
    const int rt = w % C::RT, ct = w / C::RT;                // this wave's row tile of a block row / column tile of the workgroup tile
    T* Y = (T*)smem;
    const T* packed = packed_all + (size_t)b * pkstride;
    const T* Lpk = packed;
    const T* Upk = packed + (size_t)(K * (K + 1) / 2) * LQP_BLK;
    const int* dest = dest_all + (size_t)b * dstride;
    T* X = X_all + (size_t)b * xstride;
    const int ntiles = (N + C::TWG - 1) / C::TWG;
    for (int tile = tile0; tile < ntiles; tile += tstep) {
        const int c0 = tile * C::TWG;
        // ---- right-hand sides: columns c0 .. of P I ----
        for (int i = tid; i < Np * C::YS; i += 256) Y[i] = T(0);
        __syncthreads();
        if (tid < C::TWG && c0 + tid < N) Y[(size_t)dest[c0 + tid] * C::YS + tid] = T(1);
        __syncthreads();
        Acc acc;
        const int ycol = ct * C::TC;
        // One block row of a phase: acc = sum_j B_j Y_(col j) over the off-diagonal blocks (A operands of the NEXT block are
        // requested before the current block's matrix instructions: the loads' latency stands behind 1024 cycles of them),
        // Y_k -= acc, barrier, Y_k <- Dinv Y_k with the pre-inverted diagonal block (loaded ahead of the barrier).
        auto block_row = [&](const T* rowblk, const int noff, const int k, auto col_of) {
            T* Yk = Y + (size_t)(64 * k) * C::YS + ycol;
            T opA[Acc::NA], opB[Acc::NA];
            constexpr int NQ = (sizeof(T) == 4 && !SMALL) ? 16 : 4;
            acc.zero();
            Acc::load(opA, rowblk, C::TR * rt, li, lg);                 // block 0 (the diagonal one when noff == 0)
            for (int j = 0; j < noff; ++j) {
                Acc::load(opB, rowblk + (size_t)(j + 1) * LQP_BLK, C::TR * rt, li, lg);
                acc.mac(opA, Y + (size_t)(64 * col_of(j)) * C::YS + ycol, C::YS, li, lg);
#pragma unroll
                for (int t = 0; t < Acc::NA; ++t) opA[t] = opB[t];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                T* p = Yk + (size_t)(C::TR * rt + Acc::row(q, lg)) * C::YS + li;
                *p = *p - acc.a[q];
            }
            __syncthreads();
            acc.zero();
            acc.mac(opA, Yk, C::YS, li, lg);                             // (opA holds the diagonal block's operands by now)
            __syncthreads();                                                   // (every wave has read the old Y_k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) Yk[(size_t)(C::TR * rt + Acc::row(q, lg)) * C::YS + li] = acc.a[q];
            __syncthreads();
        };
        // ---- L phase: Y_k <- inv(L_kk) (Y_k - sum_{j<k} L_kj Y_j), k ascending: L(k,0) .. L(k,k-1), inv(L(k,k)) ----
        for (int k = 0; k < K; ++k)
            block_row(Lpk + (size_t)(k * (k + 1) / 2) * LQP_BLK, k, k, [](int j) { return j; });
        // ---- U phase: Y_k <- inv(U_kk) (Y_k - sum_{j>k} U_kj Y_j), k descending: U(k,K-1) .. U(k,k+1), inv(U(k,k)) ----
        for (int k = K - 1; k >= 0; --k) {
            const int kr = K - 1 - k;                           // block rows the U phase has visited
            block_row(Upk + (size_t)(kr * (kr + 1) / 2) * LQP_BLK, kr, k, [K](int j) { return K - 1 - j; });
        }
        // ---- X[:, c0 ..] = Y ----
        for (int i = tid; i < N * C::TWG; i += 256) {
            const int r = i / C::TWG, c = i - r * C::TWG;
            if (c0 + c < N) X[(size_t)r * ldx + c0 + c] = Y[(size_t)r * C::YS + c];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The ADMM loop with the explicit inverse: [x; nu] = X [w; b], w = -p + rho (z - u)  (reference :258-268), n <= 256
// ---------------------------------------------------------------------------------------------------------------------
// Two workgroups of 512 threads per problem; workgroup `part` owns rows [h0, h1) of X[0:n, 0:n] -- four threads per row, 64
// columns each, IN REGISTERS for the whole launch (n = 250, float64: 126 VGPRs per thread).  Per iteration: the product with w
// (LDS, broadcast reads; one quad reduction per row), the constant term c = X[0:n, n:N] b, the own half of x published as
// tagged granules and the partner's fetched (two buffers by iteration parity), then BOTH workgroups run the element-wise update
// of all n variables redundantly (identical bits: no second exchange, and both know the verdict of a check).  nu = X[n:N, :] [w; b]
// is formed from global memory where it is needed (checks, last iteration).
constexpr int DENSE_NT = 512, DENSE_CPT = 64, DENSE_TPR = 4, DENSE_NMAX = DENSE_CPT * DENSE_TPR;
template <typename T> __host__ __device__ constexpr int dense_ws() { return DENSE_CPT + (sizeof(T) == 4 ? 4 : 2); }   // chunk stride of w in LDS
// granules per problem: [parity][row], one (float32) or two (float64) 8-byte words per element
template <typename T> __host__ __device__ constexpr size_t dense_xchg_words() { return (size_t)2 * DENSE_NMAX * (sizeof(T) / 4); }
// LDS: wl[4 WS] xs z u ps lb ub D [7 x 256] cvl[128] | bs nul [2 m] | red[8 * 8 + 8] | flags[8]
template <typename T> __host__ __device__ inline int dense_loop_lds_bytes(int m) {
    return (DENSE_TPR * dense_ws<T>() + 7 * DENSE_NMAX + DENSE_NMAX / 2 + 2 * m + (DENSE_NT / 64) * 8 + 8 + 8) * (int)sizeof(T) + 64;
}

template <typename T>
__global__ __launch_bounds__(DENSE_NT) void k_admm_loop_dense(const FwdParams<T> P, const int it0, const int it1, const int ctr_base) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NT = DENSE_NT, NWV = NT / 64, CPT = DENSE_CPT, WS = dense_ws<T>(), GW = sizeof(T) / 4, NM = DENSE_NMAX;
    int b, part;
    if (!shared_map((int)blockIdx.x, P.B, 2, b, part)) return;
    const int n = P.n, m = P.m, Np = P.Np;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (__hip_atomic_load(P.status + ST_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (it0 >= it1) return;
    T* wl = (T*)smem;
    T* xs = wl + DENSE_TPR * WS;
    T* z = xs + NM;
    T* u = z + NM;
    T* ps = u + NM;
    T* lb = ps + NM;
    T* ub = lb + NM;
    T* D = ub + NM;
    T* cvl = D + NM;
    T* bs = cvl + NM / 2;
    T* nul = bs + m;
    T* red = nul + m;
    int* flags = (int*)(red + NWV * 8 + 8);                   // [0] exchange timed out (sticky)
    const int nh = (n + 1) / 2;
    const int h0 = part ? nh : 0, h1 = part ? n : nh;
    const int r = tid >> 2, q = tid & 3, row = h0 + r;
    const bool rowok = row < h1;
    const T* X = P.M + (size_t)b * Np * Np;                   // the inverse, written over the factor's LAPACK copy by k_lu_inverse
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T* scal = P.scal + (size_t)b * SC_WORDS;
    const T rho = scal[SC_RHO];
    const T pnorm = scal[SC_PNORM];
    static_assert(dense_xchg_words<T>() <= (size_t)DNX_WORDS, "granule area");
    unsigned long long* const xq = P.dnx + (size_t)b * P.dnx_words;

    // ---- my rows of X[0:n, 0:n] into registers ----
    T H[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = q * CPT + j;
        H[j] = (rowok && c < n) ? X[(size_t)row * Np + c] : T(0);
    }
    for (int i = tid; i < NM; i += NT) {
        const bool in = i < n;
        z[i] = in ? V.z[i] : T(0); u[i] = in ? V.u[i] : T(0); ps[i] = in ? V.ps[i] : T(0);
        lb[i] = in ? V.lbs[i] : T(0); ub[i] = in ? V.ubs[i] : T(0); D[i] = in ? V.D[i] : T(1);
        xs[i] = T(0);
    }
    for (int k = tid; k < m; k += NT) { bs[k] = V.bs[k]; nul[k] = T(0); }
    if (tid < 8) flags[tid] = 0;
    __syncthreads();
    if (q == 0 && r < NM / 2) {                                  // c = X[0:n, n:N] b, my rows
        T acc = T(0);
        if (rowok)
            for (int k = 0; k < m; ++k) acc += X[(size_t)row * Np + n + k] * bs[k];
        cvl[r] = acc;
    }
    for (int i = tid; i < NM; i += NT) wl[(i >> 6) * WS + (i & 63)] = (i < n) ? -ps[i] + rho * (z[i] - u[i]) : T(0);
    __syncthreads();

    int slot = ctr_base;
    for (int it = it0; it < it1; ++it) {
        const bool check = (it % P.check_solved) == 0;
        // ---- x (my rows) = X w + c ----
        const T* wq = wl + q * WS;
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);          // (four chains: one chain of 64 dependent FMAs is latency, not work)
#pragma unroll
        for (int j = 0; j < CPT; j += 4) {
            a0 += H[j] * wq[j]; a1 += H[j + 1] * wq[j + 1]; a2 += H[j + 2] * wq[j + 2]; a3 += H[j + 3] * wq[j + 3];
        }
        T acc = (a0 + a1) + (a2 + a3);
        acc += dpp<0xB1>(acc);
        acc += dpp<0x4E>(acc);
        const unsigned int tag = (unsigned int)(it + 1);
        unsigned long long* const xb = xq + (size_t)(it & 1) * NM * GW;
        if (q == 0 && rowok) {
            const T xi = acc + cvl[r];
            xs[row] = xi;
            if constexpr (GW == 1) {
                __hip_atomic_store(xb + row, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned int, xi),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const unsigned long long bits = __builtin_bit_cast(unsigned long long, xi);
                __hip_atomic_store(xb + 2 * row, ((unsigned long long)tag << 32) | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(xb + 2 * row + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- the partner's rows ----
        {
            const int p0 = part ? 0 : nh, p1 = part ? nh : n;
            const int prow = p0 + tid;
            if (prow < p1) {
                unsigned long long g[GW];
                bool bad = false;
#pragma unroll
                for (int e = 0; e < GW; ++e) {
                    unsigned int spins = 0;
                    unsigned long long t0 = 0;
                    for (;;) {
                        g[e] = __hip_atomic_load(xb + GW * prow + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned int)(g[e] >> 32) == tag || flags[0]) break;
                        if ((++spins & 255u) == 0) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();         // 100 MHz
                            if (t0 == 0) t0 = now;
                            else if (now - t0 > 50000000ULL) {                                       // 0.5 s: give up, flagged
                                __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                flags[0] = 1;
                                bad = true;
                                break;
                            }
                        }
                    }
                }
                if (!bad) {
                    if constexpr (GW == 1) xs[prow] = __builtin_bit_cast(float, (unsigned int)g[0]);
                    else xs[prow] = __builtin_bit_cast(double, (g[GW - 1] << 32) | (g[0] & 0xFFFFFFFFull));
                }
            }
        }
        __syncthreads();
        // ---- nu = X[n:N, :] [w; b] where a check or the end of the launch needs it (w is still this iteration's) ----
        if ((check || it + 1 == it1) && m > 0) {
            for (int k = w; k < m; k += NWV) {
                const T* xr = X + (size_t)(n + k) * Np;
                T a = T(0);
                for (int i = lane; i < n; i += 64) a += xr[i] * wl[(i >> 6) * WS + (i & 63)];
                for (int j = lane; j < m; j += 64) a += xr[n + j] * bs[j];
                a = wave_sum(a);
                if (lane == 0) nul[k] = a;
            }
            __syncthreads();
        }
        // ---- z-update, residuals, dual (:271-282), all n variables on both workgroups ----
        T mx[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) mx[e] = T(0);
        if (tid < NM) {
            const int i = tid;
            const bool in = i < n;
            const T xi = xs[i];
            const T zp = z[i];
            const T ui = u[i];
            T zn = xi + ui;
            zn = tmin(tmax(zn, lb[i]), ub[i]);
            const T rr = xi - zn;
            const T ss = rho * (zn - zp);
            const T un = ui + rr;
            if (in) { z[i] = zn; u[i] = un; }
            if (check && in) {
                const T di = D[i];
                mx[0] = tabs(di * rr);
                mx[1] = tabs(di * ss);
                mx[2] = tabs(di * xi);
                mx[3] = tabs(di * zn);
                mx[4] = tabs((rho * di) * un);
                T qx = -ps[i] + rho * (zp - ui) - rho * xi;
                for (int k = 0; k < m; ++k) qx -= V.As[(size_t)k * n + i] * nul[k];
                mx[5] = tabs(qx / di);
            }
            wl[(i >> 6) * WS + (i & 63)] = in ? -ps[i] + rho * (zn - un) : T(0);
        }
        if (check) {
            T mv[6] = {mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]};
            wg_max_n<T, 6, NWV>(mv, red);
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
                unsigned int r1 = 0, r2 = 0;
                if (part == 0) {
                    scal[SC_RATIO] = ratio;
                    scal[SC_WANTS] = wants ? T(1) : T(0);
                    scal[SC_PRI] = mv[0];
                    scal[SC_DUA] = mv[1];
                    trace_check(P.vtrace, it, P.check_solved, P.ring, mv[0], mv[1]);
                    if (wants) r1 = atomicAdd(ct + CT_WANTS, 1u);
                    if (trig) r2 = atomicAdd(ct + CT_TRIG, 1u);
                }
                asm volatile("s_waitcnt vmcnt(0)" :: "v"(r1), "v"(r2) : "memory");
                // both workgroups arrive (they computed the same numbers); the verdict is counted once, by part 0
                __hip_atomic_fetch_add((unsigned long long*)(ct + CT_NOTOPT), ((part == 0 && !solved) ? 1ull : 0ull) | (1ull << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ++slot;
            grid_wait(ct + CT_ARRIVE, 2u * (unsigned int)P.B, P.status);      // device-wide "all optimal?" (torch.all at :312)
            const unsigned int notopt = __hip_atomic_load(ct + CT_NOTOPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int tmo = __hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (notopt == 0 || tmo) {
                if (blockIdx.x == 0 && tid == 0) {
                    P.status[ST_FINAL_ITER] = it;
                    __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                break;
            }
        }
        __syncthreads();
    }
    // ---- state for the continuation launch / the epilogue ----
    if (part == 0) {
        for (int i = tid; i < n; i += NT) { V.z[i] = z[i]; V.u[i] = u[i]; V.x[i] = xs[i]; }
        for (int k = tid; k < m; k += NT) V.nu[k] = nul[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same loop on W workgroups per problem, for batches that leave most of the chip idle and any n the inverse kernel takes
// (float32 to 2048, float64 to ~1100): the cached-LU loop is ONE workgroup per problem streaming the whole factor every
// iteration (B = 8, n = 1500: 9.8 MB at the ~80 GB/s one CU keeps in flight = 123 us per iteration, 9.9 of the 21 ms of a step),
// and triangular solves do not spread over CUs -- a matrix-vector product does.  Workgroup `part` of W holds rows
// [part RPW, (part + 1) RPW) of X[0:n, 0:n] in registers, TPR threads per row with CPT columns each (RPW = 512 / TPR); per
// iteration it forms its RPW entries of x, publishes them as tagged granules and takes everybody else's (an all-gather through
// global memory: two buffers by iteration parity), then runs the element-wise update of ALL n variables like every other
// workgroup of the problem (identical bits; everybody knows the verdict of a check).
// LDS: wl[TPR chunks of CPT + pad] xs z u ps lb ub D [7 x NV] cvl[128] | bs nul [2 m] | red | flags   (NV = n rounded up to 64)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DENSEW_NT = 512;
template <typename T> __host__ __device__ constexpr int densew_cpt() { return sizeof(T) == 4 ? 192 : 64; }      // columns of a row per thread: 192 / 128 VGPRs (72 doubles: 12 spilled registers, 88: 44, 96: 60)
template <typename T> __host__ __device__ constexpr int densew_ws() { return densew_cpt<T>() + (sizeof(T) == 4 ? 4 : 2); }
__host__ __device__ inline int densew_tpr(int n, int cpt) {
    int t = 4;
    while (t * cpt < n) t <<= 1;
    return t;                                       // threads per row: 4, 8, 16 or 32 (adjacent lanes)
}
template <typename T> __host__ __device__ inline int densew_lds_bytes(int n, int m) {
    const int NV = round_up(n, 64), tpr = densew_tpr(n, densew_cpt<T>());
    return (tpr * densew_ws<T>() + 7 * NV + 128 + 2 * (m > 0 ? m : 1) + (DENSEW_NT / 64) * 8 + 8 + 8) * (int)sizeof(T) + 64;
}
// granules per problem: [parity][element], one (float32) or two (float64) 8-byte words per element
template <typename T> __host__ __device__ inline size_t densew_xchg_words(int n) { return (size_t)2 * round_up(n, 64) * (sizeof(T) / 4); }

template <typename T>
__global__ __launch_bounds__(DENSEW_NT) void k_admm_loop_dense_w(const FwdParams<T> P, const int it0, const int it1, const int ctr_base,
                                                                 const int TPR) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NT = DENSEW_NT, NWV = NT / 64, CPT = densew_cpt<T>(), WS = densew_ws<T>(), GW = sizeof(T) / 4;
    const int b = (int)blockIdx.x % P.B, part = (int)blockIdx.x / P.B, W = (int)gridDim.x / P.B;
    const int n = P.n, m = P.m, Np = P.Np, NV = round_up(n, 64);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (__hip_atomic_load(P.status + ST_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (it0 >= it1) return;
    T* wl = (T*)smem;
    T* xs = wl + TPR * WS;
    T* z = xs + NV;
    T* u = z + NV;
    T* ps = u + NV;
    T* lb = ps + NV;
    T* ub = lb + NV;
    T* D = ub + NV;
    T* cvl = D + NV;
    T* bs = cvl + 128;
    T* nul = bs + (m > 0 ? m : 1);
    T* red = nul + (m > 0 ? m : 1);
    int* flags = (int*)(red + NWV * 8 + 8);                   // [0] exchange timed out (sticky)
    const int RPW = NT / TPR;
    const int h0 = part * RPW, h1 = (h0 + RPW < n) ? h0 + RPW : n;
    const int r = tid / TPR, q = tid % TPR, row = h0 + r;
    const bool rowok = row < h1;
    const T* X = P.M + (size_t)b * Np * Np;                   // the inverse, written over the factor's LAPACK copy by k_lu_inverse
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    T* scal = P.scal + (size_t)b * SC_WORDS;
    const T rho = scal[SC_RHO];
    const T pnorm = scal[SC_PNORM];
    unsigned long long* const xq = P.dnx + (size_t)b * P.dnx_words;

    // ---- my rows of X[0:n, 0:n] into registers ----
    T H[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = q * CPT + j;
        H[j] = (rowok && c < n) ? X[(size_t)row * Np + c] : T(0);
    }
    for (int i = tid; i < NV; i += NT) {
        const bool in = i < n;
        z[i] = in ? V.z[i] : T(0); u[i] = in ? V.u[i] : T(0); ps[i] = in ? V.ps[i] : T(0);
        lb[i] = in ? V.lbs[i] : T(0); ub[i] = in ? V.ubs[i] : T(0); D[i] = in ? V.D[i] : T(1);
        xs[i] = T(0);
    }
    for (int k = tid; k < m; k += NT) { bs[k] = V.bs[k]; nul[k] = T(0); }
    for (int i = tid; i < TPR * WS; i += NT) wl[i] = T(0);
    if (tid < 8) flags[tid] = 0;
    __syncthreads();
    if (q == 0 && r < 128) {                                     // c = X[0:n, n:N] b, my rows
        T acc = T(0);
        if (rowok)
            for (int k = 0; k < m; ++k) acc += X[(size_t)row * Np + n + k] * bs[k];
        cvl[r] = acc;
    }
    for (int i = tid; i < n; i += NT) wl[(i / CPT) * WS + (i % CPT)] = -ps[i] + rho * (z[i] - u[i]);
    __syncthreads();

    int slot = ctr_base;
    for (int it = it0; it < it1; ++it) {
        const bool check = (it % P.check_solved) == 0;
        // ---- x (my rows) = X w + c ----
        const T* wq = wl + q * WS;
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
#pragma unroll
        for (int j = 0; j < CPT; j += 4) {
            a0 += H[j] * wq[j]; a1 += H[j + 1] * wq[j + 1]; a2 += H[j + 2] * wq[j + 2]; a3 += H[j + 3] * wq[j + 3];
        }
        T acc = (a0 + a1) + (a2 + a3);
        acc += dpp<0xB1>(acc);                                   // the TPR threads of a row are adjacent lanes
        acc += dpp<0x4E>(acc);
        if (TPR >= 8) acc += (T)__shfl_xor(acc, 4);              // (butterfly: every lane of the row's group ends with the sum)
        if (TPR >= 16) acc += (T)__shfl_xor(acc, 8);
        if (TPR >= 32) acc += xor16(acc);
        const unsigned int tag = (unsigned int)(it + 1);
        unsigned long long* const xb = xq + (size_t)(it & 1) * NV * GW;
        if (q == 0 && rowok) {
            const T xi = acc + cvl[r];
            xs[row] = xi;
            if constexpr (GW == 1) {
                __hip_atomic_store(xb + row, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned int, xi),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const unsigned long long bits = __builtin_bit_cast(unsigned long long, xi);
                __hip_atomic_store(xb + 2 * row, ((unsigned long long)tag << 32) | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(xb + 2 * row + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- everybody else's rows ----
        for (int i = tid; i < n; i += NT) {
            if (i >= h0 && i < h1) continue;
            unsigned long long g[GW];
            bool bad = false;
#pragma unroll
            for (int e = 0; e < GW; ++e) {
                unsigned int spins = 0;
                unsigned long long t0 = 0;
                for (;;) {
                    g[e] = __hip_atomic_load(xb + GW * i + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned int)(g[e] >> 32) == tag || flags[0]) break;
                    if ((++spins & 255u) == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();         // 100 MHz
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > 50000000ULL) {                                       // 0.5 s: give up, flagged
                            __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            flags[0] = 1;
                            bad = true;
                            break;
                        }
                    }
                }
            }
            if (!bad) {
                if constexpr (GW == 1) xs[i] = __builtin_bit_cast(float, (unsigned int)g[0]);
                else xs[i] = __builtin_bit_cast(double, (g[GW - 1] << 32) | (g[0] & 0xFFFFFFFFull));
            }
        }
        __syncthreads();
        // ---- nu = X[n:N, :] [w; b] where a check or the end of the launch needs it (w is still this iteration's) ----
        if ((check || it + 1 == it1) && m > 0) {
            for (int k = w; k < m; k += NWV) {
                const T* xr = X + (size_t)(n + k) * Np;
                T a = T(0);
                for (int i = lane; i < n; i += 64) a += xr[i] * wl[(i / CPT) * WS + (i % CPT)];
                for (int j = lane; j < m; j += 64) a += xr[n + j] * bs[j];
                a = wave_sum(a);
                if (lane == 0) nul[k] = a;
            }
            __syncthreads();
        }
        // ---- z-update, residuals, dual (:271-282), all n variables on every workgroup ----
        T mx[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) mx[e] = T(0);
        for (int i = tid; i < n; i += NT) {
            const T xi = xs[i];
            const T zp = z[i];
            const T ui = u[i];
            T zn = xi + ui;
            zn = tmin(tmax(zn, lb[i]), ub[i]);
            const T rr = xi - zn;
            const T ss = rho * (zn - zp);
            const T un = ui + rr;
            z[i] = zn; u[i] = un;
            if (check) {
                const T di = D[i];
                mx[0] = tmax(mx[0], tabs(di * rr));
                mx[1] = tmax(mx[1], tabs(di * ss));
                mx[2] = tmax(mx[2], tabs(di * xi));
                mx[3] = tmax(mx[3], tabs(di * zn));
                mx[4] = tmax(mx[4], tabs((rho * di) * un));
                T qx = -ps[i] + rho * (zp - ui) - rho * xi;
                for (int k = 0; k < m; ++k) qx -= V.As[(size_t)k * n + i] * nul[k];
                mx[5] = tmax(mx[5], tabs(qx / di));
            }
            wl[(i / CPT) * WS + (i % CPT)] = -ps[i] + rho * (zn - un);
        }
        if (check) {
            T mv[6] = {mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]};
            wg_max_n<T, 6, NWV>(mv, red);
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
                unsigned int r1 = 0, r2 = 0;
                if (part == 0) {
                    scal[SC_RATIO] = ratio;
                    scal[SC_WANTS] = wants ? T(1) : T(0);
                    scal[SC_PRI] = mv[0];
                    scal[SC_DUA] = mv[1];
                    trace_check(P.vtrace, it, P.check_solved, P.ring, mv[0], mv[1]);
                    if (wants) r1 = atomicAdd(ct + CT_WANTS, 1u);
                    if (trig) r2 = atomicAdd(ct + CT_TRIG, 1u);
                }
                asm volatile("s_waitcnt vmcnt(0)" :: "v"(r1), "v"(r2) : "memory");
                // every workgroup arrives (they computed the same numbers); the verdict is counted once, by part 0
                __hip_atomic_fetch_add((unsigned long long*)(ct + CT_NOTOPT), ((part == 0 && !solved) ? 1ull : 0ull) | (1ull << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ++slot;
            grid_wait(ct + CT_ARRIVE, (unsigned int)gridDim.x, P.status);      // device-wide "all optimal?" (torch.all at :312)
            const unsigned int notopt = __hip_atomic_load(ct + CT_NOTOPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int tmo = __hip_atomic_load(P.status + ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (notopt == 0 || tmo) {
                if (blockIdx.x == 0 && tid == 0) {
                    P.status[ST_FINAL_ITER] = it;
                    __hip_atomic_store(P.status + ST_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                break;
            }
        }
        __syncthreads();
    }
    // ---- state for the continuation launch / the epilogue ----
    if (part == 0) {
        for (int i = tid; i < n; i += NT) { V.z[i] = z[i]; V.u[i] = u[i]; V.x[i] = xs[i]; }
        for (int k = tid; k < m; k += NT) V.nu[k] = nul[k];
    }
}

}  // namespace lqp
