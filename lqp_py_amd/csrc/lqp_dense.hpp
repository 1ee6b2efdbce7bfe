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
template <typename T> struct InvCfg {
    static constexpr int TR = sizeof(T) == 4 ? 32 : 16;       // rows / columns of one matrix-instruction tile
    static constexpr int TC = TR;
    static constexpr int KS = sizeof(T) == 4 ? 2 : 4;         // k-depth of one instruction (= lane groups)
    static constexpr int RT = 64 / TR;                        // row tiles per 64-row block: 2 | 4
    static constexpr int CT = 4 / RT;                         // column tiles per workgroup: 2 | 1
    static constexpr int TWG = CT * TC;                       // columns of X per workgroup tile: 64 | 16
    static constexpr int YS = TWG + (sizeof(T) == 4 ? 4 : 2); // row stride of Y in LDS (padded against bank conflicts)
};
template <typename T> __host__ __device__ inline int lu_inverse_lds_bytes(int Np) { return Np * InvCfg<T>::YS * (int)sizeof(T); }

// acc (one TR x TC tile, C layout) -= A[TR x 64] * Y[64 x TC]:  A = rows r0 .. r0+TR of a 64 x 64 row-major block `blk`;
// Y rows yr0 .. yr0+64, columns yc0 .. of the LDS array (row stride YS)
template <typename T>
struct InvAcc;
template <> struct InvAcc<float> {
    f32x16 a;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = 0.f;
    }
    // element (row, col) of register q in lane (li, lg)
    __device__ static __forceinline__ int row(int q, int lg) { return (q & 3) + 8 * (q >> 2) + 4 * lg; }
    __device__ __forceinline__ void mac(const float* __restrict__ blk, const int r0, const float* __restrict__ Y, const int YS,
                                        const int li, const int lg, const float sign) {
        const float* ap = blk + (size_t)(r0 + li) * 64 + lg;
        const float* bp = Y + (size_t)lg * YS + li;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 2)
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(sign * ap[kk], bp[(size_t)kk * YS], a, 0, 0, 0);
    }
};
template <> struct InvAcc<double> {
    f64x4 a;
    __device__ __forceinline__ void zero() { a = f64x4{0.0, 0.0, 0.0, 0.0}; }
    __device__ static __forceinline__ int row(int q, int lg) { return 4 * q + lg; }
    __device__ __forceinline__ void mac(const double* __restrict__ blk, const int r0, const double* __restrict__ Y, const int YS,
                                        const int li, const int lg, const double sign) {
        const double* ap = blk + (size_t)(r0 + li) * 64 + lg;
        const double* bp = Y + (size_t)lg * YS + li;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 4)
            a = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * ap[kk], bp[(size_t)kk * YS], a, 0, 0, 0);
    }
};

// X[0:N, 0:N] (row-major, leading dimension ldx) = M^-1; packed / dest: what k_pack left (dest[r]: position of original
// right-hand-side row r after the row interchanges).  grid = (B, G): workgroup (b, g) takes column tiles g, g + G, ...
template <typename T>
__global__ __launch_bounds__(256) void k_lu_inverse(const T* __restrict__ packed_all, const size_t pkstride, const int Nuni,
                                                    const int Kmax, const int* __restrict__ dest_all, const int dstride,
                                                    T* __restrict__ X_all, const size_t xstride, const int ldx,
                                                    const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    typedef InvCfg<T> C;
    if (gate && *gate == 0) return;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int N = Nuni;
    const int K = round_up(N, LQP_NB) / LQP_NB, Np = K * LQP_NB;
    (void)Kmax;
    const int li = lane % C::TR, lg = lane / C::TR;
    const int rt = w % C::RT, ct = w / C::RT;                // this wave's row tile of a block row / column tile of the workgroup tile
    T* Y = (T*)smem;
    const T* packed = packed_all + (size_t)b * pkstride;
    const T* Lpk = packed;
    const T* Upk = packed + (size_t)(K * (K + 1) / 2) * LQP_BLK;
    const int* dest = dest_all + (size_t)b * dstride;
    T* X = X_all + (size_t)b * xstride;
    const int ntiles = (N + C::TWG - 1) / C::TWG;
    for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
        const int c0 = tile * C::TWG;
        // ---- right-hand sides: columns c0 .. of P I ----
        for (int i = tid; i < Np * C::YS; i += 256) Y[i] = T(0);
        __syncthreads();
        if (tid < C::TWG && c0 + tid < N) Y[(size_t)dest[c0 + tid] * C::YS + tid] = T(1);
        __syncthreads();
        InvAcc<T> acc;
        const int ycol = ct * C::TC;
        // ---- L phase: Y_k <- inv(L_kk) (Y_k - sum_{j<k} L_kj Y_j), k ascending ----
        for (int k = 0; k < K; ++k) {
            const T* rowblk = Lpk + (size_t)(k * (k + 1) / 2) * LQP_BLK;       // L(k,0) .. L(k,k-1), inv(L(k,k))
            T* Yk = Y + (size_t)(64 * k) * C::YS + ycol;
            acc.zero();
            for (int j = 0; j < k; ++j)
                acc.mac(rowblk + (size_t)j * LQP_BLK, C::TR * rt, Y + (size_t)(64 * j) * C::YS + ycol, C::YS, li, lg, T(-1));
            constexpr int NQ = sizeof(T) == 4 ? 16 : 4;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                T* p = Yk + (size_t)(C::TR * rt + InvAcc<T>::row(q, lg)) * C::YS + li;
                *p = *p + acc.a[q];
            }
            __syncthreads();
            acc.zero();
            acc.mac(rowblk + (size_t)k * LQP_BLK, C::TR * rt, Yk, C::YS, li, lg, T(1));
            __syncthreads();                                   // (every wave has read the old Y_k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) Yk[(size_t)(C::TR * rt + InvAcc<T>::row(q, lg)) * C::YS + li] = acc.a[q];
            __syncthreads();
        }
        // ---- U phase: Y_k <- inv(U_kk) (Y_k - sum_{j>k} U_kj Y_j), k descending ----
        for (int k = K - 1; k >= 0; --k) {
            const int kr = K - 1 - k;                           // block rows the U phase has visited
            const T* rowblk = Upk + (size_t)(kr * (kr + 1) / 2) * LQP_BLK;     // U(k,K-1) .. U(k,k+1), inv(U(k,k))
            T* Yk = Y + (size_t)(64 * k) * C::YS + ycol;
            acc.zero();
            for (int j = K - 1; j > k; --j)
                acc.mac(rowblk + (size_t)(K - 1 - j) * LQP_BLK, C::TR * rt, Y + (size_t)(64 * j) * C::YS + ycol, C::YS, li, lg, T(-1));
            constexpr int NQ = sizeof(T) == 4 ? 16 : 4;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                T* p = Yk + (size_t)(C::TR * rt + InvAcc<T>::row(q, lg)) * C::YS + li;
                *p = *p + acc.a[q];
            }
            __syncthreads();
            acc.zero();
            acc.mac(rowblk + (size_t)kr * LQP_BLK, C::TR * rt, Yk, C::YS, li, lg, T(1));
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NQ; ++q) Yk[(size_t)(C::TR * rt + InvAcc<T>::row(q, lg)) * C::YS + li] = acc.a[q];
            __syncthreads();
        }
        // ---- X[:, c0 ..] = Y ----
        for (int i = tid; i < N * C::TWG; i += 256) {
            const int r = i / C::TWG, c = i - r * C::TWG;
            if (c0 + c < N) X[(size_t)r * ldx + c0 + c] = Y[(size_t)r * C::YS + c];
        }
        __syncthreads();
    }
}

}  // namespace lqp
