// Common device helpers for the gfx950 box-QP kernels.
// One workgroup (LQP_NT = 1024 threads = 16 wave64) owns one QP at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LQP_NT 1024            // threads per workgroup
#define LQP_NW (LQP_NT / 64)   // waves per workgroup
#define LQP_NB 64              // triangular-solve block edge
#define LQP_BLK (LQP_NB * LQP_NB)

namespace lqp {

__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Kernels that share a problem between NP workgroups want those workgroups on ONE XCD (their hand-offs then stay in its L2).  The
// dispatcher deals workgroup ids out round robin over the 8 XCDs, so with (problem, part) = (id % B, id / B) the partners b and
// b + B meet on one XCD only when B is a multiple of 8 (B = 100: 0.747 ms per step against 0.714 between B = 96 and 104; B = 30:
// 0.637 against 0.574 at B = 32).  For every B: XCD id % 8, slot id / 8 on it, problem 8 (slot / NP) + XCD, part slot % NP,
// on a grid of NP * 8 * ceil(B / 8) workgroups whose surplus ids (problem >= B) leave at once.
__host__ __device__ constexpr int shared_grid(int B, int NP) { return NP * 8 * ((B + 7) / 8); }
__device__ __forceinline__ bool shared_map(const int id, const int B, const int NP, int& b, int& part) {
    const int slot = id >> 3;
    b = 8 * (slot / NP) + (id & 7);
    part = slot % NP;
    return b < B;
}

// 4-element vector of T: one 16-B (f32) or 32-B (f64) global access per lane.
template <typename T> struct V4;
template <> struct __attribute__((aligned(16))) V4<float>  { float  v[4]; };
template <> struct __attribute__((aligned(32))) V4<double> { double v[4]; };

template <typename T> __device__ __forceinline__ T tabs(T a) { return a < T(0) ? -a : a; }
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }
__device__ __forceinline__ float  tsqrt(float a)  { return sqrtf(a); }
__device__ __forceinline__ double tsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ float  tfloor(float a)  { return floorf(a); }
__device__ __forceinline__ double tfloor(double a) { return floor(a); }
__device__ __forceinline__ float  tceil(float a)  { return ceilf(a); }
__device__ __forceinline__ double tceil(double a) { return ceil(a); }

// ---- DPP moves (no LDS crossbar): quad_perm / row_ror inside a 16-lane row ----
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = dpp_i32<CTRL>((int)(b & 0xffffffffLL));
    int hi = dpp_i32<CTRL>((int)(b >> 32));
    long long r = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, r);
}
// sum over the 16 lanes of a DPP row; every lane of the row gets the total
template <typename T> __device__ __forceinline__ T row16_sum(T v) {
    v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp<0x124>(v);   // row_ror:4
    v += dpp<0x128>(v);   // row_ror:8
    return v;
}
// ---- lanes l and l ^ 16 / l ^ 32 WITHOUT the LDS crossbar (gfx950: v_permlane16_swap / v_permlane32_swap) ----
// lane_swap16: a := [a rows 0, b rows 0, a rows 2, b rows 2], b := [a rows 1, b rows 1, a rows 3, b rows 3] (rows of 16 lanes):
//              afterwards a + b holds a's lane ^ 16 sum in the even rows and b's in the odd ones;
// lane_swap32: a := [a lanes 0-31, b lanes 0-31], b := [a lanes 32-63, b lanes 32-63]: a + b = a's lane ^ 32 sum below lane 32,
//              b's above.
// With a == b every lane gets the fold of the one value (fold16 / fold32: what v + __shfl_xor(v, 16) computes -- the same two
// summands, hence the same bits -- as two VALU instructions instead of a ds_bpermute round trip); with two different values one
// add folds both and the number of live values halves (col_fold).  Inline assembly: the builtin of hipcc 7.2 mixed up its two
// results (lqp_spd.hpp, piv_pair); the s_nops cover the swap's wait states behind a VALU write of its operands and before its
// results are read.
#ifndef LQP_LANE_SWAP
#define LQP_LANE_SWAP 1
#endif
__device__ __forceinline__ void lane_swap16(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap32(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <typename T> __device__ __forceinline__ T xor16(T v) { return (T)__shfl_xor(v, 16); }
template <typename T> __device__ __forceinline__ T xor32(T v) { return (T)__shfl_xor(v, 32); }
#if LQP_LANE_SWAP
// the value of lane l ^ 16 / l ^ 32
template <> __device__ __forceinline__ float xor16<float>(float v) {
    float a = v, b = v;
    lane_swap16(a, b);                  // a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
    return (threadIdx.x & 16) ? a : b;
}
template <> __device__ __forceinline__ float xor32<float>(float v) {
    float a = v, b = v;
    lane_swap32(a, b);                  // a = [lo lo], b = [hi hi]
    return (threadIdx.x & 32) ? a : b;
}
#endif
// Column sums over the rows a wave holds: lane l carries EPT partial sums of the columns EPT (l % LPR) .. of row l / LPR
// (LPR = 64 / EPT lanes per row); the fold runs over lane bits log2(LPR) .. 5 in ascending order (the sums
// ((a + a^8) + a^16) + a^32 of the plain shuffle fold: the same bits).  Afterwards lane l holds in a[k], k < EPT / 4, the
// total of element 4 k + col_fold_elem(l) of its column group.
template <int EPT> __device__ __forceinline__ void col_fold(float (&a)[EPT]) {
    static_assert(EPT == 4 || EPT == 8, "1024 or 512 threads per 64x64 block");
    if constexpr (EPT == 8) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) a[e] += dpp<0x128>(a[e]);          // row_ror:8 = lane ^ 8 inside a 16-lane DPP row
    }
#pragma unroll
    for (int e = 0; e < EPT; e += 2) { lane_swap16(a[e], a[e + 1]); a[e >> 1] = a[e] + a[e + 1]; }
#pragma unroll
    for (int e = 0; e < EPT / 2; e += 2) { lane_swap32(a[e], a[e + 1]); a[e >> 1] = a[e] + a[e + 1]; }
}
__device__ __forceinline__ int col_fold_elem(const int lane) { return 2 * (lane >> 5) + ((lane >> 4) & 1); }

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
    v = row16_sum(v);
    v += xor16(v);
    v += xor32(v);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
    v = tmax(v, dpp<0xB1>(v));
    v = tmax(v, dpp<0x4E>(v));
    v = tmax(v, dpp<0x124>(v));
    v = tmax(v, dpp<0x128>(v));
    v = tmax(v, xor16(v));
    v = tmax(v, xor32(v));
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. it
// drains every global load in flight -- fatal for a kernel whose prefetch ring must stay full across
// the barrier (the streaming triangular solve has two barriers per 64-row block).  Here: wait for
// this wave's LDS operations, then a raw s_barrier; global loads keep flying.
__device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Workgroup reductions through a small LDS scratch (>= LQP_NW elements of T).
// Both end with every thread holding the result; both contain barriers, so
// every thread of the workgroup must call them.
template <typename T> __device__ __forceinline__ T wg_sum(T v, T* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();                       // scratch may still be read by a previous call
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = T(0);
#pragma unroll
    for (int i = 0; i < LQP_NW; ++i) r += scratch[i];
    return r;
}
// the same for a workgroup of NW waves (scratch >= NW elements).  The partial sums are added in wave order, as wg_sum adds
// them: when the waves beyond NW of a 1024-thread workgroup hold zeros, both give the same bits.
template <int NW, typename T> __device__ __forceinline__ T wg_sum_nw(T v, T* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = T(0);
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NW, typename T> __device__ __forceinline__ T wg_max_nw(T v, T* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = tmax(r, scratch[i]);
    return r;
}
template <typename T> __device__ __forceinline__ T wg_max(T v, T* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = scratch[0];
#pragma unroll
    for (int i = 1; i < LQP_NW; ++i) r = tmax(r, scratch[i]);
    return r;
}

}  // namespace lqp
