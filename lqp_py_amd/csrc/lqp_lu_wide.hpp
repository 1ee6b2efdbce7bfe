// Pivoted LU above the one-row-per-thread tier (1024 < N <= 2048; float32 to 4096: k_lu_factor_wide_tall) on SEVERAL workgroups per matrix, for batches that leave the
// chip idle (B = 8, n = 1500: the one-workgroup kernel of lqp_lu_big.hpp streams the trailing matrix through ONE CU 188 times --
// 25 ms per factorisation, 34 of the 47 ms of a forward + backward step).  Same right-looking algorithm, LAPACK layout and pivot
// rule (replaces torch.linalg.lu_factor at lqp_py/solve_box_qp_admm_torch.py:215,254, lqp_py/lu_layer.py:10,31), float32.
//
// Columns are dealt out in tiles of 32 (one 128-B line per row): tile g belongs to workgroup g % W for the whole factorisation,
// and NOBODY else ever reads or writes those columns of the matrix -- the owner brings them up to date, factors the panels that
// lie in them (PB = 8 columns, two rows per thread, the code of the one-workgroup kernel), applies the interchanges to them when
// they are left of the panel.  What crosses between workgroups is the factored panel alone: L21^T (PB x M2), L11, the gather map
// of the interchanges -- a message in a two-slot ring in global memory, written through (sc1 stores), one tagged `go` word per
// slot, read with sc1 loads (MI355X_MICROARCH R1: the form of lqp_lu2.hpp).  No barrier between the workgroups: a consumer
// waits for the message of panel k, the publisher of panel k waits until every workgroup has taken message k - 2 out of the slot
// it is about to overwrite (one progress word per workgroup).  All W B workgroups must be resident (the host checks).
#pragma once
#include "lqp_lu2.hpp"
#include "lqp_lu_big.hpp"

namespace lqp {

template <typename T> __host__ __device__ constexpr int luw_pb() { return lu_big_panel<T>(); }     // panel width: the one-workgroup kernel's (8 | 4)
template <typename T> __host__ __device__ constexpr int luw_tw() { return 128 / (int)sizeof(T); }   // columns of a tile: one 128-B line per row (32 | 16)
constexpr int LUW_HDR = 64;                   // progress words (W <= 64)
constexpr int LUW_MSG = 128;                  // 4-byte words of a message head: go | ne | zero pivot | - | src[PB] | xdst[PB] | xsrc[PB] | ... | [32 ..) L11[PB (PB + 1)]
constexpr int LUW_L11 = 32;
template <typename T> __host__ __device__ inline size_t luw_slot_words(int Mpad) { return (size_t)LUW_MSG + (size_t)luw_pb<T>() * Mpad * (sizeof(T) / 4); }
template <typename T> __host__ __device__ inline size_t luw_scratch_words(int N) { return (size_t)LUW_HDR + 2 * luw_slot_words<T>(round_up(N, 64)); }

__device__ __forceinline__ unsigned int luw_tag(const unsigned int epoch, const int k) { return (epoch << 12) | (unsigned int)(k + 1); }

// bounded wait for *p == want (thread 0 of the workgroup; the others stand at the barrier behind it)
__device__ __forceinline__ bool luw_wait_eq(const int* p, const unsigned int want, const int* fail, const unsigned int fail_tag) {
    unsigned int spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        if ((unsigned int)ld_sc1(p) == want) return true;
        if ((++spins & 255u) == 0) {
            if ((unsigned int)ld_sc1(fail) == fail_tag) return false;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();         // 100 MHz
            if (t0 == 0) t0 = now;
            else if (now - t0 > 100000000ULL) return false;                          // 1 s
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// PB: panel width, R: panel rows per thread of waves 0..3 (R * 256 rows: 8 -> N <= 2048; float32 above, to 4096: R = 16 with panels of 4
// columns -- the LDS holds 2 * PB * N elements of a panel; the message slots keep the size of the 8-column form)
template <typename T, int PB, int R>
__device__ __forceinline__ void wg_lu_factor_wide(T* __restrict__ Mall, const int Nuni, const int ld, const size_t mstride,
                                                  int* __restrict__ piv, const int pstride, int* __restrict__ info_all,
                                                  const int* __restrict__ gate, const int* __restrict__ Nvec,
                                                  int* __restrict__ scr_all, const size_t scr_stride, const unsigned int epoch,
                                                  const int B, unsigned long long* __restrict__ dbg, char* __restrict__ smem) {
    constexpr int NT = LQP_NT, TW = luw_tw<T>(), NV = PB / 4;
    constexpr int NTP = 256;                  // the panel: waves 0..3, R rows per thread (sixteen waves with two rows each share four SIMDs:
                                              // 7 k cycles per column; four waves alone on theirs: see DESIGN.md)
    static_assert(PB <= luw_pb<T>() && PB % 4 == 0, "the message slots are sized for the widest panel");
    if (gate && *gate == 0) return;
    const int b = (int)blockIdx.x % B, y = (int)blockIdx.x / B, W = (int)gridDim.x / B;
    const int N = Nvec ? Nvec[b] : Nuni;
    T* A = Mall + (size_t)b * mstride;
    int* ipiv = piv + (size_t)b * pstride;
    int* scr = scr_all + (size_t)b * scr_stride;
    const int Mpad = round_up(Nuni, 64);                  // (the message layout follows the launch's size, not the problem's)
    const LuLds<T, PB> L(Mpad);
    T* LT = (T*)(smem + L.lt);
    T* UPt = (T*)(smem + L.up);                           // [own tile slot][PB][32]
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
    int* cnt = (int*)(smem + L.cnt);                      // [0] interchange count | [1] first zero pivot | [2] failed | [3] ne of the message
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    typedef V4<T> vec;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    int* progress = scr;                                  // [W]
    int* failw = scr + LUW_HDR - 1;                       // sticky: somebody gave up waiting (tagged with the epoch)
    const size_t SL = luw_slot_words<T>(Mpad);
    if (tid == 0) { cnt[1] = 0; cnt[2] = 0; }
    if (tid < 2 * LQP_NW) wval[tid] = T(-2);
    if (tid == 0 && y == 0) info_all[b] = 0;
    __syncthreads();
    const unsigned int fail_tag = ((epoch & 0xFFFFFu) << 12) | 0xFFFu;
    int first_zero = 0;
    bool dead = false;
    // debug (lqp_debug_set_lu_counters): cycles of workgroup 1's thread 0 in [0] panel [1] slot wait [2] publish [3] message wait
    // [4] message copy [5] interchanges + U12 [6] trailing update [7] total
    unsigned long long dbt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dts = 0;
    const bool dbg_on = dbg != nullptr && y == (W > 1 ? 1 : 0) && tid == 0;
    const unsigned long long dt_all = dbg_on ? clock64() : 0;
#define LUW_STAMP(i) do { if (dbg_on) { const unsigned long long t_ = clock64(); dbt[i] += t_ - dts; dts = t_; } } while (0)

    const int npanels = (N + PB - 1) / PB;
    for (int k = 0; k < npanels; ++k) {
        const int k0 = k * PB;
        const int pb = (N - k0 < PB) ? (N - k0) : PB;
        const int M = N - k0, M2 = M - pb;
        const int gk = k0 / TW, owner = gk % W;
        int* slot = scr + LUW_HDR + (size_t)(k & 1) * SL;
        T* LTg = (T*)(slot + LUW_MSG);
        int ne = 0;
        if (dbg_on) dts = clock64();
        if (y == owner) {
            // ---- the panel: my own columns, two rows per thread (the one-workgroup kernel's code) ----
            {
                vec row4[R][PB / 4];
                int curpos[R];
                bool done[R];
                const bool wact = wbase < M && tid < NTP;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int r = tid + q * NTP;
                    const bool act = r < M && tid < NTP;
                    curpos[q] = r;
                    done[q] = !act;
                    // (a full panel is two 16-B pieces of a row, 32-B aligned: element by element every lane asked for its line
                    //  eight times -- 2 MB of requests per panel through one CU, 30 k of the panel's 70 k cycles)
                    if (pb == PB) {
                        const vec* rp = (const vec*)(A + (size_t)(k0 + (act ? r : 0)) * ld + k0);
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const vec a = rp[v];
#pragma unroll
                            for (int e = 0; e < 4; ++e) row4[q][v].v[e] = act ? a.v[e] : T(0);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < PB; ++c)
                            row4[q][c >> 2].v[c & 3] = (act && c < pb) ? A[(size_t)(k0 + r) * ld + k0 + c] : T(0);
                    }
                }
                if (tid == 0) *cnt = 0;
                {
                    const PanelLds<T> S{rowP, wval, wrcp, widx, wtid, pidx, cnt};
                    lu_panel_columns_r<T, PB, R, NTP>(row4, curpos, done, wact, pb, k0, S);
                }
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int r = tid + q * NTP;
                    if (r < M && tid < NTP) {
                        const int cp = curpos[q];
                        if (pb == PB) {
                            vec* wp = (vec*)(A + (size_t)(k0 + cp) * ld + k0);
#pragma unroll
                            for (int v = 0; v < NV; ++v) wp[v] = row4[q][v];
                        } else {
#pragma unroll
                            for (int c = 0; c < PB; ++c)
                                if (c < pb) A[(size_t)(k0 + cp) * ld + k0 + c] = row4[q][c >> 2].v[c & 3];
                        }
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
            ne = __builtin_amdgcn_readfirstlane(*cnt);
            LUW_STAMP(0);
            if (W > 1) {
                // ---- publish: nobody may still be reading message k - 2 out of this slot ----
                if (k >= 2 && tid < W) {                   // (a lane per workgroup: one after the other the W polls were 30 us a panel)
                    unsigned int spins = 0;
                    unsigned long long t0 = 0;
                    for (;;) {
                        const unsigned int v = (unsigned int)ld_sc1(progress + tid);
                        if ((v >> 12) == (epoch & 0xFFFFFu) && (int)(v & 0xFFFu) >= k - 1) break;
                        if ((++spins & 255u) == 0) {
                            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                            if ((unsigned int)ld_sc1(failw) == fail_tag) { cnt[2] = 1; break; }
                            if (t0 == 0) t0 = now;
                            else if (now - t0 > 100000000ULL) { cnt[2] = 1; break; }
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
                LUW_STAMP(1);
                if (cnt[2]) { dead = true; break; }
                {   // (16-byte write-through stores: element by element the message was 12 fabric writes per thread)
                    const int nv4 = (M2 + 3) >> 2;
                    for (int e = tid; e < PB * nv4; e += NT) {
                        const int c = e / nv4, i4 = e - c * nv4;
                        st_vec_sc1(LTg + (size_t)c * Mpad + 4 * i4, *(const vec*)(LT + c * Mpad + 4 * i4));
                    }
                }
                if (tid < PB * (PB + 1)) st_sc1((T*)(slot + LUW_L11) + tid, L11[tid]);
                if (tid < PB) { st_sc1(slot + 4 + tid, src[tid]); st_sc1(slot + 4 + PB + tid, tid < ne ? xdst[tid] : 0); st_sc1(slot + 4 + 2 * PB + tid, tid < ne ? xsrc[tid] : 0); }
                if (tid == 0) { st_sc1(slot + 1, ne); st_sc1(slot + 2, cnt[1]); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) { st_sc1(slot, (int)luw_tag(epoch & 0xFFFFFu, k)); st_sc1(progress + y, (int)luw_tag(epoch & 0xFFFFFu, k)); }
                LUW_STAMP(2);
            }
            if (first_zero == 0 && cnt[1] != 0) first_zero = cnt[1];
        } else {
            // ---- take the message of panel k ----
            if (tid == 0) {
                if (!luw_wait_eq(slot, luw_tag(epoch & 0xFFFFFu, k), failw, fail_tag)) cnt[2] = 1;
            }
            __syncthreads();
            LUW_STAMP(3);
            if (cnt[2]) { dead = true; break; }
            {
                const int nv4 = (M2 + 3) >> 2;
                for (int e = tid; e < PB * nv4; e += NT) {
                    const int c = e / nv4, i4 = e - c * nv4;
                    *(vec*)(LT + c * Mpad + 4 * i4) = ld_vec_sc1(LTg + (size_t)c * Mpad + 4 * i4);
                }
            }
            if (tid < PB * (PB + 1)) L11[tid] = ld_sc1((const T*)(slot + LUW_L11) + tid);
            if (tid < PB) { src[tid] = ld_sc1(slot + 4 + tid); xdst[tid] = ld_sc1(slot + 4 + PB + tid); xsrc[tid] = ld_sc1(slot + 4 + 2 * PB + tid); }
            if (tid == 0) { cnt[3] = ld_sc1(slot + 1); const int z = ld_sc1(slot + 2); if (z != 0 && cnt[1] == 0) cnt[1] = z; }
            __syncthreads();
            ne = __builtin_amdgcn_readfirstlane(cnt[3]);
            if (first_zero == 0 && cnt[1] != 0) first_zero = cnt[1];
            if (tid == 0) st_sc1(progress + y, (int)luw_tag(epoch & 0xFFFFFu, k));
            LUW_STAMP(4);
        }
        __syncthreads();
        bool anyswap = ne > 0;
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (j < pb && __builtin_amdgcn_readfirstlane(src[j]) != j) anyswap = true;

        // ---- my tiles: interchanges left and right of the panel, U12 = L11^-1 (P A)12 (a column per thread, 32 tiles at a time) ----
        const int ntile = (N + TW - 1) / TW;
        for (int q = tid / TW; y + q * W < ntile; q += NT / TW) {
            const int g = y + q * W, col = g * TW + (tid % TW);
            if (col >= N || (col >= k0 && col < k0 + pb)) continue;
            const bool right = col >= k0 + pb;
            if (!right && !anyswap) continue;
            T* Ac = A + (size_t)k0 * ld + col;
            T top[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) top[j] = (j < pb) ? Ac[(size_t)src[j] * ld] : T(0);
            {
                T ext[PB];
#pragma unroll
                for (int e = 0; e < PB; ++e) ext[e] = (e < ne) ? Ac[(size_t)xsrc[e] * ld] : T(0);
#pragma unroll
                for (int e = 0; e < PB; ++e)
                    if (e < ne) Ac[(size_t)xdst[e] * ld] = ext[e];
            }
            if (right) {
#pragma unroll
                for (int j = 0; j < PB; ++j) {
#pragma unroll
                    for (int i = j + 1; i < PB; ++i) top[i] -= L11[i * (PB + 1) + j] * top[j];
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) UPt[(q * PB + j) * TW + (tid % TW)] = top[j];
            }
#pragma unroll
            for (int j = 0; j < PB; ++j)
                if (j < pb && (right || src[j] != j)) Ac[(size_t)j * ld] = top[j];
        }
        __syncthreads();
        LUW_STAMP(5);

        // ---- trailing update of my tiles: A22[:, tile] -= L21 U12[:, tile] on the matrix cores, a tile per wave (float32: 32 x 32
        //      v_mfma_f32_32x32x2; float64: 16 x 16, ONE v_mfma_f64_16x16x4 per tile -- its depth is the panel's width) ----
        if (M2 > 0) {
            const int c_lo = k0 + pb;
            const int q0 = (gk >= y) ? (gk - y + W - 1) / W : 0;          // my first tile at or right of the panel's
            int nq = 0;
            for (int q = q0; y + q * W < ntile; ++q) ++nq;
            if constexpr (sizeof(T) == 4) {
                const int nti = (M2 + 31) >> 5;
                const int li = lane & 31, lh = lane >> 5;
                for (int t = __builtin_amdgcn_readfirstlane(w); t < nq * nti; t += NT / 64) {
                    const int qq = q0 + t / nti, ti = t % nti;
                    const int g = y + qq * W, c0 = g << 5;
                    if (c0 + 31 < c_lo) continue;                              // (the panel's own tile may have nothing right of it)
                    const int i0 = ti << 5;
                    const int col = c0 + li;
                    const bool colok = col >= c_lo && col < N;
                    const int rlim = M2 - i0 - 4 * lh;
                    float* base = (float*)A + (size_t)(c_lo + i0) * ld + c0;
                    const int voff = 4 * lh * ld + li;
                    f32x16 cur;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int qrow = (q & 3) + 8 * (q >> 2);
                        cur[q] = (colok && qrow < rlim) ? base[(size_t)qrow * ld + voff] : 0.f;
                    }
                    const float* lt = (const float*)LT + i0 + li + lh * Mpad;
                    const float* up = (const float*)UPt + (qq * PB + lh) * 32 + li;
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                    for (int kk = 0; kk < PB; kk += 2) {
                        const float a = (i0 + li < M2) ? lt[kk * Mpad] : 0.f;
                        const float bq = colok ? up[kk * 32] : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
                    }
                    cur -= acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int qrow = (q & 3) + 8 * (q >> 2);
                        if (colok && qrow < rlim) base[(size_t)qrow * ld + voff] = cur[q];
                    }
                }
            } else {
                // (operand and result layout of v_mfma_f64_16x16x4_f64 as measured, tools/microbench/mfma_f64_layout.hip:
                //  A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D register q <-> row 4 q + (l >> 4), column l & 15)
                static_assert(sizeof(T) == 4 || PB == 4, "one matrix instruction per tile");
                const int nti = (M2 + 15) >> 4;
                const int li = lane & 15, lg = lane >> 4;
                for (int t = __builtin_amdgcn_readfirstlane(w); t < nq * nti; t += NT / 64) {
                    const int qq = q0 + t / nti, ti = t % nti;
                    const int g = y + qq * W, c0 = g << 4;
                    if (c0 + 15 < c_lo) continue;
                    const int i0 = ti << 4;
                    const int col = c0 + li;
                    const bool colok = col >= c_lo && col < N;
                    double* base = (double*)A + (size_t)(c_lo + i0) * ld + c0 + li;
                    f64x4 cur;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 4 * q + lg;
                        cur[q] = (colok && i0 + row < M2) ? base[(size_t)row * ld] : 0.0;
                    }
                    const double a = (i0 + li < M2) ? ((const double*)LT)[lg * Mpad + i0 + li] : 0.0;
                    const double bq = colok ? ((const double*)UPt)[(qq * PB + lg) * 16 + li] : 0.0;
                    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc, 0, 0, 0);
                    cur -= acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 4 * q + lg;
                        if (colok && i0 + row < M2) base[(size_t)row * ld] = cur[q];
                    }
                }
            }
        }
        __syncthreads();
        LUW_STAMP(6);
    }
    if (dbg_on) {
        dbt[7] = clock64() - dt_all;
        for (int q = 0; q < 8; ++q) dbg[(size_t)b * 16 + q] = dbt[q];
    }
#undef LUW_STAMP
    if (dead) {
        if (tid == 0) { st_sc1(failw, (int)fail_tag); info_all[b] = -7; }
        return;
    }
    if (tid == 0 && y == 0 && first_zero != 0) info_all[b] = first_zero;
}

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_lu_factor_wide(T* __restrict__ Mall, const int Nuni, const int ld, const size_t mstride,
                                                           int* __restrict__ piv, const int pstride, int* __restrict__ info_all,
                                                           const int* __restrict__ gate, const int* __restrict__ Nvec,
                                                           int* __restrict__ scr_all, const size_t scr_stride, const unsigned int epoch,
                                                           const int B, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    wg_lu_factor_wide<T, luw_pb<T>(), 8>(Mall, Nuni, ld, mstride, piv, pstride, info_all, gate, Nvec, scr_all, scr_stride, epoch, B, dbg, smem);
}

// 2048 < N <= 4096, float32: sixteen panel rows per thread, panels of 4 columns
template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_lu_factor_wide_tall(T* __restrict__ Mall, const int Nuni, const int ld, const size_t mstride,
                                                                int* __restrict__ piv, const int pstride, int* __restrict__ info_all,
                                                                const int* __restrict__ gate, const int* __restrict__ Nvec,
                                                                int* __restrict__ scr_all, const size_t scr_stride, const unsigned int epoch,
                                                                const int B, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    wg_lu_factor_wide<T, 4, 16>(Mall, Nuni, ld, mstride, piv, pstride, info_all, gate, Nvec, scr_all, scr_stride, epoch, B, dbg, smem);
}

}  // namespace lqp
