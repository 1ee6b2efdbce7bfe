// `unroll=True` on the GPU: the backward pass of the UNROLLED ADMM loop as one reverse sweep.
//
// The reference differentiates its loop by letting autograd tape every iteration (lqp_py/solve_box_qp_admm_torch.py:14-15,
// 216-219, 255-256, 264-265): the x-update of iteration k is TorchLULayer (lqp_py/lu_layer.py:25-58), whose backward is
//     dx_k = M^-1 (-xbar_k)  (the cached factor, "only works for symmetric A"),   Mbar += dx_k xv_k^T,   rhsbar_k = -dx_k,
// and everything else of an iteration is element-wise:
//     rhs_k = [-p + rho (z_k - u_k); b],  x_k = (M^-1 rhs_k)[:n],  z_{k+1} = min(max(x_k + u_k, lb), ub),
//     u_{k+1} = u_k + x_k - z_{k+1}                                                         (:258-282, scaled space).
// With a constant factor (no adaptive-rho refactorisation happened) the tape collapses to a recurrence that needs, per
// iteration, ONE solve with the cached factor -- on the symmetric path one product with the packed inverse H the forward
// left in its workspace -- and element-wise work.  k_unroll_sweep (one workgroup per QP) first REPLAYS the forward loop for
// the T recorded x-updates (the same product, the same element-wise code: x_k, z_k - u_k and the clamp decisions go to
// scratch), then walks back k = T-1 .. 0.  What comes out are the gradients w.r.t. the SCALED problem the loop ran on
// (Qs, ps, As, bs, lbs, ubs, rho) and w.r.t. D through x = D x_T; the host chains them through the scaling (:160-203).
//   Qsbar = sum_k dxx_k x_k^T (k_unroll_outer),   rhobar = tr(Qsbar) - sum_k dxx_k . (z_k - u_k),   psbar = sum_k dxx_k,
//   Asbar = sum_k dnu_k x_k^T + nu_k dxx_k^T,   bsbar = -sum_k dnu_k,   lbsbar / ubsbar: what the clamps kept.
// A clamp that sits EXACTLY on its bound (x_k + u_k == lb: torch.maximum splits the gradient in two) is treated as free.
#pragma once
#include "lqp_boxqp.hpp"
#include "lqp_dense.hpp"

namespace lqp {

struct UnrollParams {
    int T;                  // recorded x-updates = iters + 1
    int rl;                 // LDS-resident blocks of the product (host: unroll_lds_blocks)
    const float* g;         // (B, n): dL/dx of the returned (unscaled) solution
    float *X, *W, *DX;      // (B, T, n) scratch: x_k | z_k - u_k | x part of dx_k
    float* NU;              // (B, T, m) scratch: nu_k
    signed char* MK;        // (B, T, n) scratch: -1 / 0 / +1 = z_{k+1} sits on lb / is free / sits on ub
    float *dps, *dlbs, *dubs, *dD;     // (B, n) outputs
    float *dAs, *dbs;       // (B, m, n), (B, m) outputs (or null when m == 0)
    float* drho;            // (B) output
};

// LDS: [rl blocks] v | ylds | part[NW][Nps] | cvl | nus[m] | dnu[m] | red[NW]
__host__ __device__ inline int unroll_lds_bytes(int m, int Ks, int rl) {
    const int Nps = Ks * LQP_NB;
    return (rl * LQP_BLK + 2 * Nps + sym_blocks(Ks) * 64 + LQP_NW * Nps + 2 * (m > 0 ? m : 1) + LQP_NW + 16) * 4 + 64;
}
__host__ __device__ inline int unroll_lds_blocks(int m, int Ks) {
    int rl = sym_blocks(Ks) - LQP_RREG;
    if (rl < 0) rl = 0;
    while (rl > 0 && unroll_lds_bytes(m, Ks, rl) > 160 * 1024) --rl;
    return rl;
}

// MA: equality rows the per-thread Asbar accumulators are built for (1: the benchmark's single row -- sixteen of them cost
// the product its registers; SPD_MAXM otherwise)
template <int MA = SPD_MAXM>
__global__ __launch_bounds__(LQP_NT) void k_unroll_sweep(const FwdParams<float> P, const UnrollParams U) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NT = LQP_NT;
    const int b = blockIdx.x, n = P.n, m = P.m, Ks = P.Ks, Nps = Ks * LQP_NB, T = U.T, rl = U.rl;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int S = sym_blocks(Ks);
    float* lds_res = (float*)smem;
    float* v = lds_res + (size_t)rl * LQP_BLK;
    float* ylds = v + Nps;
    float* part = ylds + S * 64;
    float* cvl = part + (size_t)LQP_NW * Nps;
    float* nus = cvl + Nps;
    float* dnul = nus + (m > 0 ? m : 1);
    float* red = dnul + (m > 0 ? m : 1);
    VecView<float> V(P.vecs + (size_t)b * P.vstride, n, m);
    const float rho = P.scal[(size_t)b * SC_WORDS + SC_RHO];
    const float* packed = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
    float* X = U.X + (size_t)b * T * n;
    float* Wd = U.W + (size_t)b * T * n;
    float* DX = U.DX + (size_t)b * T * n;
    float* NU = U.NU + (size_t)b * T * (m > 0 ? m : 1);
    signed char* MK = U.MK + (size_t)b * T * n;

    BlockStream<float, NT> st;
    ResidentRegs<float, NT> rr;
    sym_resident_load<NT>(rr, lds_res, packed, S, rl);
    sym_prime<NT>(st, packed, (S < resident_regs<NT>() ? S : resident_regs<NT>()) + rl, S);

    // the symmetric path holds n <= 1024: element i = tid lives in this thread's registers for the whole kernel
    const int i = tid;
    const bool live = i < n;
    const float psi = live ? V.ps[i] : 0.f, lbi = live ? V.lbs[i] : 0.f, ubi = live ? V.ubs[i] : 0.f;
    for (int e = tid; e < Nps; e += NT) cvl[e] = (e < n && m > 0) ? V.cv[e] : 0.f;

    // ---- replay of the forward loop (:258-282): x_k, z_k - u_k, nu_k and the clamp decisions ----
    float zi = 0.f, ui = 0.f;
    for (int k = 0; k < T; ++k) {
        const float wd = zi - ui;
        for (int e = tid; e < Nps; e += NT) v[e] = (e == i && live) ? -psi + rho * wd : 0.f;
        if (live) Wd[(size_t)k * n + i] = wd;
        wg_barrier_lds();
        wg_sym_gemv<true, NT>(st, rr, lds_res, rl, packed, Ks, Nps, v, ylds, part);
        wg_barrier_lds();
        for (int r = w; r < m; r += LQP_NW) {            // nu_k = T^T w - s0 (one wave per row) while v still holds w
            float acc = 0.f;
            for (int e = lane; e < n; e += 64) acc += V.Tm[(size_t)r * n + e] * v[e];
            acc = wave_sum(acc);
            if (lane == 0) NU[(size_t)k * m + r] = acc - V.s0[r];
        }
        if (live) {
            const float xi = cvl[i] - sym_combine<NT>(i, Ks, Nps, ylds, part);
            X[(size_t)k * n + i] = xi;
            const float s = xi + ui;
            const float zn = tmin(tmax(s, lbi), ubi);
            MK[(size_t)k * n + i] = (signed char)(tmax(s, lbi) > ubi ? 1 : (s < lbi ? -1 : 0));      // (maximum first, then minimum: :273-276)
            ui = ui + (xi - zn);
            zi = zn;
        }
        wg_barrier_lds();
    }

    // ---- reverse sweep ----
    const float gi = live ? U.g[(size_t)b * n + i] : 0.f;
    const float di = live ? V.D[i] : 0.f;
    float ubar = 0.f, zbar = 0.f, pbar = 0.f, lbbar = 0.f, ubbar = 0.f, rho_part = 0.f, bbar = 0.f;
    float dA[MA];
#pragma unroll
    for (int q = 0; q < MA; ++q) dA[q] = 0.f;
    __syncthreads();                                        // (the scratch rows written above: read below by their writers
                                                            //  only, except NU -- written by lane 0 of a wave, read by all)
    for (int k = T - 1; k >= 0; --k) {
        float unew = 0.f;
        {
            const int code = live ? (int)MK[(size_t)k * n + i] : 0;
            const float zt = zbar - ubar;
            const float wfree = code == 0 ? zt : 0.f;
            lbbar += code < 0 ? zt : 0.f;
            ubbar += code > 0 ? zt : 0.f;
            unew = wfree + ubar;
            const float xb = unew + (k == T - 1 ? di * gi : 0.f);
            for (int e = tid; e < Nps; e += NT) v[e] = (e == i && live) ? xb : 0.f;
        }
        wg_barrier_lds();
        wg_sym_gemv<true, NT>(st, rr, lds_res, rl, packed, Ks, Nps, v, ylds, part);
        wg_barrier_lds();
        if (m > 0) {
            for (int r = w; r < m; r += LQP_NW) {        // nu part of dx_k: -T^T xbar
                float acc = 0.f;
                for (int e = lane; e < n; e += 64) acc += V.Tm[(size_t)r * n + e] * v[e];
                acc = wave_sum(acc);
                if (lane == 0) { dnul[r] = -acc; nus[r] = NU[(size_t)k * m + r]; }
            }
            wg_barrier_lds();
        }
        if (live) {
            const float dxx = sym_combine<NT>(i, Ks, Nps, ylds, part);      // packed = -H:  -H xbar
            const float xk = X[(size_t)k * n + i];
            DX[(size_t)k * n + i] = dxx;
            pbar += dxx;
            rho_part += dxx * (xk - Wd[(size_t)k * n + i]);
            zbar = -rho * dxx;
            ubar = unew + rho * dxx;
#pragma unroll
            for (int q = 0; q < MA; ++q)
                if (q < m) dA[q] += dnul[q] * xk + nus[q] * dxx;
        }
        if (tid < m) bbar -= dnul[tid];
        wg_barrier_lds();
    }
    if (live) {
        U.dps[(size_t)b * n + i] = pbar;
        U.dlbs[(size_t)b * n + i] = lbbar;
        U.dubs[(size_t)b * n + i] = ubbar;
        U.dD[(size_t)b * n + i] = gi * X[(size_t)(T - 1) * n + i];
#pragma unroll
        for (int q = 0; q < MA; ++q)
            if (q < m) U.dAs[((size_t)b * m + q) * n + i] = dA[q];
    }
    if (tid < m) U.dbs[(size_t)b * m + tid] = bbar;
    const float rsum = wg_sum(rho_part, red);
    if (tid == 0) U.drho[b] = rsum;
}

// The same sweep on TWO workgroups per QP (2 B <= #CUs, 3 <= Ks <= 8): the products are those of k_admm_loop_split -- each
// workgroup holds its half of the blocks of H in registers / LDS for the whole launch (wg_sym_gemv_split), the partial vectors
// cross as tagged 8-byte granules through the forward's exchange area (two buffers by step parity; tag = 0x10000000 | run | step:
// apart from the loop's own) and are added in the fixed order part 0 + part 1 -- so both workgroups hold bit-identical iterates
// and run all element-wise work redundantly; both write the (identical) scratch rows they read back, part 0 the results.  One
// workgroup per QP streams two thirds of H per product from beyond the L2: 122 products = 1.6 ms at the headline size.
// LDS (floats): [rl blocks] v yrow cvl part[8][Nps] | nus[m] dnu[m] red[8 + 8] | flags[8]
template <int KS> __host__ __device__ inline int unroll_split_lds_bytes(int m) {
    const int mm = m > 0 ? m : 1;
    return (split_lds_blocks<512, 2>(KS) * LQP_BLK + (3 + 8) * KS * LQP_NB + 2 * mm + 16 + 8) * 4 + 64;
}
template <int KS, int MA>
__global__ __launch_bounds__(512) void k_unroll_sweep_split(const FwdParams<float> P, const UnrollParams U, const unsigned int run) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NT = 512, NP = 2, NWV = NT / 64, Nps = KS * LQP_NB, rl = split_lds_blocks<NT, NP>(KS);
    constexpr int XPART = SPD_MAXK * LQP_NB, XPAR = NP * XPART;
    int b, part_id;
    if (!shared_map((int)blockIdx.x, P.B, NP, b, part_id)) return;
    const int n = P.n, m = P.m, T = U.T;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* const lds_res = (float*)smem;
    float* const v = lds_res + (size_t)rl * LQP_BLK;
    float* const yrow = v + Nps;
    float* const cvl = yrow + Nps;
    float* const part = cvl + Nps;
    float* const nus = part + (size_t)NWV * Nps;
    float* const dnul = nus + (m > 0 ? m : 1);
    float* const red = dnul + (m > 0 ? m : 1);
    int* const flags = (int*)(red + 16);
    VecView<float> V(P.vecs + (size_t)b * P.vstride, n, m);
    const float rho = P.scal[(size_t)b * SC_WORDS + SC_RHO];
    const float* packed = P.packed + (size_t)b * packed_blocks(P.K) * LQP_BLK;
    unsigned long long* xq = P.xchg + (size_t)b * XCHG_WORDS;
    float* X = U.X + (size_t)b * T * n;
    float* Wd = U.W + (size_t)b * T * n;
    float* DX = U.DX + (size_t)b * T * n;
    float* NU = U.NU + (size_t)b * T * (m > 0 ? m : 1);
    signed char* MK = U.MK + (size_t)b * T * n;

    SplitResident<NT> rr;
#define LQP_BY_PART2(CALL) do { if (part_id == 0) { constexpr int PARTC = 0; CALL; } else { constexpr int PARTC = 1; CALL; } } while (0)
    LQP_BY_PART2((split_resident_load<KS, PARTC, NT, NP>(rr, lds_res, packed)));
    for (int e = tid; e < Nps; e += NT) { cvl[e] = (e < n && m > 0) ? V.cv[e] : 0.f; yrow[e] = 0.f; }
    for (int e = tid; e < NWV * Nps; e += NT) part[e] = 0.f;
    if (tid < 8) flags[tid] = 0;
    const int i = tid;                                       // (Nps <= 512: element i of every vector lives in thread i)
    const bool live = i < n;
    const float psi = live ? V.ps[i] : 0.f, lbi = live ? V.lbs[i] : 0.f, ubi = live ? V.ubs[i] : 0.f;
    __syncthreads();

    // y = (this workgroup's partial) + (the partner's), in the order part 0 + part 1; step: the number of the product in the launch
    auto full_product = [&](const int step) -> float {
        LQP_BY_PART2((wg_sym_gemv_split<KS, PARTC, NT, NP>(rr, lds_res, Nps, v, yrow, part)));
        wg_barrier_lds();
        float y = 0.f;
        if (tid < Nps) {
            const float own = split_combine<NT>(tid, Nps, yrow, part);
            const unsigned int tag = 0x10000000u | ((run & 0xFFu) << 16) | (unsigned int)(step & 0xFFFF);
            unsigned long long* base = xq + (size_t)(step & 1) * XPAR;
            __hip_atomic_store(base + (size_t)part_id * XPART + tid,
                               ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned int, own),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long* src = base + (size_t)(1 - part_id) * XPART + tid;
            unsigned long long g = 0;
            if (!flags[0]) {
                unsigned int spins = 0;
                unsigned long long t0 = 0;
                for (;;) {
                    g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned int)(g >> 32) == tag) break;
                    if ((++spins & 1023u) == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();     // 100 MHz
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > 50000000ULL) {                                   // 0.5 s: give up, flagged
                            __hip_atomic_store(P.status + ST_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            flags[0] = 1;
                            break;
                        }
                    }
                }
            }
            // (a partner that never showed up -- the wait is bounded -- must not leave plausible numbers: NaN from here on, through
            //  every later product into every gradient of this problem, and the time-out word for whoever reads the status)
            const float other = flags[0] ? __builtin_nanf("") : __builtin_bit_cast(float, (unsigned int)g);
            y = part_id == 0 ? own + other : other + own;
        }
        return y;
    };

    // ---- replay of the forward loop (:258-282) ----
    float zi = 0.f, ui = 0.f;
    for (int k = 0; k < T; ++k) {
        const float wd = zi - ui;
        if (tid < Nps) v[tid] = live ? -psi + rho * wd : 0.f;
        if (live) Wd[(size_t)k * n + i] = wd;
        wg_barrier_lds();
        const float y = full_product(k);
        for (int r = w; r < m; r += NWV) {                   // nu_k = T^T w - s0 (one wave per row) while v still holds w
            float acc = 0.f;
            for (int e = lane; e < n; e += 64) acc += V.Tm[(size_t)r * n + e] * v[e];
            acc = wave_sum(acc);
            if (lane == 0) NU[(size_t)k * m + r] = acc - V.s0[r];
        }
        if (live) {
            const float xi = cvl[i] - y;
            X[(size_t)k * n + i] = xi;
            const float s = xi + ui;
            const float zn = tmin(tmax(s, lbi), ubi);
            MK[(size_t)k * n + i] = (signed char)(tmax(s, lbi) > ubi ? 1 : (s < lbi ? -1 : 0));
            ui = ui + (xi - zn);
            zi = zn;
        }
        wg_barrier_lds();
    }

    // ---- reverse sweep ----
    const float gi = live ? U.g[(size_t)b * n + i] : 0.f;
    const float di = live ? V.D[i] : 0.f;
    float ubar = 0.f, zbar = 0.f, pbar = 0.f, lbbar = 0.f, ubbar = 0.f, rho_part = 0.f, bbar = 0.f;
    float dA[MA];
#pragma unroll
    for (int q = 0; q < MA; ++q) dA[q] = 0.f;
    __syncthreads();
    for (int k = T - 1; k >= 0; --k) {
        float unew = 0.f;
        {
            const int code = live ? (int)MK[(size_t)k * n + i] : 0;
            const float zt = zbar - ubar;
            const float wfree = code == 0 ? zt : 0.f;
            lbbar += code < 0 ? zt : 0.f;
            ubbar += code > 0 ? zt : 0.f;
            unew = wfree + ubar;
            const float xb = unew + (k == T - 1 ? di * gi : 0.f);
            if (tid < Nps) v[tid] = live ? xb : 0.f;
        }
        wg_barrier_lds();
        const float dxx = full_product(T + (T - 1 - k));       // packed = -H:  -H xbar
        if (m > 0) {
            for (int r = w; r < m; r += NWV) {               // nu part of dx_k: -T^T xbar
                float acc = 0.f;
                for (int e = lane; e < n; e += 64) acc += V.Tm[(size_t)r * n + e] * v[e];
                acc = wave_sum(acc);
                if (lane == 0) { dnul[r] = -acc; nus[r] = NU[(size_t)k * m + r]; }
            }
            wg_barrier_lds();
        }
        if (live) {
            const float xk = X[(size_t)k * n + i];
            DX[(size_t)k * n + i] = dxx;
            pbar += dxx;
            rho_part += dxx * (xk - Wd[(size_t)k * n + i]);
            zbar = -rho * dxx;
            ubar = unew + rho * dxx;
#pragma unroll
            for (int q = 0; q < MA; ++q)
                if (q < m) dA[q] += dnul[q] * xk + nus[q] * dxx;
        }
        if (tid < m) bbar -= dnul[tid];
        wg_barrier_lds();
    }
#undef LQP_BY_PART2
    const float rsum = wg_sum_nw<NWV>(rho_part, red);
    if (part_id != 0) return;
    if (live) {
        U.dps[(size_t)b * n + i] = pbar;
        U.dlbs[(size_t)b * n + i] = lbbar;
        U.dubs[(size_t)b * n + i] = ubbar;
        U.dD[(size_t)b * n + i] = gi * X[(size_t)(T - 1) * n + i];
#pragma unroll
        for (int q = 0; q < MA; ++q)
            if (q < m) U.dAs[((size_t)b * m + q) * n + i] = dA[q];
    }
    if (tid < m) U.dbs[(size_t)b * m + tid] = bbar;
    if (tid == 0) U.drho[b] = rsum;
}

// Qsbar[b] = sum_k DX[b][k][:]^T X[b][k][:]  (n x n, T terms): 64 x 64 tile per 256-thread workgroup, 4 x 4 per thread
template <int LQP_ANY = 0>
__global__ __launch_bounds__(256) void k_unroll_outer(const float* __restrict__ DXall, const float* __restrict__ Xall,
                                                      float* __restrict__ out, const int n, const int T) {
    __shared__ float sa[16][64 + 4], sb[16][64 + 4];
    const int b = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const float* DX = DXall + (size_t)b * T * n;
    const float* X = Xall + (size_t)b * T * n;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    for (int k0 = 0; k0 < T; k0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int kk = e >> 6, j = e & 63;
            const bool ok = k0 + kk < T;
            sa[kk][j] = (ok && r0 + j < n) ? DX[(size_t)(k0 + kk) * n + r0 + j] : 0.f;
            sb[kk][j] = (ok && c0 + j < n) ? X[(size_t)(k0 + kk) * n + c0 + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = sa[kk][ty * 4 + a]; bv[a] = sb[kk][tx * 4 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] += av[a] * bv[c];
        }
        __syncthreads();
    }
    float* o = out + (size_t)b * n * n;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = r0 + ty * 4 + a;
        if (r < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = c0 + tx * 4 + c;
                if (cc < n) o[(size_t)r * n + cc] = acc[a][c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same reverse sweep where the x-update is the PIVOTED LU of the KKT matrix (float64, more than 16 equality rows, a
// non-symmetric Q, linsolve = 'lu'): the tape's node is TorchLULayer itself (lqp_py/lu_layer.py:25-58 -- forward lu_solve with the
// cached factor, backward dxv_k = M^-1 (-xvbar_k) with the SAME factor, Mbar += dxv_k xv_k^T, rhsbar_k = -dxv_k), xv = [x; nu].
// One workgroup per QP streams the packed factor (lqp_trsv.hpp) through two triangular solves per x-update: T solves to replay
// the loop, T to walk it back -- the launches and the host round trips of the eager tape (~30 torch ops per iteration, a
// rocBLAS product per check) are gone.  n + m to the LU tier's limit: a thread keeps four elements of every n-vector in registers.
//   Qsbar = sum_k dxx_k x_k^T (k_unroll_outer / k_unroll_outer_any),  Asbar = sum_k dnu_k x_k^T + nu_k dxx_k^T and bsbar = -sum_k dnu_k
//   (k_unroll_lu_eq, from the scratch rows), the rest as above.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct UnrollLuParams {
    int T_;                 // recorded x-updates = iters + 1
    const T* g;             // (B, n): dL/dx of the returned (unscaled) solution
    T *X, *W, *DX;          // (B, T, n) scratch: x_k | z_k - u_k | x part of dxv_k
    T *NU, *DNU;            // (B, T, m) scratch: nu_k | nu part of dxv_k
    signed char* MK;        // (B, T, n) scratch: the clamp decisions
    T *dps, *dlbs, *dubs, *dD;     // (B, n) outputs
    T *dAs, *dbs;           // (B, m, n), (B, m) outputs (or null when m == 0)
    T* drho;                // (B) output
    // ---- a tape in SEGMENTS (a solve in which rho was adapted: one factor per epoch, lqp_boxqp_unroll_tape_segment) ----
    int k0, k1;             // x-updates [k0, k1) of the T_ recorded ones (k1 == 0: all of them)
    int mode;               // bit 0: replay the segment, bit 1: walk it back (0: both, the whole tape)
    const T* packed_ov;     // the epoch's packed factor (lqp_lu_pack layout, blocks) or null: the forward's
    const int* dest_ov;     // ... and its row permutation
    const T* rho_ov;        // (B) the epoch's rho, or null: the forward's
    T* state;               // (B, 2, n): replay -- (z, u) in at k0 > 0, out at k1;  reverse -- (zbar, ubar) in at k1 < T_, out at k0
    T *Zr, *Ur;             // (B, T, n) or null: z_{k+1}, u_{k+1} of every replayed x-update (what the rho adaptation reads)
    int inj_k;              // reverse: the x-update whose cotangents take `inj` (-1: none)
    const T* inj;           // (B, 4, n): added to the cotangents of x_c | z_{c+1} | u_{c+1} | z_c
};
// LDS: v[Np] | tmp[64] | red[NW] | dest[Np] (int)
template <typename T> __host__ __device__ inline int unroll_lu_lds_bytes(int Np) { return (Np + 64 + LQP_NW) * (int)sizeof(T) + Np * 4; }

template <typename T>
__global__ __launch_bounds__(LQP_NT) void k_unroll_sweep_lu(const FwdParams<T> P, const UnrollLuParams<T> U) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NT = LQP_NT, EPT = 4;               // n + m <= 4096: element tid + q NT of every vector, q < 4
    const int b = blockIdx.x, n = P.n, m = P.m, N = P.N, K = P.K, Np = P.Np, TT = U.T_;
    const int tid = threadIdx.x;
    T* v = (T*)smem;
    T* tmp = v + Np;
    T* red = tmp + 64;
    int* dest = (int*)(red + LQP_NW);
    VecView<T> V(P.vecs + (size_t)b * P.vstride, n, m);
    const T rho = U.rho_ov ? U.rho_ov[b] : P.scal[(size_t)b * SC_WORDS + SC_RHO];
    const T* packed = U.packed_ov ? U.packed_ov + (size_t)b * packed_blocks(K) * LQP_BLK : P.packed + (size_t)b * packed_blocks(K) * LQP_BLK;
    const int mm = m > 0 ? m : 1;
    T* X = U.X + (size_t)b * TT * n;
    T* Wd = U.W + (size_t)b * TT * n;
    T* DX = U.DX + (size_t)b * TT * n;
    T* NU = U.NU + (size_t)b * TT * mm;
    T* DNU = U.DNU + (size_t)b * TT * mm;
    signed char* MK = U.MK + (size_t)b * TT * n;
    const int k0 = U.k1 > 0 ? U.k0 : 0, k1 = U.k1 > 0 ? U.k1 : TT;
    const bool do_replay = U.mode == 0 || (U.mode & 1), do_reverse = U.mode == 0 || (U.mode & 2);
    T* stz = U.state ? U.state + (size_t)b * 2 * n : nullptr;

    BlockStream<T, NT> st;
    stream_prime<T, NT>(st, packed, K * (K + 1));
    {
        const int* gdest = U.dest_ov ? U.dest_ov + (size_t)b * Np : P.dest + (size_t)b * Np;
        for (int i = tid; i < Np; i += NT) dest[i] = gdest[i];
    }
    T psi[EPT], lbi[EPT], ubi[EPT], zi[EPT], ui[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid + q * NT;
        const bool live = i < n;
        psi[q] = live ? V.ps[i] : T(0); lbi[q] = live ? V.lbs[i] : T(0); ubi[q] = live ? V.ubs[i] : T(0);
        zi[q] = (live && k0 > 0 && stz && do_replay) ? stz[i] : T(0);
        ui[q] = (live && k0 > 0 && stz && do_replay) ? stz[n + i] : T(0);
    }
    __syncthreads();

    // ---- replay of the forward loop (:258-282): x_k, nu_k, z_k - u_k and the clamp decisions ----
    for (int k = k0; k < k1 && do_replay; ++k) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            if (i < Np) {
                T val = T(0);
                if (i < n) {
                    const T wd = zi[q] - ui[q];
                    Wd[(size_t)k * n + i] = wd;
                    val = -psi[q] + rho * wd;
                } else if (i < N) val = V.bs[i - n];
                v[dest[i]] = val;
            }
        }
        wg_barrier_lds();
        wg_packed_solve<T, NT>(st, packed, K, v, tmp, true);
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            if (i < n) {
                const T xi = v[i];
                X[(size_t)k * n + i] = xi;
                const T s = xi + ui[q];
                const T zn = tmin(tmax(s, lbi[q]), ubi[q]);
                MK[(size_t)k * n + i] = (signed char)(tmax(s, lbi[q]) > ubi[q] ? 1 : (s < lbi[q] ? -1 : 0));      // (maximum first, then minimum: :273-276)
                ui[q] = ui[q] + (xi - zn);
                zi[q] = zn;
                if (U.Zr) { U.Zr[((size_t)b * TT + k) * n + i] = zn; U.Ur[((size_t)b * TT + k) * n + i] = ui[q]; }
            } else if (i < N) NU[(size_t)k * m + (i - n)] = v[i];
        }
        wg_barrier_lds();
    }
    if (do_replay && stz && k1 < TT) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            if (i < n) { stz[i] = zi[q]; stz[n + i] = ui[q]; }
        }
    }
    if (!do_reverse) return;

    // ---- reverse sweep ----
    T gi[EPT], di[EPT], ubar[EPT], zbar[EPT], pbar[EPT], lbbar[EPT], ubbar[EPT];
    T rho_part = T(0);
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid + q * NT;
        const bool live = i < n;
        gi[q] = live ? U.g[(size_t)b * n + i] : T(0);
        di[q] = live ? V.D[i] : T(0);
        ubar[q] = zbar[q] = pbar[q] = lbbar[q] = ubbar[q] = T(0);
        if (live && k1 < TT) {            // a later segment has been walked already: its cotangents and what it accumulated
            if (stz) { zbar[q] = stz[i]; ubar[q] = stz[n + i]; }
            pbar[q] = U.dps[(size_t)b * n + i]; lbbar[q] = U.dlbs[(size_t)b * n + i]; ubbar[q] = U.dubs[(size_t)b * n + i];
        }
    }
    const T* injb = (U.inj && U.inj_k >= 0) ? U.inj + (size_t)b * 4 * n : nullptr;
    __syncthreads();                                       // (the scratch rows above are read back by their writers only)
    for (int k = k1 - 1; k >= k0; --k) {
        T unew[EPT];
        const bool inj_here = injb != nullptr && k == U.inj_k;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            unew[q] = T(0);
            if (i < Np) {
                T val = T(0);
                if (i < n) {
                    if (inj_here) { zbar[q] += injb[(size_t)n + i]; ubar[q] += injb[(size_t)2 * n + i]; }
                    const int code = (int)MK[(size_t)k * n + i];
                    const T zt = zbar[q] - ubar[q];
                    const T wfree = code == 0 ? zt : T(0);
                    lbbar[q] += code < 0 ? zt : T(0);
                    ubbar[q] += code > 0 ? zt : T(0);
                    unew[q] = wfree + ubar[q];
                    val = -(unew[q] + (k == TT - 1 ? di[q] * gi[q] : T(0)) + (inj_here ? injb[i] : T(0)));      // rhs = -xvbar_k (its nu part is zero: nu is not an output)
                }
                v[dest[i]] = val;
            }
        }
        wg_barrier_lds();
        wg_packed_solve<T, NT>(st, packed, K, v, tmp, true);
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            if (i < n) {
                const T dxx = v[i];
                const T xk = X[(size_t)k * n + i];
                DX[(size_t)k * n + i] = dxx;
                pbar[q] += dxx;
                rho_part += dxx * (xk - Wd[(size_t)k * n + i]);
                zbar[q] = -rho * dxx + (inj_here ? injb[(size_t)3 * n + i] : T(0));
                ubar[q] = unew[q] + rho * dxx;
            } else if (i < N) DNU[(size_t)k * m + (i - n)] = v[i];
        }
        wg_barrier_lds();
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = tid + q * NT;
        if (i < n) {
            U.dps[(size_t)b * n + i] = pbar[q];
            U.dlbs[(size_t)b * n + i] = lbbar[q];
            U.dubs[(size_t)b * n + i] = ubbar[q];
            if (k1 == TT) U.dD[(size_t)b * n + i] = gi[q] * X[(size_t)(TT - 1) * n + i];
            if (stz && k0 > 0) { stz[i] = zbar[q]; stz[n + i] = ubar[q]; }
        }
    }
    const T rsum = wg_sum(rho_part, red);
    if (tid == 0) U.drho[b] = rsum;
}

// Asbar[b][r][i] = sum_k DNU[k][r] X[k][i] + NU[k][r] DX[k][i],  bsbar[b][r] = -sum_k DNU[k][r]   (gridDim = (B, slabs over m n))
template <typename T>
__global__ __launch_bounds__(256) void k_unroll_lu_eq(const UnrollLuParams<T> U, const int n, const int m) {
    const int b = blockIdx.x, TT = U.T_;
    const T* X = U.X + (size_t)b * TT * n;
    const T* DX = U.DX + (size_t)b * TT * n;
    const T* NU = U.NU + (size_t)b * TT * m;
    const T* DNU = U.DNU + (size_t)b * TT * m;
    for (int t = blockIdx.y * 256 + threadIdx.x; t < m * n; t += gridDim.y * 256) {
        const int r = t / n, i = t - r * n;
        T acc = T(0);
        for (int k = 0; k < TT; ++k) acc += DNU[(size_t)k * m + r] * X[(size_t)k * n + i] + NU[(size_t)k * m + r] * DX[(size_t)k * n + i];
        U.dAs[((size_t)b * m + r) * n + i] = acc;
    }
    if (blockIdx.y == 0)
        for (int r = threadIdx.x; r < m; r += 256) {
            T acc = T(0);
            for (int k = 0; k < TT; ++k) acc -= DNU[(size_t)k * m + r];
            U.dbs[(size_t)b * m + r] = acc;
        }
}

// Qsbar[b] = sum_k DX[b][k][:]^T X[b][k][:] in any dtype (the float32 tile kernel above, typed): 64 x 64 tile per 256 threads
template <typename T>
__global__ __launch_bounds__(256) void k_unroll_outer_any(const T* __restrict__ DXall, const T* __restrict__ Xall,
                                                          T* __restrict__ out, const int n, const int TT) {
    __shared__ T sa[16][64 + 4], sb[16][64 + 4];
    const int b = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const T* DX = DXall + (size_t)b * TT * n;
    const T* X = Xall + (size_t)b * TT * n;
    T acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = T(0);
    for (int k0 = 0; k0 < TT; k0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int kk = e >> 6, j = e & 63;
            const bool ok = k0 + kk < TT;
            sa[kk][j] = (ok && r0 + j < n) ? DX[(size_t)(k0 + kk) * n + r0 + j] : T(0);
            sb[kk][j] = (ok && c0 + j < n) ? X[(size_t)(k0 + kk) * n + c0 + j] : T(0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            T av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = sa[kk][ty * 4 + a]; bv[a] = sb[kk][tx * 4 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] += av[a] * bv[c];
        }
        __syncthreads();
    }
    T* o = out + (size_t)b * n * n;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = r0 + ty * 4 + a;
        if (r < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cc = c0 + tx * 4 + c;
                if (cc < n) o[(size_t)r * n + cc] = acc[a][c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The scaling (:160-203) behind the unrolled loop: what of its derivative touches the B x n x n tensors.
//
// The reference lets autograd tape the pre-conditioning too; of its ~25 operations four walk over Q-sized tensors
// (:163 the column maxima of |Q|, :176 Qs = D Q D, :201 ||Qs||_F, and their backward nodes; ||Qs||_F itself is rho sqrt(n) of the forward), which as eager torch ops are
// ~25 passes over 128 MB at the headline size -- 2.0 of the 4.2 ms of an unroll step.  Here each is ONE pass:
//   k_unroll_scale_colmax   cn_j = max_i |Q_ij|, the first row that attains it, how many rows do (amax's backward shares
//                           the gradient evenly between ties)
//   k_unroll_scale_grad     G := D (G + s D Q D) D in place (G = dL/dQs from the sweep, s = dL/d||Qs||_F / ||Qs||_F) and the two
//                           reductions the gradient of D needs: r_i = sum_j T_ij Q_ij d_j, c_j = sum_i T_ij d_i Q_ij (T = G + s Qs)
//   k_unroll_scale_scatter  G_ij += gcn_j sign(Q_ij) / count_j where |Q_ij| = cn_j  (the backward of :163)
//   k_unroll_scale_vectors  the n-sized rest of the chain (quantiles, beta, the equality rows, the bounds), forward and backward
// float32, any n (one column per thread and pass); D may be null (scale = False).
// ---------------------------------------------------------------------------------------------------------------------
// gridDim.y column slabs per problem; inside a workgroup G = 1024 / (padded slab width) groups of threads share the rows of a
// column (rows g, g + G, ...: eight loads in flight per thread) and merge through LDS -- first row of the maximum, tie count.
template <int LQP_ANY = 0>
__global__ __launch_bounds__(LQP_NT) void k_unroll_scale_colmax(const float* __restrict__ Qall, const int n, float* __restrict__ cn,
                                                                int* __restrict__ arg, int* __restrict__ cnt) {
    __shared__ float sbest[LQP_NT];
    __shared__ int sidx[LQP_NT], scnt[LQP_NT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Q = Qall + (size_t)b * n * n;
    const int wcols = (n + (int)gridDim.y - 1) / (int)gridDim.y;           // columns of this slab
    const int c0 = (int)blockIdx.y * wcols, c1 = c0 + wcols < n ? c0 + wcols : n;
    constexpr int RIF = 8;
    for (int cb = c0; cb < c1; cb += LQP_NT) {                               // (slabs wider than 1024 columns: several passes)
        const int wc = c1 - cb < LQP_NT ? c1 - cb : LQP_NT;
        int wp = 64;
        while (wp < wc) wp <<= 1;                                           // padded width: a power of two
        const int G = LQP_NT / wp, g = tid / wp, j = cb + (tid & (wp - 1));
        const bool live = j < c1;
        float best = -1.f;
        int bi = 0, bc = 0;
        if (live) {
            for (int i0 = g; i0 < n; i0 += RIF * G) {
                float v[RIF];
#pragma unroll
                for (int r = 0; r < RIF; ++r) { const int i = i0 + r * G; v[r] = Q[(size_t)(i < n ? i : i0) * n + j]; }
#pragma unroll
                for (int r = 0; r < RIF; ++r) {
                    const int i = i0 + r * G;
                    if (i < n) {
                        const float a = fabsf(v[r]);
                        if (a > best) { best = a; bi = i; bc = 1; }
                        else if (a == best) ++bc;
                    }
                }
            }
        }
        sbest[tid] = best; sidx[tid] = bi; scnt[tid] = bc;
        __syncthreads();
        if (g == 0 && live) {
            for (int q = 1; q < G; ++q) {
                const float ob = sbest[tid + q * wp];
                const int oi = sidx[tid + q * wp], oc = scnt[tid + q * wp];
                if (ob > best) { best = ob; bi = oi; bc = oc; }
                else if (ob == best) { bc += oc; bi = oi < bi ? oi : bi; }
            }
            cn[(size_t)b * n + j] = best;
            arg[(size_t)b * n + j] = bi;
            cnt[(size_t)b * n + j] = bc;
        }
        __syncthreads();
    }
}

// parts: (B, 1 + gridDim.y, n): slot 0 the row sums r_i (written by the slab that owns row i), slot 1 + y the column sums of slab y.
// 256 threads: wave w takes rows r0 + w, r0 + w + 4, ... of its slab TWO at a time, a lane the columns lane, lane + 64, ...
template <int LQP_ANY = 0>
__global__ __launch_bounds__(256) void k_unroll_scale_grad(const float* __restrict__ Qall, const float* __restrict__ dall,
                                                           const float* __restrict__ sall, float* __restrict__ Gall, const int n,
                                                           float* __restrict__ parts) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    float* cred = (float*)smem;                         // [4][n] column sums of the waves
    constexpr int NWV = 4, CPL = 8, RIF = 2;
    const int b = blockIdx.x, y = blockIdx.y, ny = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* Q = Qall + (size_t)b * n * n;
    float* G = Gall + (size_t)b * n * n;
    const float* d = dall ? dall + (size_t)b * n : nullptr;
    const float s = sall ? sall[b] : 0.f;
    float* P0 = parts + (size_t)b * (1 + ny) * n;
    float* Py = P0 + (size_t)(1 + y) * n;
    const int rows = (n + ny - 1) / ny, r0 = y * rows, r1 = (r0 + rows < n) ? r0 + rows : n;
    for (int c0 = 0; c0 < n; c0 += 64 * CPL) {          // (n <= 512: one pass)
        float dj[CPL], cacc[CPL];
        int jc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int j = c0 + lane + 64 * q;
            jc[q] = j < n ? j : n - 1;
            dj[q] = (j < n) ? (d ? d[j] : 1.f) : 0.f;
            cacc[q] = 0.f;
        }
        for (int i0 = r0 + w; i0 < r1; i0 += RIF * NWV) {
            float qv[RIF][CPL], gv[RIF][CPL], di[RIF];
#pragma unroll
            for (int r = 0; r < RIF; ++r) {
                const int i = (i0 + r * NWV < r1) ? i0 + r * NWV : i0;
                di[r] = d ? d[i] : 1.f;
#pragma unroll
                for (int q = 0; q < CPL; ++q) { const size_t o = (size_t)i * n + jc[q]; qv[r][q] = Q[o]; gv[r][q] = G[o]; }
            }
#pragma unroll
            for (int r = 0; r < RIF; ++r) {
                const int i = i0 + r * NWV;
                if (i < r1) {
                    float racc = 0.f;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        if (c0 + lane + 64 * q < n) {
                            const float dq = di[r] * qv[r][q];                 // d_i Q_ij
                            const float t = gv[r][q] + s * (dq * dj[q]);         // T_ij
                            G[(size_t)i * n + jc[q]] = (di[r] * t) * dj[q];
                            racc += t * (qv[r][q] * dj[q]);
                            cacc[q] += t * dq;
                        }
                    }
                    racc = wave_sum(racc);
                    if (lane == 0) { if (c0 == 0) P0[i] = racc; else P0[i] += racc; }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) { const int j = c0 + lane + 64 * q; if (j < n) cred[(size_t)w * n + j] = cacc[q]; }
        __syncthreads();
        for (int j = c0 + tid; j < n && j < c0 + 64 * CPL; j += 256) {
            float a = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) a += cred[(size_t)ww * n + j];
            Py[j] = a;
        }
        __syncthreads();
    }
}

// The n-sized part of the chain, forward (phase 0: the scaling vector only) and backward (phase 1), one workgroup per problem
// (scale = True; n <= 1024 + m <= 16: what the native unroll takes).  Forward as the setup kernel forms it (wg_scaling_vector,
// :163-175): c = column maxima (non-positive ones floored, :164-168), d0 = c^-1/2, beta = 1 - q10(d0) / q90(d0) (torch.quantile's
// linear interpolation) unless given, d = (1 - beta) d0 + beta mean(d0); ps = d p; A1 = A d, E = 1 / rowmax |A1| (floored), As = E A1,
// bs = E b (:179-190); lbs = lb / d, ubs = ub / d (:192-194).  Backward: the derivative of exactly that, node by node as autograd
// would take it (amax shares its gradient between ties, clamp passes it inside its range, the quantile to its two neighbours).
struct ScaleVecParams {
    int n, m, phase, has_box, beta_given, nparts;
    float beta_value;
    const float *cn, *p, *A, *b, *lb, *ub;                       // (B,n) (B,n) (B,m,n) (B,m) (B,n) (B,n)
    const float *gps, *gAs, *gbs, *glbs, *gubs, *gD, *parts;     // upstream: dL/d(ps, As, bs, lbs, ubs, D) and (B, nparts, n) dL/dd through Qs
    float *d_out;                                                // phase 0: (B,n)
    float *dp, *dA, *db, *dlb, *dub, *gcn;                       // phase 1
};

template <int LQP_ANY = 0>
__global__ __launch_bounds__(LQP_NT) void k_unroll_scale_vectors(const ScaleVecParams P) {
    extern __shared__ __attribute__((aligned(32))) char smem[];
    constexpr int NT = LQP_NT;
    const int b = blockIdx.x, n = P.n, m = P.m, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int N2 = 1;
    while (N2 < n) N2 <<= 1;
    float* c = (float*)smem;              // floored column maxima
    float* d0 = c + n;
    float* dd = d0 + n;
    float* gd = dd + n;
    float* sb = gd + n;                   // [N2] sort buffer
    float* red = sb + N2;                 // [NW + 8]
    float* rn = red + LQP_NW + 8;         // [m] row maxima of |A d| (floored)
    float* Er = rn + (m > 0 ? m : 1);     // [m]
    float* grn = Er + (m > 0 ? m : 1);    // [m]
    int* kr = (int*)(grn + (m > 0 ? m : 1));   // [m] tie counts
    int* sel_i = kr + (m > 0 ? m : 1);    // [4] original indices of the four order statistics
    float* sel = (float*)(sel_i + 4);     // [4] their values | [4..8) scalars
    const float* cn = P.cn + (size_t)b * n;

    // ---- forward ----
    float part = 0.f;
    int anybad = 0;
    for (int j = tid; j < n; j += NT) { const float v = cn[j]; part += v; anybad |= (v <= 0.f) ? 1 : 0; }
    const float meanc = wg_sum(part, red) / (float)n;
    anybad = __syncthreads_or(anybad);
    const float floor_c = tmax(meanc, 1e-6f);
    part = 0.f;
    for (int j = tid; j < n; j += NT) {
        float v = cn[j];
        if (v <= 0.f) v = tmax(v, floor_c);
        c[j] = v;
        v = sqrtf(1.f / v);
        d0[j] = v;
        part += v;
    }
    const float mean0 = wg_sum(part, red) / (float)n;
    float beta = P.beta_value, q10 = 0.f, q90 = 1.f, w0 = 0.f, w1 = 0.f;
    if (!P.beta_given) {
        const float pos0 = 0.10f * (float)(n - 1), pos1 = 0.90f * (float)(n - 1);
        const int lo0 = (int)floorf(pos0), hi0 = (int)ceilf(pos0), lo1 = (int)floorf(pos1), hi1 = (int)ceilf(pos1);
        for (int i = tid; i < N2; i += NT) sb[i] = i < n ? d0[i] : INFINITY;
        __syncthreads();
        for (int k = 2; k <= N2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (N2 >> 1); t += NT) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const float x0 = sb[i], x1 = sb[l];
                    if ((x0 > x1) == ((i & k) == 0)) { sb[i] = x1; sb[l] = x0; }
                }
                __syncthreads();
            }
        }
        if (tid == 0) { sel[0] = sb[lo0]; sel[1] = sb[hi0]; sel[2] = sb[lo1]; sel[3] = sb[hi1]; }
        if (tid < 4) sel_i[tid] = 0;
        __syncthreads();
        w0 = pos0 - floorf(pos0); w1 = pos1 - floorf(pos1);
        q10 = (w0 < 0.5f) ? sel[0] + w0 * (sel[1] - sel[0]) : sel[1] - (sel[1] - sel[0]) * (1.f - w0);
        q90 = (w1 < 0.5f) ? sel[2] + w1 * (sel[3] - sel[2]) : sel[3] - (sel[3] - sel[2]) * (1.f - w1);
        beta = 1.f - q10 / q90;
        if (P.phase == 1) {
            // who holds the four order statistics: the element of (stable) rank lo / hi -- only a value equal to one of them can
            for (int j = tid; j < n; j += NT) {
                const float v = d0[j];
                if (v == sel[0] || v == sel[1] || v == sel[2] || v == sel[3]) {
                    int rank = 0;
                    for (int i = 0; i < n; ++i) { const float o = d0[i]; rank += (o < v || (o == v && i < j)) ? 1 : 0; }
                    if (rank == lo0) sel_i[0] = j;
                    if (rank == hi0) sel_i[1] = j;
                    if (rank == lo1) sel_i[2] = j;
                    if (rank == hi1) sel_i[3] = j;
                }
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += NT) dd[j] = (1.f - beta) * d0[j] + beta * mean0;
    __syncthreads();
    if (P.phase == 0) {
        for (int j = tid; j < n; j += NT) P.d_out[(size_t)b * n + j] = dd[j];
        return;
    }

    // ---- backward ----
    const float* p = P.p + (size_t)b * n;
    for (int j = tid; j < n; j += NT) {
        const size_t o = (size_t)b * n + j;
        const float dj = dd[j], gp = P.gps ? P.gps[o] : 0.f;
        float g = (P.gD ? P.gD[o] : 0.f) + gp * p[j];
        for (int y = 0; y < P.nparts; ++y) g += P.parts[((size_t)b * P.nparts + y) * n + j];
        if (P.dp) P.dp[o] = dj * gp;
        if (P.has_box) {
            const float glb = P.glbs ? P.glbs[o] : 0.f, gub = P.gubs ? P.gubs[o] : 0.f;
            const float lbj = P.lb[o], ubj = P.ub[o];
            // (an infinite bound never binds: its gradient is zero and 0 * inf is taken as 0 here)
            if (lbj > -INFINITY && lbj < INFINITY) g -= glb * lbj / (dj * dj);
            if (ubj > -INFINITY && ubj < INFINITY) g -= gub * ubj / (dj * dj);
            if (P.dlb) P.dlb[o] = glb / dj;
            if (P.dub) P.dub[o] = gub / dj;
        }
        gd[j] = g;
    }
    __syncthreads();
    if (m > 0) {
        const float* A = P.A + (size_t)b * m * n;
        const float* gA = P.gAs + (size_t)b * m * n;
        // row maxima of |A d| and their tie counts: a row per wave
        for (int r = w; r < m; r += LQP_NW) {
            float mx = 0.f;
            for (int j = lane; j < n; j += 64) mx = tmax(mx, fabsf(A[(size_t)r * n + j] * dd[j]));
            mx = wave_max(mx);
            int k = 0;
            for (int j = lane; j < n; j += 64) k += (fabsf(A[(size_t)r * n + j] * dd[j]) == mx) ? 1 : 0;
            k = (int)wave_sum((float)k);                     // (small integers: exact in float)
            if (lane == 0) { rn[r] = mx; kr[r] = k; }
        }
        __syncthreads();
        float meanr = 0.f;
        int badr = 0;
        for (int r = 0; r < m; ++r) { meanr += rn[r]; badr |= rn[r] <= 0.f ? 1 : 0; }
        meanr /= (float)m;
        const float floor_r = tmax(meanr, 1e-6f);
        // E, dL/dE = sum_j gAs_rj A1_rj + gbs_r b_r, dL/d(row maximum): a row per wave
        for (int r = w; r < m; r += LQP_NW) {
            const float rv = rn[r] <= 0.f ? tmax(rn[r], floor_r) : rn[r];
            float acc = 0.f;
            for (int j = lane; j < n; j += 64) acc += gA[(size_t)r * n + j] * (A[(size_t)r * n + j] * dd[j]);
            acc = wave_sum(acc);
            if (lane == 0) {
                const float e = 1.f / rv;
                const float gbs = P.gbs ? P.gbs[(size_t)b * m + r] : 0.f;
                const float gE = acc + gbs * P.b[(size_t)b * m + r];                // (As = E A1, bs = E b)
                Er[r] = e;
                grn[r] = -gE / (rv * rv);
                if (P.db) P.db[(size_t)b * m + r] = e * gbs;
            }
        }
        __syncthreads();
        if (badr) {                                    // the floor (:182-186): floored rows pass their gradient to the mean
            float gfl = 0.f;
            for (int r = 0; r < m; ++r) gfl += rn[r] <= 0.f ? grn[r] : 0.f;
            __syncthreads();
            if (tid < m) grn[tid] = (rn[tid] <= 0.f ? 0.f : grn[tid]) + (meanr >= 1e-6f ? gfl / (float)m : 0.f);
            __syncthreads();
        }
        // dA and the share of the equality rows in dL/dd: a column per thread, rows in order
        for (int j = tid; j < n; j += NT) {
            const float dj = dd[j];
            float g = gd[j];
            for (int r = 0; r < m; ++r) {
                const float a = A[(size_t)r * n + j], a1 = a * dj;
                float ga1 = Er[r] * gA[(size_t)r * n + j];
                if (rn[r] > 0.f && fabsf(a1) == rn[r]) ga1 += (a1 > 0.f ? grn[r] : -grn[r]) / (float)kr[r];
                if (P.dA) P.dA[((size_t)b * m + r) * n + j] = ga1 * dj;
                g += ga1 * a;
            }
            gd[j] = g;
        }
        __syncthreads();
    }
    // d = (1 - beta) d0 + beta mean(d0)
    float s1 = 0.f, s2 = 0.f;
    for (int j = tid; j < n; j += NT) { s1 += gd[j]; s2 += gd[j] * d0[j]; }
    s1 = wg_sum(s1, red);
    s2 = wg_sum(s2, red);
    const float gbeta = s1 * mean0 - s2;
    for (int j = tid; j < n; j += NT) gd[j] = (1.f - beta) * gd[j] + beta * s1 / (float)n;      // now dL/dd0
    __syncthreads();
    if (!P.beta_given && tid == 0) {
        const float gq10 = -gbeta / q90, gq90 = gbeta * q10 / (q90 * q90);
        gd[sel_i[0]] += (1.f - w0) * gq10; gd[sel_i[1]] += w0 * gq10;
        gd[sel_i[2]] += (1.f - w1) * gq90; gd[sel_i[3]] += w1 * gq90;
    }
    __syncthreads();
    // d0 = c^-1/2, the floor (:164-168)
    float gfl = 0.f;
    for (int j = tid; j < n; j += NT) {
        const float gc = -0.5f * gd[j] * d0[j] / c[j];
        const bool bad = cn[j] <= 0.f;
        gfl += bad ? gc : 0.f;
        gd[j] = bad ? 0.f : gc;
    }
    if (anybad) gfl = wg_sum(gfl, red);
    for (int j = tid; j < n; j += NT) P.gcn[(size_t)b * n + j] = gd[j] + ((anybad && meanc >= 1e-6f) ? gfl / (float)n : 0.f);
}
__host__ __device__ inline int unroll_scale_vectors_lds_bytes(int n, int m) {
    int N2 = 1;
    while (N2 < n) N2 <<= 1;
    const int mm = m > 0 ? m : 1;
    return (4 * n + N2 + LQP_NW + 8 + 3 * mm) * 4 + (mm + 4) * 4 + 8 * 4 + 64;
}

template <int LQP_ANY = 0>
__global__ __launch_bounds__(LQP_NT) void k_unroll_scale_scatter(const float* __restrict__ Qall, const float* __restrict__ cn,
                                                                 const int* __restrict__ arg, const int* __restrict__ cnt,
                                                                 const float* __restrict__ gcn, float* __restrict__ Gall, const int n) {
    const int b = blockIdx.x;
    const float* Q = Qall + (size_t)b * n * n;
    float* G = Gall + (size_t)b * n * n;
    for (int j = threadIdx.x; j < n; j += LQP_NT) {
        const size_t o = (size_t)b * n + j;
        const float g = gcn[o], c = cn[o];
        const int k = cnt[o];
        if (g == 0.f || !(c > 0.f)) continue;              // (sign(0) = 0: a zero column passes nothing on)
        if (k == 1) {
            const size_t e = (size_t)arg[o] * n + j;
            G[e] += Q[e] > 0.f ? g : -g;
        } else {
            const float gk = g / (float)k;
            for (int i = 0; i < n; ++i) {
                const size_t e = (size_t)i * n + j;
                const float q = Q[e];
                if (fabsf(q) == c) G[e] += q > 0.f ? gk : -gk;
            }
        }
    }
}

}  // namespace lqp
