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

}  // namespace lqp
