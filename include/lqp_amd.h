/*
 * lqp_amd.h -- C ABI of the MI355X-native batched box-QP ADMM layer.
 *
 * The reference (ipo-lab/lqp_py) has no FFI: its boundary is a Python API on
 * top of torch.linalg.  Each entry point below replaces the reference call
 * sites listed next to it; the Python shim in lqp_py_amd/ binds them with
 * ctypes and keeps the reference's signatures.
 *
 * Conventions
 *   - every function returns an int status (LQP_OK == 0), never throws,
 *     never allocates: scratch comes from the caller (`workspace`), sized by
 *     the matching *_workspace_bytes query;
 *   - all tensor pointers are DEVICE pointers to dense, batch-first,
 *     row-major (C-contiguous) arrays, exactly the layout of a contiguous
 *     torch tensor of the stated shape;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - dtype: LQP_F32 or LQP_F64 (the arithmetic type of every tensor).
 */
#ifndef LQP_AMD_H
#define LQP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LQP_ABI_VERSION 13

enum { LQP_F32 = 0, LQP_F64 = 1 };

enum {
    LQP_OK = 0,
    LQP_ERR_INVALID = 1,      /* bad argument (null pointer, negative size, unknown dtype) */
    LQP_ERR_WORKSPACE = 2,    /* workspace smaller than *_workspace_bytes               */
    LQP_ERR_SINGULAR = 3,     /* exactly-zero pivot; batch index in stats / info         */
    LQP_ERR_HIP = 4,          /* a HIP runtime call failed                               */
    LQP_ERR_TIMEOUT = 5,      /* in-kernel grid barrier gave up (bounded spin)           */
    LQP_ERR_UNSUPPORTED = 6,  /* size outside what the kernels are built for: n + m > 4096 (float32) / 2048 (float64) */
    LQP_ERR_NOT_SPD = 7       /* lqp_boxqp_forward_finish only: the matrix left the symmetric x-update (not symmetric / Qs + rho I not
                                 positive definite in f32): repeat lqp_boxqp_forward with ctrl.linsolve = 1                     */
};

/* Resolved solver controls.  Key-name resolution of the reference's control
 * dict (lqp_py/solve_box_qp_admm_torch.py:133-154, lqp_py/control.py:1-24) and
 * the dict side effect (:37-38) stay in the Python shim; this struct carries
 * the numbers the loop actually uses. */
typedef struct lqp_boxqp_ctrl {
    int32_t max_iters;
    int32_t check_solved;            /* convergence test every this many iterations   */
    int32_t adaptive_rho;            /* 0/1                                            */
    int32_t adaptive_rho_iter;       /* already rounded to a multiple of check_solved */
    int32_t adaptive_rho_max_iter;
    int32_t scale;                   /* 0/1 auto-scaling (:160-197)                    */
    int32_t any_lb;                  /* not read since ABI 5: whether max(lb) > -inf / min(ub) < +inf over the batch  */
    int32_t any_ub;                  /* (:129-130) is found on the DEVICE (the clamps always run -- an infinite bound
                                        is an exact no-op -- and the answer comes back in lqp_boxqp_stats.any_lb/ub
                                        or, for un-synchronised calls, in status words 12/13).  What the caller must
                                        still decide itself is the reference's rho = 0 shortcut for a batch without any
                                        finite bound (:157-158): pass rho_mode 1, rho_value 0 for it.              */
    int32_t rho_mode;                /* 0 auto (||Q||_F/sqrt(n), :200-203), 1 scalar rho_value,
                                        2 per-problem array `rho_in` (B values)        */
    int32_t beta_mode;               /* 0 auto (quantile rule :171-174), 1 scalar beta_value, 2 per-problem array
                                        `beta_in` (B values; the reference accepts a (B,1) tensor, :171-175)  */
    int32_t launch_mode;             /* 0 auto, 1 one launch per check segment,
                                        2 persistent loop kernel with in-kernel grid barrier */
    int32_t reserved;                /* 1: do not synchronise with the host (needs persistent launches and a
                                        short adaptive-rho schedule, else ignored): the whole schedule is
                                        enqueued speculatively, stats come back as -1, and the caller fetches
                                        status / info later (lqp_boxqp_forward_layout)
                                        2: split synchronous call (needs host_report): the schedule is enqueued
                                        and the call returns with stats.mode_used = 4; the caller does its own
                                        host work while the GPU runs and then calls lqp_boxqp_forward_finish,
                                        which waits and reports like a synchronous call would have.  Where the
                                        library cannot enqueue the whole schedule (segmented launches, a check
                                        hook) the call simply waits itself (mode_used 1 / 2)                  */
    int32_t linsolve;                /* x-update linear algebra: 0 auto, 1 pivoted LU of the KKT matrix (the
                                        reference's, :214-215/:267), 2 symmetric inverse of Qs + rho I with a
                                        rank-m equality correction (f32, n <= 1024, m <= 16, rho > 0; anything
                                        else, or a matrix that is not positive definite, runs on LU)            */
    int32_t reserved2;               /* bit 0: leave the complete factor (equality correction included) in the workspace for
                                        lqp_boxqp_unroll_backward; bit 1 (ABI 11): share nothing between workgroups -- one
                                        workgroup per matrix everywhere, host-launched check segments: what the library falls
                                        back to by itself when a kernel that waits for a partner workgroup gives up (CUs held
                                        by another stream / process), and what a caller of lqp_boxqp_forward_finish that got
                                        LQP_ERR_TIMEOUT passes when it repeats the forward; bit 2 (ABI 12): keep a
                                        trace of the convergence checks -- the largest primal and dual error over the batch at
                                        every check, what the reference prints under verbose=True (:289-294) -- for
                                        lqp_boxqp_check_trace                                                            */
    double eps_abs;
    double eps_rel;
    double rho_value;
    double rho_min;
    double rho_max;
    double adaptive_rho_tol;
    double adaptive_rho_threshold;
    double beta_value;
    const void* beta_in;             /* beta_mode == 2: B values of the tensors' dtype (device pointer), else NULL */
    /* Strict global stopping across batch shards (one process per GPU): when set, the solve runs one
     * launch per check segment and, right after the launch that holds check number `check_index`, calls
     *     check_hook(check_hook_user, stream, counters, check_index)
     * on the host with the DEVICE address of that check's FOUR uint32 words {problems not yet optimal, arrivals
     * (unused in this mode), problems that want a rho update, problems whose residual ratio triggers one} (:310-312,
     * :244-246).  The hook must enqueue, ordered on `stream`, an in-place SUM all-reduce of all four words over the
     * ranks (RCCL); every consumer of the counters (stop test, adaptive-rho decision) is enqueued after it, so all
     * ranks take the single-process decisions and report the single-process iteration count.
     * check_index == -1: the same reduction over the four-word failure vote of a factorisation {ranks whose matrix
     * left the symmetric x-update, ranks with an exactly singular KKT matrix, 0, 0}: every rank then repeats the solve
     * on the LU path, or fails, together -- the sequence of collectives stays the same on all ranks.
     * Non-zero return aborts the solve (LQP_ERR_HIP).  NULL: decisions are per call (per shard).                  */
    int (*check_hook)(void* user, void* stream, void* counters_dev, int check_index);
    void* check_hook_user;
    const void* bound_flags_in;      /* optional: two int32 on the device {any_lb, any_ub} of a LARGER batch this call
                                        holds one shard of (one all-reduce MAX over the ranks); they are
                                        OR-ed into the answer reported back.  NULL: this call is the whole batch.    */
    void* host_report;               /* optional: PINNED host memory the device can write (hipHostMalloc /
                                        torch pin_memory), 16 + 2 B int32.  The forward's last kernel then leaves
                                        there, with no device-to-host copy: [0..16) the status block (see
                                        lqp_boxqp_forward_layout), [16..16+B) the info word of every problem,
                                        [16+B..16+2B) per-problem flag bits {1: a lower bound is finite, 2: an upper
                                        bound is finite, 4: its workgroup saw a barrier timeout, 8: its matrix left the
                                        symmetric x-update}.  Valid once the stream has passed the call (an event):
                                        ctrl.reserved = 1 callers read it then.  A synchronous call sets every
                                        word to -1 before its first launch and POLLS them (all stored words are >= 0)
                                        instead of synchronising the stream: it returns about a microsecond after the
                                        last kernel's stores, not after an interrupt-driven stream wait.            */
} lqp_boxqp_ctrl;

/* Host-side bookkeeping returned by the forward solve. */
typedef struct lqp_boxqp_stats {
    int32_t iters;          /* == the reference's sol["iter"]                          */
    int32_t n_factor;       /* LU factorisations (1 + adaptive-rho refactors)          */
    int32_t n_solve;        /* x-updates == iters + 1                                  */
    int32_t n_check;        /* convergence checks performed                            */
    int32_t rho_updated;    /* 1 if adaptive rho changed rho at least once             */
    int32_t fail_index;     /* batch index of the first singular problem, or -1        */
    int32_t n_launch;       /* kernel launches issued                                  */
    int32_t mode_used;      /* 1 segmented, 2 persistent, 3 persistent without host sync, 4 enqueued, lqp_boxqp_forward_finish pending */
    int32_t linsolve_used;  /* 1 pivoted LU, 2 symmetric inverse (what linsolve 0 / a fallback resolved to) */
    int32_t factor_launches; /* kernel launches per (re)factorisation: 1, 2 (LU + pack) or Ks + 2 when a small batch
                              * shares each matrix between two workgroups (one launch per pivot step) */
    int32_t loop_workgroups; /* workgroups per QP in the first (hot) loop launch: 1, 2 (2 B <= CUs) or 4 (4 B <= CUs) when a small batch
                              * on the symmetric path splits every product between two CUs */
    int32_t any_lb;          /* 1: some lower bound of the batch is finite (:129), 0: none, -1: not known on the host */
    int32_t any_ub;          /* the same for the upper bounds (:130)                                                 */
} lqp_boxqp_stats;

int lqp_abi_version(void);
const char* lqp_status_string(int status);

/* ---- measurement hooks (not part of the reference surface) ----------------
 * When enabled, every kernel launch issued by this library is bracketed by a
 * pair of HIP events recorded on the launch stream; lqp_profile_get waits for
 * them and returns, per kernel class, the summed device time in ms and the
 * launch count since the last reset.  bench.py uses this for the roofline. */
void lqp_profile_enable(int on);
void lqp_profile_reset(void);
int lqp_profile_classes(void);
const char* lqp_profile_class_name(int cls);
int lqp_profile_get(double* total_ms, long long* launches, int n);
/* debug: device buffer of 4 uint64 per problem; every LU launch then writes its shader-clock
 * cycles spent in (panel, swaps+U12, trailing update, total).  NULL switches it off. */
void lqp_debug_set_lu_counters(void* device_buf);
/* test aid: occupies `blocks` workgroups (512 threads, `lds_bytes` of LDS each) for `usec` microseconds on `stream`:
 * the co-residency tests run the two-workgroup schedules of the forward solve beside it. */
int lqp_debug_spin(void* stream, int blocks, int usec, int lds_bytes);
/* test aid: out_dev[b] (int32, device) := the XCD (0..7, HW_REG_XCC_ID) workgroup b of a `blocks`-workgroup launch runs on --
 * the question the workgroups that share a matrix ask at every launch before they choose their exchange protocol.        */
int lqp_debug_xcd(void* stream, int blocks, void* out_dev);
/* test access to the dense tier's first kernel (csrc/lqp_dense.hpp): X (B, N, N) = M^-1 from a packed factor
 * (`packed`: the buffer lqp_lu_pack filled).  Any N whose column tile fits the LDS: float32 to 2048 (above 576 rows on 16-column
 * tiles), float64 to about 1100; LQP_ERR_UNSUPPORTED beyond.  No reference counterpart. */
int lqp_debug_lu_inverse(void* stream, int dtype, int B, int N, const void* packed, void* X_out);

/* ---- forward ADMM solve ------------------------------------------------
 * Replaces torch_solve_box_qp (lqp_py/solve_box_qp_admm_torch.py:108-333):
 * scaling :160-197, rho :199-203, KKT assembly + LU :205-215, the hot loop
 * :235-313 (lu_solve :267, clamp :271-276, residuals :279-282, check
 * :285-313, adaptive-rho refactor :237-256) and unscale/duals :315-327.
 * Shapes: Q (B,n,n) p (B,n,1) A (B,m,n)|NULL b (B,m,1)|NULL lb,ub (B,n,1);
 * outputs x,z,u (B,n,1) lams (B,2n,1) nus (B,m,1)|NULL rho_out (B) (always
 * written, one value per problem).  rho_in: B values when rho_mode == 2.  */
size_t lqp_boxqp_forward_workspace_bytes(int dtype, int B, int n, int m);
/* Where, inside the workspace, the device-side status block (16 int32: [0] done, [1] final iteration,
 * [3] adaptive-rho refactorisations, [4] rho updated, [5] grid-barrier timeout, [7] linsolve 2 met a matrix
 * that is not positive definite: results invalid, repeat with linsolve 1, [12] / [13] some lower / upper bound of the
 * batch is finite) and the per-problem LU
 * info array (B int32, non-zero = exactly singular) live -- for callers that skipped the host sync
 * (ctrl.reserved = 1) and fetch them asynchronously. */
int lqp_boxqp_forward_layout(int dtype, int B, int n, int m, size_t* status_offset, size_t* status_bytes,
                             size_t* info_offset, size_t* info_bytes);
int lqp_boxqp_forward(void* stream, int dtype, int B, int n, int m,
                      const void* Q, const void* p, const void* A, const void* b,
                      const void* lb, const void* ub,
                      const lqp_boxqp_ctrl* ctrl, const void* rho_in,
                      void* x, void* z, void* u, void* lams, void* nus, void* rho_out,
                      lqp_boxqp_stats* stats,
                      void* workspace, size_t workspace_bytes);

/* Second half of a split synchronous forward (ctrl.reserved = 2, stats.mode_used == 4 on return): waits until the
 * forward's last kernel has stored its report into `host_report` -- the pinned words are POLLED, the stream is not
 * synchronised: the outputs are stream-ordered like those of any asynchronous operator -- and completes `stats`
 * (the same struct the forward call filled).  max_iters / check_solved: the values of the ctrl struct of that call.
 * Returns what the synchronous call would have: LQP_OK, LQP_ERR_SINGULAR (stats.fail_index), LQP_ERR_TIMEOUT, or
 * LQP_ERR_NOT_SPD: repeat the forward with ctrl.linsolve = 1 (a one-call synchronous forward does that by itself). */
int lqp_boxqp_forward_finish(void* stream, int B, int max_iters, int check_solved, const void* host_report,
                             lqp_boxqp_stats* stats);

/* ---- unroll=True: backward through the unrolled ADMM loop ------------------------------------------------
 * The reference's `unroll` mode (lqp_py/solve_box_qp_admm_torch.py:14-15, 216-219, 255-256, 264-265) lets autograd tape
 * every iteration, each x-update being TorchLULayer (lqp_py/lu_layer.py:25-58: backward dx = lu_solve(LU, P, -g),
 * dA = dx x^T, db = -dx).  With a constant factor that tape is ONE reverse recurrence: this entry replays the `iters + 1`
 * x-updates of the float32 forward that used `fwd_workspace` (it must have run the symmetric x-update WITHOUT an
 * adaptive-rho refactorisation, with ctrl.reserved2 bit 0 set so that its factor is complete in the workspace; nothing
 * else may have used that workspace since) and walks back through them.  Outputs are the gradients w.r.t. the SCALED
 * problem the loop ran on -- dQs (B,n,n; NULL = skip), dps (B,n), dAs (B,m,n), dbs (B,m), dlbs, dubs (B,n), drho (B) --
 * and dD (B,n) = dl_dx * x_scaled (the returned x is D x); the caller chains them through the scaling (:160-203).
 * dl_dx (B,n): gradient w.r.t. the returned solution.  float32 only.                                            */
size_t lqp_boxqp_unroll_backward_workspace_bytes(int B, int n, int m, int iters);
int lqp_boxqp_unroll_backward(void* stream, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes,
                              int iters, const void* dl_dx, void* dQs, void* dps, void* dAs, void* dbs, void* dlbs,
                              void* dubs, void* drho, void* dD, void* scratch, size_t scratch_bytes);

/* ABI 13: the same reverse recurrence for a forward that ran the PIVOTED LU of the KKT matrix (stats.linsolve_used == 1: float64,
 * more than 16 equality rows, a non-symmetric Q, ctrl.linsolve = 1) without a refactorisation -- the tape's node is
 * TorchLULayer as it stands (lqp_py/lu_layer.py:25-58: forward lu_solve with the cached factor, backward dxv = lu_solve(LU, P, -g)
 * with the SAME factor, dM = dxv xv^T, drhs = -dxv; xv = [x; nu]): `iters + 1` cached solves to replay the loop, as many to walk it
 * back, on the packed factor the forward left in `fwd_workspace`.  Same outputs as lqp_boxqp_unroll_backward, in `dtype`;
 * n + m up to what the forward takes.  Replaces the eager tape of lqp_py/solve_box_qp_admm_torch.py:235-313 under unroll=True. */
size_t lqp_boxqp_unroll_backward_lu_workspace_bytes(int dtype, int B, int n, int m, int iters);
int lqp_boxqp_unroll_backward_lu(void* stream, int dtype, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes,
                                 int iters, const void* dl_dx, void* dQs, void* dps, void* dAs, void* dbs, void* dlbs, void* dubs,
                                 void* drho, void* dD, void* scratch, size_t scratch_bytes);

/* ABI 13: a tape whose factor CHANGES along it -- a solve in which rho was adapted (solve_box_qp_admm_torch.py:237-256 inside the
 * tape).  The caller walks it epoch by epoch: for every epoch the packed factor of ITS KKT matrix (`packed_buf`: what lqp_lu_pack
 * filled; NULL: the forward's) and its rho (B values; NULL: the forward's), the x-updates [k0, k1) of the `iters + 1` recorded ones.
 * mode bit 0: replay them from `state` = (z, u) (B,2,n; read when k0 > 0, written when k1 < iters + 1) into the scratch rows;
 * mode bit 1: walk them back -- `state` = the cotangents of (z, u) at k1 (read when k1 < iters + 1, written when k0 > 0), dps /
 * dlbs / dubs accumulate over the segments, drho is THIS segment's, dD is written by the last one (k1 == iters + 1); `inj` (B,4,n):
 * cotangents the caller adds at x-update `inj_k` to x_c | z_{c+1} | u_{c+1} | z_c -- what the gradient of the NEXT epoch's rho sends
 * into the iterates of the check it was computed from.  z_rows / u_rows / x_rows (may be NULL): where the scratch keeps
 * z_{k+1}, u_{k+1}, x_k as (B, iters + 1, n) (k1 <= k0: only this query).  lqp_boxqp_unroll_tape_finish: dQs, dAs, dbs over the whole
 * tape once every segment has been walked.  The tape's nodes are TorchLULayer's (lu_layer.py:25-58), as in
 * lqp_boxqp_unroll_backward_lu. */
size_t lqp_boxqp_unroll_tape_workspace_bytes(int dtype, int B, int n, int m, int iters);
int lqp_boxqp_unroll_tape_segment(void* stream, int dtype, int B, int n, int m, const void* fwd_workspace, size_t fwd_workspace_bytes,
                                  int iters, int k0, int k1, int mode, const void* packed_buf, const void* rho, void* state, int inj_k,
                                  const void* inj, const void* dl_dx, void* dps, void* dlbs, void* dubs, void* drho, void* dD,
                                  void* scratch, size_t scratch_bytes, void** z_rows, void** u_rows, void** x_rows);
int lqp_boxqp_unroll_tape_finish(void* stream, int dtype, int B, int n, int m, int iters, void* dQs, void* dAs, void* dbs, void* scratch,
                                 size_t scratch_bytes);

/* The scaling (solve_box_qp_admm_torch.py:160-203) behind the unrolled loop (ABI 10): what of its derivative walks over the
 * (B,n,n) tensors, one pass each; the reference lets autograd tape these as torch ops (:163 column maxima of |Q| --
 * torch.linalg.norm(ord=inf, dim=1) --, :176 Qs = D Q D, :201 ||Qs||_F and their backward nodes: ~25 passes over Q-sized
 * tensors), and the n-sized rest of the chain (quantiles / beta :169-175, equality rows :179-190, bounds :192-194) in one
 * kernel.  float32; Q (B,n,n) row-major; d (B,n) the scaling vector or NULL (scale = False).
 *   colmax   colmax_j = max_i |Q_ij| (B,n) float, argmax (B,n) int32 = the first row attaining it, count (B,n) int32 = how many do
 *   grad     G (B,n,n) in: dL/dQs (lqp_boxqp_unroll_backward's dQs); out: D (G + s D Q D) D with s (B) = dL/d||Qs||_F / ||Qs||_F
 *            (||Qs||_F = rho sqrt(n) of the forward where :201-203 did not clamp, else the term is zero) or NULL = 0;
 *            parts (B, 1 + slabs, n): [0] r_i = sum_j T_ij Q_ij d_j, [1 + y] the share of row slab y in c_j = sum_i T_ij d_i Q_ij,
 *            T = G + s D Q D: dL/dd through Qs is parts.sum(1); slabs = lqp_unroll_scale_grad_slabs(B, n)
 *   scatter  G_ij += g_colmax_j sign(Q_ij) / count_j wherever |Q_ij| = colmax_j (amax shares its gradient between ties)   */
int lqp_unroll_scale_colmax(void* stream, int B, int n, const void* Q, void* colmax, void* argmax, void* count);
int lqp_unroll_scale_grad_slabs(int B, int n);
int lqp_unroll_scale_grad(void* stream, int B, int n, const void* Q, const void* d, const void* s, void* G, void* parts, int slabs);
/*   vectors  the n-sized rest of the chain for scale = True, one workgroup per problem.  phase 0: d_out (B,n) = the scaling vector from
 *            colmax (floor :164-168, d0 = colmax^-1/2, beta = 1 - q10(d0) / q90(d0) with torch.quantile's interpolation unless
 *            beta_given, d = (1 - beta) d0 + beta mean(d0), :169-175).  phase 1: its backward and that of ps = d p, As = E (A d),
 *            bs = E b (E = 1 / row maxima of |A d|, floored, :179-190), lbs = lb / d, ubs = ub / d (has_box, :192-194): from the
 *            upstream gradients g_ps (B,n), g_As (B,m,n), g_bs (B,m), g_lbs, g_ubs, g_D (B,n) (any may be NULL = 0) and `parts`
 *            (B,nparts,n; summed over nparts: dL/dd through Qs, from `grad`) to dp, dA, db, dlb, dub (NULL = skip) and g_colmax (B,n),
 *            node by node as autograd takes it (amax shares its gradient between ties, the quantile passes it to its two
 *            neighbours); an infinite bound contributes nothing (0 * inf taken as 0).  n <= ~8000 (LDS), float32.            */
int lqp_unroll_scale_vectors(void* stream, int B, int n, int m, int phase, int has_box, int beta_given, double beta_value,
                             const void* colmax, const void* p, const void* A, const void* b, const void* lb, const void* ub,
                             const void* g_ps, const void* g_As, const void* g_bs, const void* g_lbs, const void* g_ubs, const void* g_D,
                             const void* parts, int nparts, void* d_out, void* dp, void* dA, void* db, void* dlb, void* dub,
                             void* g_colmax);
int lqp_unroll_scale_scatter(void* stream, int B, int n, const void* Q, const void* colmax, const void* argmax, const void* count,
                             const void* g_colmax, void* G);

/* Primal / dual error (inf-norms of D r and D s, :287-288) of the LAST convergence check of the forward that
 * used `workspace`, one value per problem -- the two numbers the reference's NumPy solver returns next to the
 * solution (lqp_py/solve_box_qp_admm.py:265-267).  Either output may be NULL.                              */
int lqp_boxqp_last_residuals(void* stream, int dtype, int B, int n, int m,
                             const void* workspace, size_t workspace_bytes,
                             void* primal_out, void* dual_out);

/* ABI 12: the trace of a forward solve that ran with ctrl.reserved2 bit 2 -- for check c = 0 .. n_checks - 1 (held at
 * iteration c * check_solved) trace_out[2 c] = max over the batch of ||D r||_inf, trace_out[2 c + 1] = max of ||D s||_inf,
 * float32 on the DEVICE (2 * n_checks values; at most 2048 checks are kept).  Replaces the prints of
 * solve_box_qp_admm_torch.py:289-294 (verbose=True): the loop runs on the device without the host, so the lines are
 * printed after the solve from this trace.  Enqueued on `stream`.                                                        */
int lqp_boxqp_check_trace(void* stream, int dtype, int B, int n, int m,
                          const void* workspace, size_t workspace_bytes, int n_checks, void* trace_out);

/* ---- fixed-point implicit backward --------------------------------------
 * Replaces torch_solve_box_qp_grad (solve_box_qp_admm_torch.py:349-432):
 * active-set mask :360-365, non-symmetric system :378-392, linalg.solve
 * :393, gradient epilogue :396-430.  rho_mode 1 = scalar rho_value, 2 =
 * per-problem rho_in.  Any of dQ (B,n,n), dp (B,n,1), dA (B,m,n), db (B,m,1),
 * dlb, dub (B,n,1) may be NULL = skip.  fail_index NULL = do not wait for the
 * GPU (errors stay in the workspace's info array).  linsolve: 0/1 pivoted LU of
 * the system like torch.linalg.solve (:393); 2 = Q is known to be symmetric
 * (the forward ran linsolve 2 on it): blocked Cholesky of the free-set block
 * Q_FF + the Schur complement of the equality rows (f32, n <= 512, m <= 16;
 * otherwise, or when Q_FF is not positive definite, LU).                     */
size_t lqp_boxqp_backward_fp_workspace_bytes(int dtype, int B, int n, int m);
/* The part of the Cholesky form that does not need the cotangent -- the free set (:360-365), Q_FF and its factorisation --
 * enqueued ahead of it, e.g. right behind the forward whose x, u it reads (stream order): a caller that waits for its
 * forward (the reference's semantics) then has the factorisation running while its host code travels from `forward` to
 * `backward`.  lqp_boxqp_backward_fp called afterwards on the SAME workspace (nothing else may have used it in between)
 * with linsolve = 2 | LQP_BWD_PREFACTORED only gathers dl_dz, solves and writes the gradients; a factorisation that
 * failed still ends in that call's pivoted-LU retry.  ABI 11: the LU form (float64, linsolve 0 / 1, sizes without a Cholesky
 * form) has the same two phases -- free set, reduced system, pivoted LU and packed factor ahead; gather, solve, refinement and
 * epilogue behind.  LQP_ERR_UNSUPPORTED: this form has no phases (LQP_BWD_FULL builds) -- nothing was enqueued, call
 * lqp_boxqp_backward_fp without the flag.
 * host_report (optional, pinned host memory, B ints; ABI 11): the factorisation is the only step of the Cholesky form that
 * sends a caller to the LU retry, so its info words are final when this call's last kernel ends -- it stores them there
 * (set to -1 by this call before its first launch).  lqp_boxqp_backward_fp given the SAME buffer and
 * linsolve = 2 | LQP_BWD_PREFACTORED | LQP_BWD_REPORTED neither resets nor stores them again: with fail_index it waits for
 * those words only, i.e. it returns as soon as the factorisation is known to have succeeded, while its own solves and the
 * gradient epilogue still run (stream-ordered results, like every other output of this library).  Without the prefactor call
 * lqp_boxqp_backward_fp's Cholesky form reports at the same point: right behind its factorisation.                       */
#define LQP_BWD_PREFACTORED 0x100
#define LQP_BWD_REPORTED 0x200
int lqp_boxqp_backward_fp_prefactor(void* stream, int dtype, int B, int n, int m,
                                    const void* x, const void* u,
                                    const void* Q, const void* A, const void* lb, const void* ub,
                                    void* workspace, size_t workspace_bytes, int linsolve, void* host_report);
int lqp_boxqp_backward_fp(void* stream, int dtype, int B, int n, int m,
                          const void* dl_dz, const void* x, const void* u,
                          const void* lams, const void* nus,
                          const void* Q, const void* A, const void* lb, const void* ub,
                          int rho_mode, double rho_value, const void* rho_in,
                          void* dQ, void* dp, void* dA, void* db, void* dlb, void* dub,
                          int32_t* fail_index,
                          void* workspace, size_t workspace_bytes,
                          int linsolve,
                          void* host_report);   /* NULL, or pinned host memory of B int32: the last kernel stores the
                                                   per-problem info words there (read instead of a copy when
                                                   fail_index is given, left for the caller when it is NULL)    */

/* ---- KKT-system backward (control['backward'] = 'kkt') -------------------
 * Replaces torch_solve_box_qp_grad_kkt (solve_box_qp_admm_torch.py:435-584) for a batch with finite lower AND upper
 * bounds somewhere (any_lb and any_ub, :439-441).  The reference solves the (3n+m) system
 *     [[Q, G^T diag(lam), A^T], [G, -diag(s), 0], [A, 0, 0]] [dx; dlam; dnu] = [-dl_dz; 0; 0],   G = [-I; I],
 * s = clamp(h - G x, 1e-8), lam = clamp(lams, 1e-8) (:450-452).  Eliminating dlam = diag(1/s) G dx leaves
 *     [[Q + diag(w), A^T], [A, 0]] [dx; dnu] = [-dl_dz; 0],   w = lam_lo / s_lo + lam_hi / s_hi,
 * the same solution on the kernels of the fixed-point backward: every variable free, w in place of its 1e-8
 * regulariser (linsolve 2: blocked Cholesky when Q is known to be symmetric, else pivoted LU), and an epilogue that
 * forms dQ, dp, dA, db (:527-562) and dlb = -dl_dh[:n], dub = dl_dh[n:], dl_dh = -lam dlam (:544, :573-575).
 * Arguments as lqp_boxqp_backward_fp without u / rho.                                                          */
int lqp_boxqp_backward_kkt(void* stream, int dtype, int B, int n, int m, const void* dl_dz, const void* x,
                           const void* lams, const void* nus, const void* Q, const void* A, const void* lb, const void* ub,
                           void* dQ, void* dp, void* dA, void* db, void* dlb, void* dub, int32_t* fail_index,
                           void* workspace, size_t workspace_bytes, int linsolve, void* host_report);

/* ---- batched LU (partial pivoting) and cached LU solve ------------------
 * Replace torch.linalg.lu_factor / lu_solve as used by lqp_py/lu_layer.py:
 * 10,31,33,52 and solve_box_qp_admm_torch.py:215,254,267.  M (B,N,N) is
 * overwritten by the packed LU (LAPACK layout), piv (B,N) int32 1-based,
 * info (B) int32: 0 or the 1-based index of the first zero pivot.          */
size_t lqp_lu_factor_workspace_bytes(int dtype, int B, int N);
int lqp_lu_factor_batched(void* stream, int dtype, int B, int N,
                          void* M_inout, int32_t* piv_out, int32_t* info_out,
                          void* workspace, size_t workspace_bytes);

/* rhs (B,N,k) overwritten by the solution.  The factor is first re-laid
 * out into solve-ordered 64x64 panels inside `workspace`.                   */
size_t lqp_lu_solve_workspace_bytes(int dtype, int B, int N);
int lqp_lu_solve_batched(void* stream, int dtype, int B, int N, int k,
                         const void* LU, const int32_t* piv, void* rhs_inout,
                         void* workspace, size_t workspace_bytes);

/* Two-step form for a factor that is reused many times (TorchLU in unroll
 * mode): pack once into `packed` (size lqp_lu_packed_bytes), solve often.   */
size_t lqp_lu_packed_bytes(int dtype, int B, int N);
int lqp_lu_pack(void* stream, int dtype, int B, int N,
                const void* LU, const int32_t* piv, void* packed);
int lqp_lu_solve_packed(void* stream, int dtype, int B, int N, int k,
                        const void* packed, void* rhs_inout);

/* ---- batched SPD inverse (f32, n <= 1024) ---------------------------------
 * Building block of linsolve 2 (the x-update x = H w + c replaces the cached LU solve of
 * solve_box_qp_admm_torch.py:267): Kinv_out (B,n,n) = K_in^-1 for symmetric positive definite K_in (B,n,n);
 * info (B) int32: 0, or 1 + index of the first non-positive pivot (K_in not positive definite).           */
size_t lqp_spd_inverse_workspace_bytes(int dtype, int B, int n);
int lqp_spd_inverse_batched(void* stream, int dtype, int B, int n, const void* K_in, void* Kinv_out,
                            int32_t* info_out, void* workspace, size_t workspace_bytes);

/* ---- equality-constrained / unconstrained QP = one KKT solve ------------
 * Replaces torch_solve_qp_eqcon (lqp_py/solve_qp_eqcon_torch.py:6-34, KKT
 * block from lqp_py/utils.py:23-32) and, with m == 0, torch_solve_qp_uncon
 * (lqp_py/solve_qp_uncon_torch.py:4-15): [[Q,A^T],[A,0]] [x;nu] = [-p;b].   */
size_t lqp_kkt_solve_workspace_bytes(int dtype, int B, int n, int m);
int lqp_kkt_solve(void* stream, int dtype, int B, int n, int m,
                  const void* Q, const void* p, const void* A, const void* b,
                  void* x, void* nus, int32_t* fail_index,
                  void* workspace, size_t workspace_bytes);

/* Symmetric rank-2 epilogue shared by the eq-con / uncon gradients
 * (solve_qp_eqcon_torch.py:57-66, solve_qp_uncon_torch.py:29-33):
 * dQ = 0.5 (dx x^T + x dx^T), dA = dnu x^T + nus dx^T (dA skipped if m==0). */
int lqp_qp_outer_grads(void* stream, int dtype, int B, int n, int m,
                       const void* dx, const void* x, const void* dnu, const void* nus,
                       void* dQ, void* dA);

#ifdef __cplusplus
}
#endif
#endif /* LQP_AMD_H */
