"""``unroll=True``: differentiate THROUGH the ADMM loop (reference: lqp_py/solve_box_qp_admm_torch.py:14-15 routes the
module here, :216-219/:255-256/:264-265 swap the plain LU solve for the ``TorchLU`` layer so that autograd tapes every
iteration).

Native path (one factor for the whole solve, i.e. no adaptive-rho refactorisation): the forward is the ordinary persistent HIP
solve; the backward is ONE reverse sweep over the recorded iterations in the HIP library (csrc/lqp_unroll.hpp): float32 on the
symmetric x-update (the benchmark case) ``lqp_boxqp_unroll_backward`` -- per iteration one product with the cached inverse --,
the pivoted-LU x-update (float64, m > 16, non-symmetric Q, ``linsolve='lu'``) ``lqp_boxqp_unroll_backward_lu`` -- per iteration the
two cached triangular solves of ``TorchLULayer`` --: no taped node, no per-iteration torch op, no host sync.  The kernel differentiates the loop, i.e. it returns the
gradients w.r.t. the SCALED problem (Qs, ps, As, bs, lbs, ubs, rho, D); the scaling itself (:160-203: ~25 element-wise /
reduction ops, once per call) is differentiated by autograd on a small eager graph rebuilt in ``backward``.

What is left (a solve in which rho was adapted -- the reference's tape then runs through the adaptation itself --, no finite bound)
takes the eager path below: the loop as torch ops with ``TorchLU`` (HIP LU factor / cached solves) as the taped solve.
"""
import ctypes
import os

import torch

from . import _lib
from .lu_layer import TorchLU
from .utils import get_ncon

_INF = float("inf")
_TINY = 1e-16


def _floor_nonpositive(norms):
    """entries <= 0 -> max(row mean, 1e-6)   (reference :164-168, :182-186)"""
    bad = norms <= 0.0
    if torch.any(bad):
        floor = torch.clamp(norms.mean(dim=1), min=1e-6).unsqueeze(1)
        repl = torch.clamp(norms, min=floor)
        norms = torch.where(bad, repl, norms)
    return norms


def _inf_norm(v):
    return torch.linalg.norm(v, ord=_INF, dim=1, keepdim=True)


def _scaled_problem(Q, p, A, b, lb, ub, r, has_box, colmax=None, fro=None):
    """The reference's pre-conditioning (:160-203) as differentiable torch ops -> (Qs, ps, As, bs, lbs, ubs, D, E, rho);
    D / E are 1.0 without scaling, rho may be a python number.
    colmax (B,n) / fro (B,1,1) given: the column maxima of |Q| (:163) and ||Qs||_F (:201) come from the caller (leaves of its own
    graph, formed by the library's one-pass kernels) and Qs is NOT formed: the first return value is then the scaling vector d
    (B,n) (None without scaling) -- the n-sized part of the chain only, see _UnrolledLoop.backward."""
    dev = p.device
    n = p.shape[1]
    m = get_ncon(A, dim=1)
    rho = r['rho']
    if not has_box:
        rho = 0
    D = E = 1.0
    d = None
    if r['scale']:
        d = torch.sqrt(1 / _floor_nonpositive(torch.linalg.norm(Q, ord=_INF, dim=1) if colmax is None else colmax))
        beta = r['beta']
        if beta is None:
            q = torch.quantile(d, q=torch.tensor([0.10, 0.90], dtype=d.dtype, device=dev), dim=1)
            beta = (1 - q[[0]] / q[[1]]).T
        d = (1 - beta) * d + beta * d.mean(dim=1, keepdim=True)
        if colmax is None:
            Q = d.unsqueeze(2) * Q * d.unsqueeze(1)
        p = d.unsqueeze(2) * p
        if m > 0:
            A = A * d.unsqueeze(1)
            E = (1 / _floor_nonpositive(torch.linalg.norm(A, ord=_INF, dim=2))).unsqueeze(2)
            A = E * A
            b = E * b
        D = d.unsqueeze(2)
        if has_box:
            lb, ub = lb / D, ub / D
    if rho is None:
        rho = torch.clamp((torch.linalg.matrix_norm(Q, keepdim=True) if fro is None else fro) / n ** 0.5, min=r['rho_min'], max=r['rho_max'])
    return (Q if colmax is None and fro is None else d), p, A, b, lb, ub, D, E, rho


class _UnrolledLoop(torch.autograd.Function):
    """forward: the persistent HIP solve on a workspace of its own (kept for the backward); backward: the reverse sweep of
    lqp_boxqp_unroll_backward + the scaling chain by autograd.  Raises _NotNative when the solve did not take the
    symmetric x-update with a constant factor."""

    @staticmethod
    def forward(ctx, Q, p, A, b, lb, ub, control, r, bounds):
        from .solve_box_qp_admm_torch import _forward_solve
        lib = _lib.load()
        B, n = Q.shape[0], p.shape[1]
        m = get_ncon(A, dim=1)
        ws = torch.empty(int(lib.lqp_boxqp_forward_workspace_bytes(_lib.dtype_code(p), B, n, m)), dtype=torch.uint8, device=p.device)
        sol = _forward_solve(Q, p, A, b, lb, ub, control, bounds=bounds, sync=True, private_ws=ws, keep_factor=True)
        st = sol['_stats']
        # one factor for the whole solve: the symmetric x-update (float32: the packed inverse) or the pivoted LU (any dtype: the packed
        # factor, lqp_boxqp_unroll_backward_lu); a solve in which rho was adapted keeps the eager tape
        if st['n_factor'] != 1 or st['linsolve_used'] not in (1, 2) or (st['linsolve_used'] == 2 and p.dtype != torch.float32):
            raise _NotNative()
        ctx.lu = st['linsolve_used'] == 1
        ctx.ws, ctx.iters, ctx.r, ctx.has_box = ws, int(st['iters']), r, bool(bounds[0] or bounds[1])
        ctx.rho_fwd = sol['rho'] if torch.is_tensor(sol['rho']) else None      # (B,1,1): clamp(||Qs||_F / sqrt(n)) when rho was not given
        ctx.save_for_backward(Q, p, A, b, lb, ub)
        return sol['x']

    @staticmethod
    def backward(ctx, g):
        Q, p, A, b, lb, ub = ctx.saved_tensors
        lib = _lib.load()
        B, n = Q.shape[0], p.shape[1]
        m = get_ncon(A, dim=1)
        dev, dt = p.device, p.dtype
        need = ctx.needs_input_grad
        mk = lambda *shape: torch.empty(shape, dtype=dt, device=dev)
        dQs = mk(B, n, n) if need[0] else None
        dps, dlbs, dubs, dD, drho = mk(B, n, 1), mk(B, n, 1), mk(B, n, 1), mk(B, n, 1), mk(B, 1, 1)
        dAs, dbs = (mk(B, m, n), mk(B, m, 1)) if m > 0 else (None, None)
        stream = torch.cuda.current_stream(dev).cuda_stream
        gc = _lib.norm(g, dt)
        if ctx.lu:
            dtc = _lib.dtype_code(p)
            nbytes = lib.lqp_boxqp_unroll_backward_lu_workspace_bytes(dtc, B, n, m, ctx.iters)
            scratch = _lib.workspace(dev, nbytes, "unroll", stream)
            with _lib.on_device(dev):
                _lib.check(lib.lqp_boxqp_unroll_backward_lu(
                    ctypes.c_void_p(stream), dtc, B, n, m, _lib.ptr(ctx.ws), ctx.ws.numel(), ctx.iters, _lib.ptr(gc),
                    _lib.ptr(dQs), _lib.ptr(dps), _lib.ptr(dAs), _lib.ptr(dbs), _lib.ptr(dlbs), _lib.ptr(dubs),
                    _lib.ptr(drho), _lib.ptr(dD), _lib.ptr(scratch), scratch.numel()), "unroll_backward_lu")
        else:
            nbytes = lib.lqp_boxqp_unroll_backward_workspace_bytes(B, n, m, ctx.iters)
            scratch = _lib.workspace(dev, nbytes, "unroll", stream)
            with _lib.on_device(dev):
                _lib.check(lib.lqp_boxqp_unroll_backward(
                    ctypes.c_void_p(stream), B, n, m, _lib.ptr(ctx.ws), ctx.ws.numel(), ctx.iters, _lib.ptr(gc),
                    _lib.ptr(dQs), _lib.ptr(dps), _lib.ptr(dAs), _lib.ptr(dbs), _lib.ptr(dlbs), _lib.ptr(dubs),
                    _lib.ptr(drho), _lib.ptr(dD), _lib.ptr(scratch), scratch.numel()), "unroll_backward")
        ctx.ws = None
        if need[0] and dt == torch.float32 and os.environ.get("LQP_UNROLL_SCALE_NATIVE", "1") != "0":
            return _scaling_backward_native(ctx, lib, stream, (Q, p, A, b, lb, ub), need,
                                            dict(dQs=dQs, dps=dps, dAs=dAs, dbs=dbs, dlbs=dlbs, dubs=dubs, dD=dD, drho=drho))
        # ---- the scaling (:160-203) by autograd: leaves -> (Qs, ps, As, bs, lbs, ubs, D, rho) ----
        leaves = [None if t is None else t.detach().requires_grad_(bool(nd)) for t, nd in zip((Q, p, A, b, lb, ub), need[:6])]
        with torch.enable_grad():
            Qs, ps, As, bs, lbs, ubs, D, _E, rho = _scaled_problem(*leaves, ctx.r, ctx.has_box)
        pairs = [(Qs, dQs), (ps, dps), (As, dAs), (bs, dbs), (lbs, dlbs), (ubs, dubs), (D, dD), (rho, drho)]
        outs = [(o, go) for o, go in pairs if torch.is_tensor(o) and o.requires_grad and go is not None]
        wanted = [t for t in leaves if t is not None and t.requires_grad]
        grads = iter(torch.autograd.grad([o for o, _ in outs], wanted, [go for _, go in outs], allow_unused=True) if outs and wanted
                     else [None] * len(wanted))
        res = [next(grads) if (t is not None and t.requires_grad) else None for t in leaves]
        return tuple(res) + (None, None, None)


def _scaling_backward_native(ctx, lib, stream, inputs, need, up):
    """The scaling chain (:160-203) behind the unrolled loop on the library's kernels (include/lqp_amd.h, lqp_unroll_scale_*): column
    maxima of |Q|, the n-sized chain forward (-> the scaling vector), the backward of Qs = D Q D in place over dQs with the
    reductions dL/dd needs, the n-sized chain backward, the scatter of the maxima's gradient: five launches.  As eager torch ops the
    chain was ~150 launches per backward, ~25 of them passes over 128 MB at the headline size -- 2.0 of the 4.2 ms of kernels in an
    unroll step, and host-bound behind that.  LQP_UNROLL_SCALE_NATIVE=2 (and a per-problem beta tensor) keeps autograd for the
    n-sized part; =0 for everything."""
    Q, p, A, b, lb, ub = inputs
    B, n = Q.shape[0], p.shape[1]
    dev, dt = p.device, p.dtype
    r = ctx.r
    Qc = _lib.norm(Q, dt)
    sp = ctypes.c_void_p(stream)
    G = up['dQs']                                          # dL/dQs in, dL/dQ out (in place)
    scale = bool(r['scale'])
    rho_from_norm = r['rho'] is None and ctx.has_box
    m = get_ncon(A, dim=1)
    beta = r['beta']
    all_native = os.environ.get("LQP_UNROLL_SCALE_NATIVE", "1") != "2" and (not scale or beta is None or not torch.is_tensor(beta))
    if all_native:
        # ---- everything on the library's kernels: five launches, no autograd graph (a per-problem beta tensor keeps the hybrid below) ----
        s = None
        if rho_from_norm and ctx.rho_fwd is not None:
            rho_f = ctx.rho_fwd.reshape(B).to(dt)
            inside = (rho_f > r['rho_min']) & (rho_f < r['rho_max'])
            s = torch.where(inside, up['drho'].reshape(B) / (n * rho_f), torch.zeros_like(rho_f)).contiguous()
        with _lib.on_device(dev):
            slabs = int(lib.lqp_unroll_scale_grad_slabs(B, n))
            parts = torch.empty((B, 1 + slabs, n), dtype=dt, device=dev)
            if not scale:
                _lib.check(lib.lqp_unroll_scale_grad(sp, B, n, _lib.ptr(Qc), None, _lib.ptr(s), _lib.ptr(G), _lib.ptr(parts), slabs), "unroll_scale_grad")
                res = [up['dps'], up['dAs'], up['dbs'], up['dlbs'], up['dubs']]
                return (G,) + tuple(g if (t is not None and nd) else None for g, t, nd in zip(res, (p, A, b, lb, ub), need[1:6])) + (None, None, None)
            cn = torch.empty((B, n), dtype=dt, device=dev)
            arg = torch.empty((B, n), dtype=torch.int32, device=dev)
            cnt = torch.empty((B, n), dtype=torch.int32, device=dev)
            dvec = torch.empty((B, n), dtype=dt, device=dev)
            gcn = torch.empty((B, n), dtype=dt, device=dev)
            pc, Ac, bc, lbc, ubc = (_lib.norm(t, dt) for t in (p, A, b, lb, ub))
            mk = lambda t, nd: torch.empty_like(t) if (t is not None and nd) else None
            dp, dA, db, dlb, dub = (mk(t, nd) for t, nd in zip((pc, Ac, bc, lbc, ubc), need[1:6]))
            bg, bv = (0, 0.0) if beta is None else (1, float(beta))
            _lib.check(lib.lqp_unroll_scale_colmax(sp, B, n, _lib.ptr(Qc), _lib.ptr(cn), _lib.ptr(arg), _lib.ptr(cnt)), "unroll_scale_colmax")
            vec = lambda phase, *tail: lib.lqp_unroll_scale_vectors(sp, B, n, m, phase, int(ctx.has_box), bg, bv, _lib.ptr(cn), _lib.ptr(pc),
                                                                    _lib.ptr(Ac), _lib.ptr(bc), _lib.ptr(lbc), _lib.ptr(ubc), *tail)
            _lib.check(vec(0, None, None, None, None, None, None, None, 0, _lib.ptr(dvec), None, None, None, None, None, None), "unroll_scale_vectors")
            _lib.check(lib.lqp_unroll_scale_grad(sp, B, n, _lib.ptr(Qc), _lib.ptr(dvec), _lib.ptr(s), _lib.ptr(G), _lib.ptr(parts), slabs), "unroll_scale_grad")
            _lib.check(vec(1, _lib.ptr(up['dps']), _lib.ptr(up['dAs']), _lib.ptr(up['dbs']), _lib.ptr(up['dlbs']), _lib.ptr(up['dubs']),
                           _lib.ptr(up['dD']), _lib.ptr(parts), 1 + slabs, None, _lib.ptr(dp), _lib.ptr(dA), _lib.ptr(db), _lib.ptr(dlb),
                           _lib.ptr(dub), _lib.ptr(gcn)), "unroll_scale_vectors")
            _lib.check(lib.lqp_unroll_scale_scatter(sp, B, n, _lib.ptr(Qc), _lib.ptr(cn), _lib.ptr(arg), _lib.ptr(cnt), _lib.ptr(gcn), _lib.ptr(G)),
                       "unroll_scale_scatter")
        shaped = [None if g is None else g.reshape(t.shape) for g, t in zip((dp, dA, db, dlb, dub), (p, A, b, lb, ub))]
        return (G,) + tuple(shaped) + (None, None, None)
    leaves = [None] + [None if t is None else t.detach().requires_grad_(bool(nd)) for t, nd in zip((p, A, b, lb, ub), need[1:6])]
    cn = arg = cnt = fro = None
    with _lib.on_device(dev):
        if scale:
            cn = torch.empty((B, n), dtype=dt, device=dev)
            arg = torch.empty((B, n), dtype=torch.int32, device=dev)
            cnt = torch.empty((B, n), dtype=torch.int32, device=dev)
            _lib.check(lib.lqp_unroll_scale_colmax(sp, B, n, _lib.ptr(Qc), _lib.ptr(cn), _lib.ptr(arg), _lib.ptr(cnt)), "unroll_scale_colmax")
            cn.requires_grad_(True)
        with torch.enable_grad():
            d, ps, As, bs, lbs, ubs, D, _E, rho = _scaled_problem(None, *leaves[1:], r, ctx.has_box,
                                                                  colmax=cn if scale else torch.empty(0),
                                                                  fro=torch.empty(0) if rho_from_norm else None)
        dvec = _lib.norm(d.detach(), dt) if scale else None
        s = None
        if rho_from_norm and ctx.rho_fwd is not None:
            # rho = clamp(||Qs||_F / sqrt(n)) (:201-203): inside the clamp ||Qs||_F = rho sqrt(n) with the FORWARD's rho and
            # dL/dQs += drho / sqrt(n) * Qs / ||Qs||_F = (drho / (n rho)) Qs; on the clamp nothing passes (no pass over Q for the norm)
            rho_f = ctx.rho_fwd.reshape(B).to(dt)
            inside = (rho_f > r['rho_min']) & (rho_f < r['rho_max'])
            s = torch.where(inside, up['drho'].reshape(B) / (n * rho_f), torch.zeros_like(rho_f)).contiguous()
        slabs = int(lib.lqp_unroll_scale_grad_slabs(B, n))
        parts = torch.empty((B, 1 + slabs, n), dtype=dt, device=dev)
        _lib.check(lib.lqp_unroll_scale_grad(sp, B, n, _lib.ptr(Qc), _lib.ptr(dvec), _lib.ptr(s), _lib.ptr(G), _lib.ptr(parts), slabs),
                   "unroll_scale_grad")
        # ---- the n-sized rest by autograd: (cn, p, A, b, lb, ub) -> (d, ps, As, bs, lbs, ubs, D) ----
        pairs = [(ps, up['dps']), (As, up['dAs']), (bs, up['dbs']), (lbs, up['dlbs']), (ubs, up['dubs']), (D, up['dD'])]
        if scale:
            pairs.append((d, parts.sum(dim=1)))
        outs = [(o, go) for o, go in pairs if torch.is_tensor(o) and o.requires_grad and go is not None]
        wanted = [t for t in ([cn] if scale else []) + leaves[1:] if t is not None and t.requires_grad]
        grads = iter(torch.autograd.grad([o for o, _ in outs], wanted, [go.reshape(o.shape) for o, go in outs], allow_unused=True)
                     if outs and wanted else [None] * len(wanted))
        gcn = next(grads) if scale else None
        res = [next(grads) if (t is not None and t.requires_grad) else None for t in leaves[1:]]
        if scale and gcn is not None:
            _lib.check(lib.lqp_unroll_scale_scatter(sp, B, n, _lib.ptr(Qc), _lib.ptr(cn.detach()), _lib.ptr(arg), _lib.ptr(cnt),
                                                    _lib.ptr(gcn.contiguous()), _lib.ptr(G)), "unroll_scale_scatter")
    return (G,) + tuple(res) + (None, None, None)


class _NotNative(Exception):
    pass


def unrolled_solve_box_qp(Q, p, A, b, lb, ub, r, has_lb, has_ub, control=None):
    """``r`` is the resolved control (solve_box_qp_admm_torch.resolve_control). Returns x only,
    as the reference does in unroll mode (:328-329)."""
    if (control is not None and p.dtype in (torch.float32, torch.float64) and (has_lb or has_ub)
            and os.environ.get("LQP_UNROLL_NATIVE", "1") != "0"):
        try:
            ctl = {k: v for k, v in control.items() if k != 'unroll'}
            return _UnrolledLoop.apply(Q, p, A, b, lb, ub, ctl, r, (has_lb, has_ub))
        except _NotNative:
            pass                   # (LU path / adapted rho: the taped loop below)
    return _eager_unrolled(Q, p, A, b, lb, ub, r, has_lb, has_ub)


def _eager_unrolled(Q, p, A, b, lb, ub, r, has_lb, has_ub, solver_cls=TorchLU):
    """The loop as ordinary differentiable torch ops; every x-update is ``TorchLU`` (lqp_py_amd/lu_layer.py): HIP batched
    LU factor + cached solves with the analytic backward of lu_layer.py:41-58.  (solver_cls: tests substitute a CPU
    float64 stand-in for the HIP layer to obtain a higher-precision truth of the same taped computation.)"""
    dev, dt = p.device, p.dtype
    B, n = Q.shape[0], p.shape[1]
    m = get_ncon(A, dim=1)
    has_box = has_lb or has_ub
    p_inf = _inf_norm(p)
    Q, p, A, b, lb, ub, D, E, rho = _scaled_problem(Q, p, A, b, lb, ub, r, has_box)

    eye = torch.eye(n, dtype=dt, device=dev).unsqueeze(0)

    def kkt(rho_now):
        M = Q + rho_now * eye
        if m > 0:
            corner = torch.zeros(B, m, m, dtype=dt, device=dev)
            M = torch.cat((torch.cat((M, A.transpose(1, 2)), 2), torch.cat((A, corner), 2)), 1)
        return M

    M = kkt(rho)
    solver = solver_cls(A=M)                      # HIP factorisation, no_grad inside

    x = z = u = torch.zeros(B, n, 1, dtype=dt, device=dev)
    tiny = torch.full((1,), _TINY, dtype=dt, device=dev)
    thr = torch.full((1,), float(r['adaptive_rho_threshold']), dtype=dt, device=dev)
    r_inf = s_inf = pri = dua = None
    wants = r['adaptive_rho']
    for it in range(r['max_iters']):
        if (r['adaptive_rho'] and it % r['adaptive_rho_iter'] == 0 and 0 < it < r['adaptive_rho_max_iter']
                and bool(torch.any(wants))):
            ratio = (torch.clamp(r_inf / pri, min=_TINY) / torch.clamp(s_inf / dua, min=_TINY)) ** 0.5
            if bool((ratio > r['adaptive_rho_tol']).any()) or bool((ratio < 1 / r['adaptive_rho_tol']).any()):
                rho = rho * torch.logical_not(wants) + (rho * ratio) * wants
                rho = torch.clamp(rho, min=r['rho_min'], max=r['rho_max'])
                M = kkt(rho)
                solver = solver_cls(A=M)
        rhs = -p + rho * (z - u)
        if m > 0:
            rhs = torch.cat((rhs, b), 1)
        xv = solver(A=M, b=rhs)
        x = xv[:, :n, :]
        z_old = z
        z = x + u
        if has_lb:
            z = torch.maximum(z, lb)
        if has_ub:
            z = torch.minimum(z, ub)
        res = x - z
        s = rho * (z - z_old)
        u = u + res
        if it % r['check_solved'] == 0:
            r_inf, s_inf = _inf_norm(D * res), _inf_norm(D * s)
            pri = torch.maximum(torch.maximum(_inf_norm(D * x), _inf_norm(D * z)), tiny)
            dua = torch.maximum(torch.maximum(torch.maximum(_inf_norm(rho * D * u), _inf_norm(torch.matmul(Q, x) / D)),
                                              p_inf), tiny)
            tol_p = r['eps_abs'] + r['eps_rel'] * pri
            tol_d = r['eps_abs'] + r['eps_rel'] * dua
            wants = torch.logical_or(r_inf > torch.maximum(tol_p, thr), s_inf > torch.maximum(tol_d, thr))
            if bool(torch.all(torch.logical_and(r_inf < tol_p, s_inf < tol_d))):
                break
    return D * x
