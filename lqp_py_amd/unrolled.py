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

A solve in which rho was ADAPTED (the reference's tape then runs through the adaptation itself, :237-256): the tape is walked
epoch by epoch in the library on the pivoted LU of each epoch's KKT matrix (``lqp_boxqp_unroll_tape_segment``) and the adaptation --
a few norms of the iterates of ONE check per event -- is differentiated by autograd on its own small graph
(``_backward_with_rho_events``): no torch op per iteration there either.  What is left (no finite bound: rho = 0, one solve)
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
        if st['linsolve_used'] not in (1, 2) or (st['linsolve_used'] == 2 and p.dtype != torch.float32):
            raise _NotNative()
        # rho was adapted along the solve: the tape is walked epoch by epoch on the pivoted LU of each epoch's KKT matrix
        # (_backward_with_rho_events); LQP_UNROLL_EVENTS=0 keeps the taped loop of torch ops for it
        ctx.events = st['n_factor'] != 1
        if ctx.events and os.environ.get("LQP_UNROLL_EVENTS", "1") == "0":
            raise _NotNative()
        ctx.lu = st['linsolve_used'] == 1
        ctx.ws, ctx.iters, ctx.r, ctx.has_box = ws, int(st['iters']), r, bool(bounds[0] or bounds[1])
        ctx.rho_fwd = sol['rho'] if torch.is_tensor(sol['rho']) else None      # (B,1,1): clamp(||Qs||_F / sqrt(n)) when rho was not given
        ctx.save_for_backward(Q, p, A, b, lb, ub)
        return sol['x']

    @staticmethod
    def backward(ctx, g):
        if ctx.events:
            return _backward_with_rho_events(ctx, g)
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


def _check_quantities(r, rho, x, z, z_old, u, D, Qs, p_inf):
    """What a check of the loop computes (lqp_py/solve_box_qp_admm_torch.py:285-305) and the adaptive step reads (:239-251), as
    differentiable torch ops on (B,n,1) vectors: -> (ratio, wants)."""
    dt = x.dtype
    tiny = torch.full((1,), _TINY, dtype=dt, device=x.device)
    thr = torch.full((1,), float(r['adaptive_rho_threshold']), dtype=dt, device=x.device)
    res = x - z
    s = rho * (z - z_old)
    r_inf, s_inf = _inf_norm(D * res), _inf_norm(D * s)
    pri = torch.maximum(torch.maximum(_inf_norm(D * x), _inf_norm(D * z)), tiny)
    dua = torch.maximum(torch.maximum(torch.maximum(_inf_norm(rho * D * u), _inf_norm(torch.matmul(Qs, x) / D)), p_inf), tiny)
    tol_p = r['eps_abs'] + r['eps_rel'] * pri
    tol_d = r['eps_abs'] + r['eps_rel'] * dua
    wants = torch.logical_or(r_inf > torch.maximum(tol_p, thr), s_inf > torch.maximum(tol_d, thr))
    ratio = (torch.clamp(r_inf / pri, min=_TINY) / torch.clamp(s_inf / dua, min=_TINY)) ** 0.5
    return ratio, wants


def _adapted_rho(r, rho, ratio, wants):
    """:246-251"""
    rho = rho * torch.logical_not(wants) + (rho * ratio) * wants
    return torch.clamp(rho, min=r['rho_min'], max=r['rho_max'])


def _backward_with_rho_events(ctx, g):
    """The tape of a solve in which rho was ADAPTED (solve_box_qp_admm_torch.py:237-256 inside the unrolled loop): the factor
    changes along it, so it is walked epoch by epoch in the library (lqp_boxqp_unroll_tape_segment: every x-update a pair of cached
    triangular solves with the pivoted LU of THAT epoch's KKT matrix, TorchLULayer's node), and the adaptation itself -- a few
    norms of the iterates of one check, once per event -- is differentiated by autograd on its own small graph: its gradient
    w.r.t. the iterates goes back into the sweep as injected cotangents, w.r.t. the previous rho into that epoch's rho.  No torch op
    per iteration.  The epochs are re-derived by replaying the loop in segments between the possible events (multiples of
    adaptive_rho_iter) with the reference's own decision rule on the replayed iterates."""
    from .lu_layer import lu_factor
    Q, p, A, b, lb, ub = ctx.saved_tensors
    lib = _lib.load()
    r = ctx.r
    B, n = Q.shape[0], p.shape[1]
    m = get_ncon(A, dim=1)
    N = n + m
    dev, dt = p.device, p.dtype
    dtc = _lib.dtype_code(p)
    need = ctx.needs_input_grad
    T = ctx.iters + 1
    stream = torch.cuda.current_stream(dev).cuda_stream
    sp = ctypes.c_void_p(stream)
    # ---- the scaling (:160-203) as a graph: leaves -> (Qs, ps, As, bs, lbs, ubs, D, rho0), and ||p||_inf of the unscaled p (:127) ----
    leaves = [None if t is None else t.detach().requires_grad_(bool(nd)) for t, nd in zip((Q, p, A, b, lb, ub), need[:6])]
    with torch.enable_grad():
        Qs, ps, As, bs, lbs, ubs, D, _E, rho0 = _scaled_problem(*leaves, r, ctx.has_box)
        p_inf = _inf_norm(leaves[1])
    ones = torch.ones(B, n, 1, dtype=dt, device=dev)
    Dd = D.detach() if torch.is_tensor(D) else ones
    Qsd = Qs.detach()
    Asd = As.detach() if m > 0 else None
    rho_e = (rho0.detach().reshape(B, 1, 1).to(dt) if torch.is_tensor(rho0) else torch.full((B, 1, 1), float(rho0), dtype=dt, device=dev)).clone()
    eye = torch.eye(n, dtype=dt, device=dev).unsqueeze(0)

    def packed_factor(rho_now):
        M = Qsd + rho_now * eye
        if m > 0:
            corner = torch.zeros(B, m, m, dtype=dt, device=dev)
            M = torch.cat((torch.cat((M, Asd.transpose(1, 2)), 2), torch.cat((Asd, corner), 2)), 1)
        LU, piv = lu_factor(M)
        buf = torch.empty(lib.lqp_lu_packed_bytes(dtc, B, N), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            _lib.check(lib.lqp_lu_pack(sp, dtc, B, N, _lib.ptr(LU), _lib.ptr(piv), _lib.ptr(buf)), "lu_pack")
        return buf

    nbytes = lib.lqp_boxqp_unroll_tape_workspace_bytes(dtc, B, n, m, ctx.iters)
    scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    gc = _lib.norm(g, dt)
    mk = lambda *shape: torch.zeros(shape, dtype=dt, device=dev)
    dps, dlbs, dubs, dD = mk(B, n, 1), mk(B, n, 1), mk(B, n, 1), mk(B, n, 1)

    def segment(k0, k1, mode, packed, rho_now, state, inj_k=-1, inj=None, drho=None, rows=None):
        zp, up, xp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        with _lib.on_device(dev):
            _lib.check(lib.lqp_boxqp_unroll_tape_segment(
                sp, dtc, B, n, m, _lib.ptr(ctx.ws), ctx.ws.numel(), ctx.iters, k0, k1, mode, _lib.ptr(packed),
                _lib.ptr(rho_now), _lib.ptr(state), inj_k, _lib.ptr(inj), _lib.ptr(gc), _lib.ptr(dps), _lib.ptr(dlbs), _lib.ptr(dubs),
                _lib.ptr(drho), _lib.ptr(dD), _lib.ptr(scratch), scratch.numel(),
                ctypes.byref(zp) if rows else None, ctypes.byref(up) if rows else None, ctypes.byref(xp) if rows else None),
                "unroll_tape_segment")
        return zp.value, up.value, xp.value

    def rows_view(ptr_value):
        off = ptr_value - scratch.data_ptr()
        return scratch[off: off + B * T * n * p.element_size()].view(dt).view(B, T, n)

    zp, up, xp = segment(0, 0, 0, None, None, None, rows=True)       # (where the scratch keeps z_{k+1}, u_{k+1}, x_k)
    Zr, Ur, Xr = rows_view(zp), rows_view(up), rows_view(xp)

    # ---- replay, segment by segment: the epochs of the tape ----
    ar, ar_iter, ar_max, chk = bool(r['adaptive_rho']), int(r['adaptive_rho_iter']), int(r['adaptive_rho_max_iter']), int(r['check_solved'])
    state = mk(B, 2, n)
    segs = []                 # (k0, k1, epoch)
    epochs = [dict(rho=rho_e.reshape(B).contiguous(), packed=packed_factor(rho_e), event=None)]
    k, wants, last = 0, None, None
    while k < T:
        if ar and k % ar_iter == 0 and 0 < k < ar_max and wants is not None and bool(torch.any(wants)):
            ratio, _ = _check_quantities(r, rho_e, *last)
            if bool((ratio > r['adaptive_rho_tol']).any()) or bool((ratio < 1 / r['adaptive_rho_tol']).any()):
                epochs[-1]['event'] = dict(check=last, wants=wants, c=last_c)
                rho_e = _adapted_rho(r, rho_e, ratio, wants)
                epochs.append(dict(rho=rho_e.reshape(B).contiguous(), packed=packed_factor(rho_e), event=None))
        k1 = T
        if ar:
            nxt = (k // ar_iter + 1) * ar_iter
            if nxt < ar_max and nxt < T:
                k1 = nxt
        e = epochs[-1]
        segment(k, k1, 1, e['packed'], e['rho'], state)
        segs.append((k, k1, len(epochs) - 1))
        if k1 < T:
            last_c = ((k1 - 1) // chk) * chk
            col = lambda R, kk: R[:, kk, :].unsqueeze(2).clone()
            z_old = col(Zr, last_c - 1) if last_c > 0 else mk(B, n, 1)
            last = (col(Xr, last_c), col(Zr, last_c), z_old, col(Ur, last_c), Dd, Qsd, p_inf.detach())
            _, wants = _check_quantities(r, rho_e, *last)
        k = k1

    # ---- walk back: segments in reverse, the adaptation of every event by autograd on its own graph ----
    rho_bar = [mk(B) for _ in epochs]
    dD_extra, dQs_extra, dpinf = mk(B, n, 1), None, mk(B, 1, 1)
    sbar = mk(B, 2, n)
    for (k0, k1, ei) in reversed(segs):
        e = epochs[ei]
        inj, inj_k = None, -1
        ev = e['event']
        if ev is not None and ei + 1 < len(epochs) and k1 == min(kk for (kk, _, ej) in segs if ej == ei + 1):
            # rho_{e+1} = adapt(rho_e, the iterates of check c): its cotangent is complete now (every later segment is walked)
            with torch.enable_grad():
                small = [t.detach().clone().requires_grad_(True) for t in ev['check']]
                rl = e['rho'].reshape(B, 1, 1).detach().clone().requires_grad_(True)
                ratio, _ = _check_quantities(r, rl, *small)
                rho_new = _adapted_rho(r, rl, ratio, ev['wants'])
                gr = torch.autograd.grad(rho_new, small + [rl], rho_bar[ei + 1].reshape(B, 1, 1), allow_unused=True)
            zero = lambda t, like: torch.zeros_like(like) if t is None else t
            gx, gz1, gz0, gu1, gDl, gQl, gpl, grl = [zero(t, like) for t, like in zip(gr, small + [rl])]
            inj = torch.stack((gx, gz1, gu1, gz0), 1).reshape(B, 4, n).contiguous()
            inj_k = ev['c']
            rho_bar[ei] += grl.reshape(B)
            dD_extra += gDl
            dQs_extra = gQl if dQs_extra is None else dQs_extra + gQl
            dpinf += gpl
        drho_seg = mk(B)
        segment(k0, k1, 2, e['packed'], e['rho'], sbar, inj_k=inj_k, inj=inj, drho=drho_seg)
        rho_bar[ei] += drho_seg
    dQs = torch.empty(B, n, n, dtype=dt, device=dev) if need[0] else None
    dAs, dbs = (mk(B, m, n), mk(B, m, 1)) if m > 0 else (None, None)
    with _lib.on_device(dev):
        _lib.check(lib.lqp_boxqp_unroll_tape_finish(sp, dtc, B, n, m, ctx.iters, _lib.ptr(dQs), _lib.ptr(dAs), _lib.ptr(dbs),
                                                   _lib.ptr(scratch), scratch.numel()), "unroll_tape_finish")
    if dQs is not None and dQs_extra is not None:
        dQs = dQs + dQs_extra
    ctx.ws = None
    # ---- the scaling chain by autograd ----
    pairs = [(Qs, dQs), (ps, dps), (As, dAs), (bs, dbs), (lbs, dlbs), (ubs, dubs), (D, dD + dD_extra), (rho0, rho_bar[0].reshape(B, 1, 1)),
             (p_inf, dpinf)]
    outs = [(o, go) for o, go in pairs if torch.is_tensor(o) and o.requires_grad and go is not None]
    wanted = [t for t in leaves if t is not None and t.requires_grad]
    grads = iter(torch.autograd.grad([o for o, _ in outs], wanted, [go.reshape(o.shape) for o, go in outs], allow_unused=True)
                 if outs and wanted else [None] * len(wanted))
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
