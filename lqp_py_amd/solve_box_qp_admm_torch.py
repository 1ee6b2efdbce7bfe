"""Batched box-constrained QP layer on MI355X: ADMM forward + fixed-point
implicit backward, behind the reference's Python surface.

    x* = argmin_x 0.5 x^T Q x + p^T x   s.t.  A x = b,  lb <= x <= ub

Drop-in for ``lqp_py/solve_box_qp_admm_torch.py`` of ipo-lab/lqp_py:
``SolveBoxQP`` (:7-18), ``SolveBoxQPLayer`` (:21-67), ``BoxQPTH`` (:70-105),
``torch_solve_box_qp`` (:108-333), ``torch_solve_box_qp_grad`` (:349-432).
Everything numeric runs in the HIP library (``lqp_boxqp_forward`` /
``lqp_boxqp_backward_fp``, include/lqp_amd.h); this module only resolves the
control dict -- including its key-name traps and the caller-dict side effect --
and moves pointers.  Tensors must be on the GPU; there is no CPU fallback.
"""
import ctypes
import os
import threading
import weakref

import torch
import torch.nn as nn

from . import _lib
from .utils import get_ncon

_INF = float("inf")
_SYNC_SPLIT = os.environ.get("LQP_SYNC_SPLIT", "1") != "0"      # (A/B knob: 0 = one-call synchronous forward)
_KKT_NATIVE = os.environ.get("LQP_KKT_NATIVE", "1") != "0"      # (A/B knob: 0 = the KKT backward composed from torch ops + lqp_kkt_solve)
_PREPARE_BWD = os.environ.get("LQP_PREPARE_BWD", "1") != "0"    # (A/B knob: 0 = the backward prepares itself when it is called)
_PREFACTOR_BWD = os.environ.get("LQP_PREFACTOR_BWD", "1") != "0"   # (A/B knob: 0 = nothing of the backward runs ahead of the cotangent)
_BWD_PREFACTORED = 0x100                                         # include/lqp_amd.h: LQP_BWD_PREFACTORED
_BWD_REPORTED = 0x200                                            # include/lqp_amd.h: LQP_BWD_REPORTED


class SolveBoxQP(nn.Module):
    """nn.Module holding the control dict (reference :7-18)."""

    def __init__(self, control):
        super().__init__()
        self.control = control

    def forward(self, Q, p, A, b, lb, ub):
        if self.control.get('unroll', False):
            return torch_solve_box_qp(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, control=self.control)
        # what the layer remembers between calls (did the last batch hold a finite bound?) it remembers per MODULE, in a
        # weak-keyed side table -- never in the caller's dict
        prev = getattr(_tls, 'holder', None)
        _tls.holder = self
        try:
            return SolveBoxQPLayer.apply(Q, p, A, b, lb, ub, self.control)
        finally:
            _tls.holder = prev


class SolveBoxQPLayer(torch.autograd.Function):
    """ADMM forward solve / fixed-point implicit backward (reference :21-67)."""

    @staticmethod
    def forward(ctx, Q, p, A, b, lb, ub, control):
        # Default (control['sync'] absent or True): the reference's semantics -- the call waits for the solve, a
        # singular KKT matrix raises HERE (:215), a Q outside the symmetric x-update falls back to LU by itself.
        # control['sync'] = False (extension, for training loops): the whole schedule is enqueued and the call
        # returns at once; an error is raised by a later call / lqp_py_amd.synchronize(), and the outputs of a
        # failed solve are NaN, never plausible numbers.
        # Whether any bound is finite (:33-38) is found on the device; when none is, control['rho'] = 0 is written into
        # the CALLER's dict as the reference does (_forward_solve, mutate=True).
        # (lqp_py_amd.dist passes the flags of the WHOLE batch when this call holds one shard of it.)
        sync = bool(control.get('sync', True))
        ctx.backward_method = control.get('backward', 'fixed_point')
        ctx.prepared = None
        while_running = None
        need = ctx.needs_input_grad
        if sync and ctx.backward_method != 'kkt' and any(need[:6]) and _PREPARE_BWD:
            # a synchronous call only waits while its schedule runs: the backward's outputs, workspace and argument list are
            # made in that window (the cotangent is all that is missing), see _fp_backward_prepare
            def while_running(parts, linsolve_used):
                if parts is None:          # (the solve is being repeated on another schedule: what was prepared is void)
                    ctx.prepared = None
                    return
                # ... and the part of the backward that does not need the cotangent (free set, Q_FF, its Cholesky
                # factorisation) is enqueued right behind the forward: it runs while this call returns, the caller forms its
                # loss and autograd finds its way to `backward`, which then only solves
                ctx.prepared = _fp_backward_prepare(parts['x'], parts['u'], parts['lams'], parts['nus'], Q, A, lb, ub,
                                                    parts['rho_out'], _wanted(need, A), sync=True, linsolve=linsolve_used,
                                                    prefactor=_PREFACTOR_BWD)
        sol = _forward_solve(Q, p, A, b, lb, ub, control, bounds=control.get('_global_bounds'), sync=sync,
                             check_hook=control.get('_check_hook'), mutate=True,
                             holder=control.get('_holder') or getattr(_tls, 'holder', None), while_running=while_running)
        ctx.rho = sol['rho']
        st_ = sol['_stats']
        ctx.bound_flags = (bool(st_['any_lb']), bool(st_['any_ub'])) if st_['any_lb'] >= 0 else None     # (pipelined: not known)
        ctx.linsolve = int(sol['_stats']['linsolve_used'])     # 2: Q was checked symmetric by the forward
        ctx.sync = sync
        ctx.save_for_backward(sol['x'], sol['u'], sol['lams'], sol['nus'], Q, A, lb, ub)
        return sol['x']

    @staticmethod
    def backward(ctx, dl_dz):
        x, u, lams, nus, Q, A, lb, ub = ctx.saved_tensors
        if ctx.backward_method == 'kkt':
            return _kkt_backward(dl_dz, x, lams, nus, Q, A, lb, ub, flags=ctx.bound_flags, linsolve=ctx.linsolve,
                                 want=_wanted(ctx.needs_input_grad, A), sync=ctx.sync)
        prep, ctx.prepared = ctx.prepared, None
        if prep is not None:
            return _fp_backward_run(prep, dl_dz)
        grads = _fp_backward(dl_dz, x, u, lams, nus, Q, A, lb, ub, ctx.rho, _wanted(ctx.needs_input_grad, A), sync=ctx.sync,
                             linsolve=ctx.linsolve)
        return grads


def _wanted(need, A):
    return dict(dQ=need[0], dp=need[1], dA=need[2] and A is not None, db=need[3] and A is not None, dlb=need[4], dub=need[5])


class BoxQPTH:
    """Stateful holder (reference :70-105), including its quirk that passing
    ``lb``/``ub`` to ``update`` clears them (:99-102)."""

    def __init__(self, Q, p, A, b, lb, ub, control):
        self.Q, self.p, self.A, self.b, self.lb, self.ub = Q, p, A, b, lb, ub
        self.control = control
        self.sol = {}

    def solve(self):
        self.sol = torch_solve_box_qp(Q=self.Q, p=self.p, A=self.A, b=self.b, lb=self.lb, ub=self.ub,
                                      control=self.control)
        return self.sol.get('x')

    def update(self, Q=None, p=None, A=None, b=None, lb=None, ub=None, control=None):
        if Q is not None:
            self.Q = Q
        if p is not None:
            self.p = p
        if A is not None:
            self.A = A
        if b is not None:
            self.b = b
        if lb is not None:
            self.lb = None
        if ub is not None:
            self.ub = None
        if control is not None:
            self.control = control
        return None


def torch_solve_box_qp(Q, p, A, b, lb, ub, control):
    """Forward solve; returns {"x","z","u","lams","nus","rho","iter"} (reference :108-333)."""
    if control.get('unroll', False):
        # autograd through the loop (:328-329 returns the bare x); see lqp_py_amd/unrolled.py
        from .unrolled import unrolled_solve_box_qp
        _lib.require_gpu(Q, p, A, b, lb, ub)
        has_lb, has_ub = _finite_bounds(lb, ub)
        return unrolled_solve_box_qp(Q, p, A, b, lb, ub, resolve_control(control, p.shape[1]), has_lb, has_ub, control=control)
    return _forward_solve(Q, p, A, b, lb, ub, control, bounds=None)


def torch_solve_box_qp_grad(dl_dz, x, u, lams, nus, Q, A, lb, ub, rho):
    """Fixed-point implicit gradients -> (dQ, dp, dA, db, dlb, dub, None) (reference :349-432)."""
    has_eq = A is not None
    want = dict(dQ=True, dp=True, dA=has_eq, db=has_eq, dlb=True, dub=True)
    return _fp_backward(dl_dz, x, u, lams, nus, Q, A, lb, ub, rho, want)


def torch_solve_box_qp_grad_kkt(dl_dz, x, lams, nus, Q, A, lb, ub):
    """KKT-system backward (reference :435-469).  The reference solves the (3n+m) system
        [[Q, G^T diag(lam), A^T], [G, -diag(slack), 0], [A, 0, 0]] [dx; dlam; dnu] = [-dl_dz; 0; 0]
    with G = [-I; I].  Eliminating dlam = diag(1/slack) G dx exactly turns it into one (n+m) KKT
    solve with Q + diag(lam_lo/slack_lo + lam_hi/slack_hi) -- the same solution, on the HIP KKT
    solve (lqp_kkt_solve).  Clamps (1e-8) and the lb/ub bookkeeping of :565-584 are kept."""
    return _kkt_backward(dl_dz, x, lams, nus, Q, A, lb, ub)


def _kkt_backward(dl_dz, x, lams, nus, Q, A, lb, ub, flags=None, linsolve=1, want=None, sync=True):
    """flags: (any_lb, any_ub) when the caller knows them (the layer's forward saw them in the device's report).  With both
    true the whole backward is the library's (lqp_boxqp_backward_kkt: the reduced system on the fixed-point backward's
    kernels, gradients formed in its epilogue); since round 5 the one-sided and unbounded cases too, with the reference's
    bookkeeping of which half of dl_dh goes where (:565-584) applied to the kernel's outputs.  _KKT_NATIVE = False keeps
    the composition below."""
    from .solve_qp_eqcon_torch import _kkt_solve
    _lib.require_gpu(dl_dz, x, lams, nus, Q, A, lb, ub)
    n = Q.shape[1]
    any_lb, any_ub = flags if flags is not None else _finite_bounds(lb, ub)
    if _KKT_NATIVE:
        lib = _lib.load()
        B = Q.shape[0]
        m = get_ncon(A, dim=1)
        dt = _lib.dtype_code(x)
        dev, dty = x.device, x.dtype
        want = want or dict(dQ=True, dp=True, dA=m > 0, db=m > 0, dlb=True, dub=True)
        gc, xc, lc, nc, Qc, Ac, lbc, ubc = (_lib.norm(t, dty) for t in (dl_dz, x, lams, nus, Q, A, lb, ub))
        mk = lambda on, shape: torch.empty(shape, dtype=dty, device=dev) if on else None
        dQ, dp = mk(want['dQ'], (B, n, n)), mk(want['dp'], (B, n, 1))
        dA, db = mk(want['dA'] and m > 0, (B, m, n)), mk(want['db'] and m > 0, (B, m, 1))
        # One-sided and unbounded batches run the same kernels (round 5): the reference keeps BOTH halves of G = [-I; I] whenever any
        # bound is finite (:446-452), an infinite bound has an infinite slack, and lam / inf = 0 takes its rows out of the reduced
        # system exactly.  Only the reference's bookkeeping of dl_dh differs (:573-584): lower bounds only -> dlb; upper bounds only
        # -> dub = dl_dh[:n], the LOWER half (all zeros: its quirk, kept); no bounds -> neither.
        only_lb, only_ub = any_lb and not any_ub, any_ub and not any_lb
        dlb = mk((want['dlb'] and any_lb) or (want['dub'] and only_ub), (B, n, 1))
        dub = mk(want['dub'] and any_lb and any_ub, (B, n, 1))
        stream = _lib.current_stream_handle(dev)
        ws = _lib.workspace(dev, lib.lqp_boxqp_backward_fp_workspace_bytes(dt, B, n, m), "bwd", stream)
        fail = ctypes.c_int32(-1)
        report = _lib.host_report(B)
        with _lib.on_device(dev):
            st = lib.lqp_boxqp_backward_kkt(ctypes.c_void_p(stream), dt, B, n, m, _lib.ptr(gc), _lib.ptr(xc), _lib.ptr(lc),
                                            _lib.ptr(nc), _lib.ptr(Qc), _lib.ptr(Ac), _lib.ptr(lbc), _lib.ptr(ubc),
                                            _lib.ptr(dQ), _lib.ptr(dp), _lib.ptr(dA), _lib.ptr(db), _lib.ptr(dlb), _lib.ptr(dub),
                                            ctypes.byref(fail) if sync else None, _lib.ptr(ws), ws.numel(), int(linsolve),
                                            ctypes.c_void_p(report.data_ptr()))
        if st == 3:
            raise RuntimeError(f"lqp_py_amd.torch_solve_box_qp_grad_kkt: the input matrix is singular (batch index {fail.value})")
        _lib.check(st, "torch_solve_box_qp_grad_kkt")
        try:
            _lib.poll_errors()
        finally:
            if not sync:
                _lib.defer_check("SolveBoxQP.backward", dev, report, B, False)
            else:
                _lib._pinned_free.setdefault(report.numel(), []).append(report)
        if only_ub:
            dlb, dub = None, (-dlb if want['dub'] else None)
        elif only_lb:
            dub = None
            if not want['dlb']:
                dlb = None
        elif not any_lb:
            dlb = dub = None
        return (dQ, dp, dA, db, dlb, dub, None)
    dlam = None
    Qw = Q
    if any_lb or any_ub:
        slack = torch.clamp(torch.cat((x - lb, ub - x), dim=1), 10 ** -8)          # h - G x
        lams = torch.clamp(lams, 10 ** -8)
        w = lams[:, :n, :] / slack[:, :n, :] + lams[:, n:, :] / slack[:, n:, :]
        Qw = Q.detach().clone()
        Qw.diagonal(dim1=1, dim2=2).add_(w.squeeze(2))
    zeros = None if A is None else torch.zeros((Q.shape[0], A.shape[1], 1), dtype=Q.dtype, device=Q.device)
    dx, dnu = _kkt_solve(Qw, dl_dz, A, zeros)
    if any_lb or any_ub:
        dlam = torch.cat((-dx / slack[:, :n, :], dx / slack[:, n:, :]), dim=1)
    return torch_qp_int_grads_admm(x=x, lams=lams, nus=nus, dx=dx, dlam=dlam, dnu=dnu, any_lb=any_lb, any_ub=any_ub)


def torch_qp_int_grads(x, lams, nus, dx, dlam, dnu):
    """(dQ, dp, dA, db, dG, dh) from the differentials (reference :527-562)."""
    from .solve_qp_eqcon_torch import _outer_grads
    dl_dQ, dl_dA = _outer_grads(dx, x, dnu, nus if dnu is not None else None)
    dl_db = -dnu if dnu is not None else None
    dl_dG = dl_dh = None
    if dlam is not None:
        dl_dG = lams * (dlam * x.transpose(1, 2)) + torch.matmul(lams, dx.transpose(1, 2))
        dl_dh = -lams * dlam
    return (dl_dQ, dx, dl_dA, dl_db, dl_dG, dl_dh)


def torch_qp_int_grads_admm(x, lams, nus, dx, dlam, dnu, any_lb, any_ub):
    """7-tuple (dQ, dp, dA, db, dlb, dub, None) (reference :565-584, including its choice of the
    first n rows of dl_dh in the ub-only case)."""
    n = x.shape[1]
    dl_dQ, dl_dp, dl_dA, dl_db, _, dl_dh = torch_qp_int_grads(x=x, lams=lams, nus=nus, dx=dx, dlam=dlam, dnu=dnu)
    dl_dlb = dl_dub = None
    if any_lb and any_ub:
        dl_dlb, dl_dub = -dl_dh[:, :n, :], dl_dh[:, n:(2 * n), :]
    elif any_lb:
        dl_dlb = -dl_dh[:, :n, :]
    elif any_ub:
        dl_dub = dl_dh[:, :n, :]
    return (dl_dQ, dl_dp, dl_dA, dl_db, dl_dlb, dl_dub, None)


# ---------------------------------------------------------------------------
# internals
# ---------------------------------------------------------------------------
def _finite_bounds(lb, ub):
    """(any_lb, any_ub): global over the whole batch, as in the reference (:33-34, :129-130) -- two reductions and a
    host round trip.  Only the cold paths use it (``unroll``, ``backward='kkt'``, the FIRST un-synchronised call with a
    given control -- a module, or a plain dict with settings not seen before); the layer itself learns the answer from the
    device (see _forward_solve)."""
    flags = torch.stack((torch.max(lb) > -_INF, torch.min(ub) < _INF)).tolist()
    return bool(flags[0]), bool(flags[1])


# "Does the batch hold any finite bound?" selects the schedule (rho = 0: one KKT solve, :157-158) and is a host decision
# in the reference.  Here the setup kernel answers it from the data of EVERY call (status words 12/13); the host only
# assumes an answer when it enqueues -- what the last solve of the same LAYER saw -- and compares afterwards: a call
# that waits for the GPU repeats itself on the other schedule, an un-synchronised one reports the mismatch late (its
# outputs are then not the reference's and must not be used).  The remembered answer lives in a side table, never in
# the caller's control dict (box_qp_control(**control), dict comparisons and serialisation see the dict unchanged):
# keyed weakly by the nn.Module that made the call (SolveBoxQP, ShardedBoxQP), or -- SolveBoxQPLayer.apply /
# torch_solve_box_qp called directly: plain dicts cannot be weakly referenced -- by the dict's id in a bounded table
# (a stale entry after an id is re-used costs one repeated solve, nothing else).
_tls = threading.local()
_seen_by_module = weakref.WeakKeyDictionary()
_seen_by_dict_id = {}


def _same_settings(snap, control):
    """Does `control` hold what `snap` (a shallow copy taken when the answer was remembered) held?  Tensor values count
    only by identity."""
    if len(snap) != len(control):
        return False
    for k, v in snap.items():
        if k not in control:
            return False
        w = control[k]
        if v is w:
            continue
        if torch.is_tensor(v) or torch.is_tensor(w):
            return False
        try:
            if not (v == w):
                return False
        except Exception:
            return False
    return True


def _assume_any_bound(holder, control, sync=True):
    if holder is not None:
        return _seen_by_module.get(holder)
    # CPython hands the address of a short-lived dict (box_qp_control(...) built per call) to the next one: an entry of the
    # id table may belong to another dict.  The entry therefore carries a copy of the settings it was remembered for and
    # only answers for a dict that holds the same (a pipelined call without a module used to look at the bounds
    # on every call instead -- two reductions and a host read that drain the stream).  A dict with EQUAL settings at a
    # recycled address is, for this purpose, the same control: "what the last solve with these settings saw"; a wrong
    # assumption is caught by the device-side flags as before (repeat / late report).
    ent = _seen_by_dict_id.get(id(control))
    if ent is None:
        return None
    snap, any_bound = ent
    return any_bound if _same_settings(snap, control) else None


def _remember_any_bound(holder, control, any_bound):
    if holder is not None:
        _seen_by_module[holder] = bool(any_bound)
        return
    if len(_seen_by_dict_id) > 256:
        _seen_by_dict_id.clear()
    _seen_by_dict_id[id(control)] = (dict(control), bool(any_bound))


def resolve_control(control, n_x):
    """Numbers the loop uses, with the reference's defaults and key names (:133-154)."""
    g = control.get
    check = g('check_solved', max(round((n_x ** 0.5) / 10) * 10, 1))
    ar_iter = max(round(g('adaptive_rho_iter', 100) / check) * check, 1)
    return dict(
        max_iters=g('max_iters', 10_000),
        eps_abs=max(g('eps_abs', 1e-3), 1e-12),
        eps_rel=max(g('eps_rel', 1e-3), 1e-12),
        check_solved=check,
        rho=g('rho', None), rho_min=g('rho_min', 1e-6), rho_max=g('rho_max', 1e6),
        adaptive_rho=bool(g('adaptive_rho', False)),
        adaptive_rho_tol=g('adaptive_rho_tol', 5),
        adaptive_rho_iter=ar_iter,
        adaptive_rho_max_iter=g('adaptive_max_iter', 1000),
        adaptive_rho_threshold=g('adaptive_rho_threshold', 1e-5),
        scale=bool(g('scale', False)),
        beta=g('beta'),
        verbose=g('verbose', False),
        launch_mode=g('launch_mode', 0),         # extension: 0 auto, 1 segmented, 2 persistent
        linsolve=g('linsolve', 'auto'),          # extension: 'auto' | 'lu' (the reference's cached LU) | 'spd'
    )


def _rho_argument(rho, B, like):
    """-> (mode, scalar value, device tensor or None): 0 auto, 1 scalar, 2 one value per problem"""
    if rho is None:
        return 0, 0.0, None
    if not torch.is_tensor(rho):
        return 1, float(rho), None
    if rho.numel() == 1:
        return 1, float(rho), None
    if rho.numel() != B:
        _bad("a rho tensor must hold one value per problem, e.g. shape (B,1,1)")
    return 2, 0.0, rho.detach().to(device=like.device, dtype=like.dtype).reshape(B).contiguous()


_LINSOLVE = {'auto': 0, 'lu': 1, 'spd': 2, 0: 0, 1: 1, 2: 2}
_ws_bytes = {}                      # (dtype, B, n, m) -> lqp_boxqp_forward_workspace_bytes (a library call per solve otherwise)
_ctl_cache = threading.local()      # .d: (control items, n, any_bound, sync, dtype) -> (resolved dict, rho, lqp_boxqp_ctrl)


def _bad(msg):
    raise ValueError("lqp_py_amd: " + msg)


def _beta_argument(beta, B, like):
    """-> (mode, scalar value, device tensor or None): 0 quantile rule, 1 scalar, 2 one value per problem
    (the reference broadcasts a (B,1) tensor against D (B,n), :171-175)"""
    if beta is None:
        return 0, 0.0, None
    if not torch.is_tensor(beta):
        return 1, float(beta), None
    if beta.numel() == 1:
        return 1, float(beta), None
    if beta.numel() != B:
        _bad("a beta tensor must hold one value per problem, e.g. shape (B,1)")
    return 2, 0.0, beta.detach().to(device=like.device, dtype=like.dtype).reshape(B).contiguous()


def _forward_solve(Q, p, A, b, lb, ub, control, bounds=None, sync=True, residuals=False, check_hook=None, mutate=False,
                   holder=None, private_ws=None, keep_factor=False, while_running=None, one_call=False):
    """bounds: (any_lb, any_ub) when the caller KNOWS them (a shard of a larger batch with host-side flags); None: found
    on the device.  mutate: apply the reference layer's dict side effect control['rho'] = 0 (:37-38).  holder: the
    nn.Module on whose behalf the call is made (keys what is remembered between calls, see _assume_any_bound)."""
    # (what stands between the call and its first kernel launch is the forward time of a step that starts from an idle
    #  queue -- experiment_1's protocol: errors of EARLIER calls are polled after this one is enqueued, the outputs are
    #  one allocation)
    _lib.require_gpu(Q, p, A, b, lb, ub)
    lib = _lib.load()
    B, n = Q.shape[0], p.shape[1]
    m = get_ncon(A, dim=1)
    dt = _lib.dtype_code(p)
    dev = p.device
    if check_hook is not None:
        sync = True                                   # the strict global stop is host-driven (one launch per check)
    known = bounds is not None
    owner = control.get('_owner') or control          # (lqp_py_amd.dist hands the layer a copy of the caller's dict)
    flags_dev = control.get('_bound_flags_dev')       # (lqp_py_amd.dist: device-side flags of the whole batch)
    if known:
        any_bound = bool(bounds[0] or bounds[1])
    else:
        any_bound = _assume_any_bound(holder, owner, sync)
        if any_bound is None:                         # first solve of this layer
            if sync:
                any_bound = True
            elif flags_dev is not None:
                # a shard: the flags of the WHOLE batch decide (one host read, the same answer on every rank -- a shard
                # without any finite bound inside a batch that has some must enqueue the schedule its peers enqueue)
                any_bound = bool(flags_dev.max().item())
            else:
                any_bound = any(_finite_bounds(lb, ub))
    # the control struct of the last call with the same settings is reused (resolving ~25 keys and filling the struct
    # costs ~15 us, in front of the first kernel launch); anything that holds a tensor (per-problem rho / beta) is
    # resolved afresh
    ckey, cached = None, None
    cache = _ctl_cache.__dict__.setdefault('d', {})   # (per thread: the struct is written to below)
    if check_hook is None:
        # (the dict of the last call, unchanged -- one C-level comparison of two dicts -- skips building and hashing the key)
        last = _ctl_cache.__dict__.get('last')
        try:
            same = last is not None and last[0] is control and last[2] == (n, any_bound, sync, p.dtype) and last[1] == control
        except Exception:                             # (a value that does not compare to a truth value: a tensor)
            same = False
        if same:
            ckey, cached = last[3], last[4]
        else:
            try:
                ckey = (tuple(kv for kv in control.items() if kv[0][0] != '_'), n, bool(any_bound), bool(sync), p.dtype)
                cached = cache.get(ckey)
            except Exception:                         # an unhashable value, a key that is not a string: no caching
                ckey = None
            if cached is not None:
                _ctl_cache.last = (control, dict(control), (n, any_bound, sync, p.dtype), ckey, cached)
    if cached is not None:
        r, rho, ctl = cached
        rho_mode, rho_value, rho_tensor, beta_tensor = ctl.rho_mode, ctl.rho_value, None, None
        ctl.bound_flags_in = None if flags_dev is None else flags_dev.data_ptr()
        ctl.host_report = None
    else:
        r = resolve_control(control, n)
        rho = r['rho']
        if not any_bound:
            rho = 0                                     # one iteration solves it (:157-158)
        rho_mode, rho_value, rho_tensor = _rho_argument(rho, B, p)
        beta_mode, beta_value, beta_tensor = _beta_argument(r['beta'], B, p)
        if r['linsolve'] not in _LINSOLVE:
            _bad("control['linsolve'] must be 'auto', 'lu' or 'spd'")
        ctl = _lib.BoxQPCtrl(
            linsolve=_LINSOLVE[r['linsolve']],
            max_iters=int(r['max_iters']), check_solved=int(r['check_solved']),
            adaptive_rho=int(r['adaptive_rho']), adaptive_rho_iter=int(r['adaptive_rho_iter']),
            adaptive_rho_max_iter=int(r['adaptive_rho_max_iter']), scale=int(r['scale']),
            any_lb=int(any_bound), any_ub=int(any_bound), rho_mode=rho_mode,
            beta_mode=beta_mode, launch_mode=int(r['launch_mode']),
            # 1: pipelined (nothing waits); 2: split synchronous call -- enqueue, build the output views while the GPU runs,
            # then lqp_boxqp_forward_finish polls the report (include/lqp_amd.h)
            reserved=(2 if (check_hook is None and _SYNC_SPLIT) else 0) if sync else 1,
            eps_abs=float(r['eps_abs']), eps_rel=float(r['eps_rel']), rho_value=rho_value,
            rho_min=float(r['rho_min']), rho_max=float(r['rho_max']),
            adaptive_rho_tol=float(r['adaptive_rho_tol']),
            adaptive_rho_threshold=float(r['adaptive_rho_threshold']),
            beta_value=beta_value, beta_in=None if beta_tensor is None else beta_tensor.data_ptr(),
            bound_flags_in=None if flags_dev is None else flags_dev.data_ptr())
        if ckey is not None and rho_tensor is None and beta_tensor is None and not any(torch.is_tensor(v) for v in control.values()):
            if len(cache) > 64:
                cache.clear()
            cache[ckey] = (r, rho, ctl)
    # 1: pipelined (nothing waits); 2: split synchronous call -- enqueue, build the output views while the GPU runs, then
    # lqp_boxqp_forward_finish polls the report (include/lqp_amd.h); 0: the library waits itself (and repeats by itself on the
    # schedule that needs no partner when a shared kernel timed out: one_call, below)
    ctl.reserved = (2 if (check_hook is None and _SYNC_SPLIT and not one_call) else 0) if sync else 1
    Qc, pc, Ac, bc, lbc, ubc = (_lib.norm(t, p.dtype) for t in (Q, p, A, b, lb, ub))
    hook_c = None
    # status / info words arrive in pinned host memory, stored there by the forward's last kernel (include/lqp_amd.h:
    # host_report): an un-synchronised call reads them later, a synchronous one right after its wait -- no copies either way
    report = _lib.host_report(_lib.ST_WORDS + 2 * B)
    ctl.host_report = report.data_ptr()
    stats = _lib.BoxQPStats()

    # x is its own allocation (a caller that collects solutions across batches keeps n floats per problem, not the
    # iterates and duals); z, u, lams, nus, rho share one.  Only ADDRESSES are needed in front of the launches -- the views
    # are made after the library call has enqueued them (everything here is host time the GPU idles through in
    # experiment_1's protocol)
    n4 = (n + 3) // 4 * 4                  # (every output starts 16-byte aligned)
    m4 = (m + 3) // 4 * 4
    x = torch.empty((B, n, 1), dtype=p.dtype, device=dev)
    outbuf = torch.empty((B * (4 * n4 + m4 + 4),), dtype=p.dtype, device=dev)
    es = outbuf.element_size()
    base = outbuf.data_ptr()
    o_x, o_z, o_u, o_l = x.data_ptr(), base, base + B * n4 * es, base + 2 * B * n4 * es
    o_nu, o_rho = base + 4 * B * n4 * es, base + (4 * B * n4 + B * m4) * es
    nbytes = _ws_bytes.get((dt, B, n, m))
    if nbytes is None:
        nbytes = _ws_bytes[(dt, B, n, m)] = lib.lqp_boxqp_forward_workspace_bytes(dt, B, n, m)
    stream = _lib.current_stream_handle(dev)
    # (private_ws / keep_factor: the unroll mode keeps the solve's workspace -- factor included -- for its backward)
    ws = private_ws if private_ws is not None else _lib.workspace(dev, nbytes, "fwd", stream)
    ctl.reserved2 = (1 if keep_factor else 0) | (2 if one_call else 0) | (4 if r['verbose'] else 0)      # (bit 1: nothing shared between workgroups, see one_call; bit 2: keep the check trace)
    if check_hook is not None:
        # check_hook(counters) all-reduces (SUM) the four uint32 words of a check -- {not optimal, arrivals, wants rho,
        # ratio trigger} -- in place; it gets a tensor VIEW of the workspace at the device address the library names.
        # check_index -1: the same reduction over the failure vote of a factorisation (include/lqp_amd.h)
        hook_error = []

        def _c_hook(_user, _stream, counters_ptr, check_index):
            try:
                off = int(counters_ptr) - ws.data_ptr()
                check_hook(ws[off:off + 16].view(torch.int32), int(check_index))
                return 0
            except Exception as exc:                  # never let an exception cross the C frame
                hook_error.append(exc)
                return 1
        hook_c = _lib.CHECK_HOOK(_c_hook)
        ctl.check_hook = hook_c
    with _lib.on_device(dev):
        st = lib.lqp_boxqp_forward(ctypes.c_void_p(stream), dt, B, n, m,
                                   _lib.ptr(Qc), _lib.ptr(pc), _lib.ptr(Ac), _lib.ptr(bc), _lib.ptr(lbc), _lib.ptr(ubc),
                                   ctypes.byref(ctl), _lib.ptr(rho_tensor),
                                   o_x, o_z, o_u, o_l, o_nu if m > 0 else None, o_rho,
                                   ctypes.byref(stats), _lib.ptr(ws), ws.numel())
    split_wait = st == 0 and stats.mode_used == 4
    parts = outbuf.split_with_sizes((B * n4, B * n4, 2 * B * n4, B * m4, 4 * B))
    if n4 == n:
        z, u, lams = parts[0].view(B, n, 1), parts[1].view(B, n, 1), parts[2].view(B, 2 * n, 1)
    else:
        z, u, lams = (parts[k][:B * n * (2 if k == 2 else 1)].view(B, n * (2 if k == 2 else 1), 1) for k in range(3))
    nus = parts[3][:B * m].view(B, m, 1) if m > 0 else None
    rho_out = parts[4][:B]
    if split_wait:
        if while_running is not None:
            while_running(dict(x=x, u=u, lams=lams, nus=nus, rho_out=rho_out.view(B, 1, 1)), int(stats.linsolve_used))
        # (the views above were made while the GPU ran; now the report: polled in pinned memory, no stream wait)
        st = lib.lqp_boxqp_forward_finish(ctypes.c_void_p(stream), B, ctl.max_iters, ctl.check_solved,
                                          ctypes.c_void_p(report.data_ptr()), ctypes.byref(stats))
        if st == 5 and not one_call:
            # a kernel that shares its problem with a partner workgroup gave up waiting for it (bounded spins; something else
            # held the CUs): the one-call form of the synchronous forward degrades by itself -- one workgroup per matrix, the
            # loop that needs nobody -- instead of failing
            _lib._pinned_free.setdefault(report.numel(), []).append(report)
            if while_running is not None:
                while_running(None, 0)
            return _forward_solve(Q, p, A, b, lb, ub, control, bounds=bounds, sync=sync, residuals=residuals, check_hook=check_hook,
                                  mutate=mutate, holder=holder, private_ws=private_ws, keep_factor=keep_factor,
                                  while_running=while_running, one_call=True)
        if st == 7:
            # the matrix left the symmetric x-update (not symmetric, or Qs + rho I not positive definite in f32): the
            # reference's algorithm -- the pivoted LU -- takes the solve, as a one-call synchronous forward does by itself
            _lib._pinned_free.setdefault(report.numel(), []).append(report)
            return _forward_solve(Q, p, A, b, lb, ub, dict(control, linsolve='lu', _owner=owner), bounds=bounds, sync=sync,
                                  residuals=residuals, check_hook=check_hook, mutate=mutate, holder=holder, private_ws=private_ws,
                                  keep_factor=keep_factor, while_running=while_running)
    if st != 0 or (check_hook is not None and hook_error):
        # kernels of the failed call may still be in flight, and they write their report into `report`: wait before the
        # pinned buffer goes back to the pool (cold path)
        torch.cuda.current_stream(dev).synchronize()
        _lib._pinned_free.setdefault(report.numel(), []).append(report)
    if check_hook is not None and hook_error:
        raise hook_error[0]
    if st == 3:
        if not known and not any_bound:
            # the rho = 0 one-shot was enqueued on the ASSUMPTION that this batch, like the layer's last one, holds no
            # finite bound.  If it does, the reference would have run its ADMM loop with rho > 0 (:157-158 do not apply),
            # whose KKT matrix can be regular where [[Q, A^T], [A, 0]] is not (Q = 0: an LP, zero rows): look (cold path)
            has = _finite_bounds(lb, ub)
            if has[0] or has[1]:
                _remember_any_bound(holder, owner, True)
                return _forward_solve(Q, p, A, b, lb, ub, control, bounds=has, sync=sync, residuals=residuals,
                                      check_hook=check_hook, mutate=mutate, holder=holder, private_ws=private_ws,
                                      keep_factor=keep_factor, while_running=while_running)
        # the reference's torch.linalg.lu_factor raises on an exactly singular KKT matrix (:215)
        raise RuntimeError(f"lqp_py_amd.torch_solve_box_qp: LU factorisation hit an exactly zero pivot "
                           f"(batch index {stats.fail_index}); the KKT matrix is singular")
    _lib.check(st, "torch_solve_box_qp")
    try:
        _lib.poll_errors()            # (an error of an EARLIER un-synchronised call surfaces here: this call is enqueued,
    finally:                          #  its own report is queued behind the poll -- it is never raised at its own call)
        if stats.mode_used == 3:      # nothing was waited for: the report is read once the stream has passed the call
            _lib.defer_check("SolveBoxQP.forward", dev, report, B, True,
                             bounds_check=None if known else (
                                 any_bound, owner, mutate, lambda _c, seen: _remember_any_bound(holder, owner, seen)))
        elif report is not None:      # (the library waited after all: nothing left to report late)
            _lib._pinned_free.setdefault(report.numel(), []).append(report)
            report = None
    if stats.mode_used != 3 and not known and stats.any_lb >= 0:
        # the device looked at the bounds: did the schedule we enqueued fit them?
        seen = bool(stats.any_lb or stats.any_ub)
        _remember_any_bound(holder, owner, seen)
        if seen != any_bound:
            return _forward_solve(Q, p, A, b, lb, ub, control, bounds=(bool(stats.any_lb), bool(stats.any_ub)), sync=sync,
                                  residuals=residuals, check_hook=check_hook, mutate=mutate, holder=holder, private_ws=private_ws,
                                  keep_factor=keep_factor, while_running=while_running)
    if mutate and not any_bound:
        control['rho'] = owner['rho'] = 0          # written into the CALLER's dict, as the reference does (:37-38)
    if r['verbose']:
        # the reference prints these three lines at every check (:289-294); here the loop runs on the device without the host, so the
        # trace it kept -- the largest primal and dual error of the batch per check -- is printed once the solve is over
        n_chk = int(stats.n_check) if stats.n_check >= 0 else int(r['max_iters'] - 1) // int(r['check_solved']) + 1
        tr = torch.empty((2 * max(min(n_chk, 2048), 1),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.lqp_boxqp_check_trace(_lib.stream_ptr(dev), dt, B, n, m, _lib.ptr(ws), ws.numel(), min(n_chk, 2048), _lib.ptr(tr)),
                       "check_trace")
        trh = tr.cpu()
        it_last = int(stats.iters) if stats.iters >= 0 else None
        for c in range(min(n_chk, 2048)):
            it_c = c * int(r['check_solved'])
            if it_last is not None and it_c > it_last:
                break
            if stats.iters < 0 and float(trh[2 * c]) == 0.0 and float(trh[2 * c + 1]) == 0.0 and c > 0:
                break                                      # (a pipelined call: the checks that never ran left no mark)
            print(f'iteration = {it_c}')
            print(f'|| primal_error|| = {float(trh[2 * c]):.10f}')
            print(f'|| dual_error|| = {float(trh[2 * c + 1]):.10f}')

    # type of the returned rho follows the reference: a python number stays one unless
    # adaptive rho rewrote it (:248-250); None becomes a (B,1,1) tensor (:200-203)
    if rho_mode == 1 and not stats.rho_updated and not (stats.mode_used == 3 and r['adaptive_rho']):
        rho_ret = rho
    else:
        rho_ret = rho_out.view(B, 1, 1)     # (un-synchronised calls cannot know whether rho was adapted: tensor)
    sol = {"x": x, "z": z, "u": u, "lams": lams, "nus": nus, "rho": rho_ret, "iter": int(stats.iters)}
    if residuals:                        # errors of the last check (the NumPy twin's extra outputs)
        pri = torch.empty((B,), dtype=p.dtype, device=dev)
        dua = torch.empty((B,), dtype=p.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.lqp_boxqp_last_residuals(_lib.stream_ptr(dev), dt, B, n, m, _lib.ptr(ws), ws.numel(),
                                                    _lib.ptr(pri), _lib.ptr(dua)), "last_residuals")
        sol["primal_error"], sol["dual_error"] = pri, dua
    sol["_stats"] = {k: getattr(stats, k) for k, _ in stats._fields_}
    _last_forward[(dev.index, stream)] = (
        ws, dt, B, n, m, dict(sol["_stats"]), int(r['check_solved']), int(r['max_iters']))
    return sol


_last_forward = {}


def last_forward_status(device):
    """Bookkeeping of the most recent forward solve on `device`, on the CURRENT stream (workspaces are per stream): {"iters", "n_check", "n_factor", "mode_used",
    "linsolve_used", "factor_launches", "loop_workgroups_per_qp"}.  A call that did not wait for the GPU
    (control['sync'] = False) could not report its iteration count; it is read here from the device-side status
    block of that call's workspace (this waits for the device).  Valid until the next forward on the same stream."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    ws, dt, B, n, m, st, check, max_iters = _last_forward[(device.index, _lib.current_stream_handle(device))]
    if st["mode_used"] == 3:
        lib = _lib.load()
        so, sb, io, ib = (ctypes.c_size_t() for _ in range(4))
        _lib.check(lib.lqp_boxqp_forward_layout(dt, B, n, m, ctypes.byref(so), ctypes.byref(sb), ctypes.byref(io),
                                                ctypes.byref(ib)), "forward_layout")
        torch.cuda.synchronize(device)
        status = ws[so.value:so.value + sb.value].view(torch.int32).cpu().tolist()
        iters = status[1] if status[0] else max_iters - 1          # [0] done, [1] final iteration, [3] refactorisations
        st = dict(st, iters=iters, n_check=iters // check + 1, n_factor=1 + status[3], n_solve=iters + 1,
                  any_lb=status[12], any_ub=status[13])
    st["loop_workgroups_per_qp"] = st.pop("loop_workgroups")
    return st


class _Prepared(dict):
    """What _fp_backward_prepare hands to _fp_backward_run.  A forward that is never followed by its backward (validation with
    grad enabled, an exception, a retry on the other schedule) drops it: the pinned report buffer then goes back to the pool
    instead of leaking one buffer per call.  The prefactor phase reports into that buffer (include/lqp_amd.h,
    lqp_boxqp_backward_fp_prefactor): it is re-used only once every word of it has arrived (_lib.pinned_release)."""

    def __del__(self):
        try:
            rep = self.get('report')
            if rep is not None and not self.get('ran'):
                if self.get('pref_reported'):
                    _lib.pinned_release(rep)
                else:
                    _lib._pinned_free.setdefault(rep.numel(), []).append(rep)
        except Exception:
            pass


# dQ of a prepared backward is allocated up front (in the window where the host only waits for the forward) up to this many
# bytes; above, at `backward` itself: a forward without a backward would hold it for nothing -- 1 GB at B = 1024, n = 500
_PREPARE_DQ_MAX_BYTES = 256 << 20


def _fp_backward_prepare(x, u, lams, nus, Q, A, lb, ub, rho, want, sync=True, linsolve=1, prefactor=False):
    """Everything of the fixed-point backward that does not need the cotangent: output tensors, workspace, report buffer,
    the argument list of lqp_boxqp_backward_fp.  A synchronous layer call runs this WHILE its forward is on the GPU (the host
    would only wait), so that `backward` is one library call behind the autograd engine's thread hop."""
    _lib.require_gpu(x, u, lams, nus, Q, A, lb, ub)
    lib = _lib.load()
    B, n = Q.shape[0], Q.shape[1]
    m = get_ncon(A, dim=1)
    dt = _lib.dtype_code(x)
    dev, dty = x.device, x.dtype
    if rho is None:
        rho = 1.0                                          # (:356-357)
    rho_mode, rho_value, rho_tensor = _rho_argument(rho, B, x)
    xc, uc, lc, nc, Qc, Ac, lbc, ubc = (_lib.norm(t, dty) for t in (x, u, lams, nus, Q, A, lb, ub))
    mk = lambda on, shape: torch.empty(shape, dtype=dty, device=dev) if on else None
    late_dQ = bool(want['dQ']) and prefactor and B * n * n * x.element_size() > _PREPARE_DQ_MAX_BYTES
    dQ = None if late_dQ else mk(want['dQ'], (B, n, n))
    dp = mk(want['dp'], (B, n, 1))
    dA = mk(want['dA'] and m > 0, (B, m, n))
    db = mk(want['db'] and m > 0, (B, m, 1))
    dlb = mk(want['dlb'], (B, n, 1))
    dub = mk(want['dub'], (B, n, 1))
    nbytes = lib.lqp_boxqp_backward_fp_workspace_bytes(dt, B, n, m)
    stream = _lib.current_stream_handle(dev)
    ws = _lib.workspace(dev, nbytes, "bwd", stream)
    fail = ctypes.c_int32(-1)
    report = _lib.host_report(B)                        # (the info words go straight into pinned host memory)
    head = (ctypes.c_void_p(stream), dt, B, n, m)
    tail = (_lib.ptr(xc), _lib.ptr(uc), _lib.ptr(lc), _lib.ptr(nc), _lib.ptr(Qc), _lib.ptr(Ac), _lib.ptr(lbc), _lib.ptr(ubc),
            rho_mode, rho_value, _lib.ptr(rho_tensor),
            _lib.ptr(dQ), _lib.ptr(dp), _lib.ptr(dA), _lib.ptr(db), _lib.ptr(dlb), _lib.ptr(dub),
            ctypes.byref(fail) if sync else None, _lib.ptr(ws), ws.numel())
    rep = None if report is None else ctypes.c_void_p(report.data_ptr())
    keep = (xc, uc, lc, nc, Qc, Ac, lbc, ubc, rho_tensor, ws)          # (the pointers above point into these)
    pref, pref_reported = None, False
    if prefactor:
        # (reads x, u behind the forward's kernels in stream order -- the Cholesky form with linsolve 2, else the reduced system's
        #  pivoted LU and its packed factor; LQP_ERR_UNSUPPORTED = this form has no phases: nothing was enqueued.  The
        #  factorisation's info words go into the report buffer of the backward call: that call then waits for them only -- it
        #  returns while its solves and the gradient epilogue run)
        with _lib.on_device(dev):
            st = lib.lqp_boxqp_backward_fp_prefactor(*head, _lib.ptr(xc), _lib.ptr(uc), _lib.ptr(Qc), _lib.ptr(Ac), _lib.ptr(lbc),
                                                     _lib.ptr(ubc), _lib.ptr(ws), ws.numel(), int(linsolve), rep if sync else None)
        if st == 0:
            pref = _lib.workspace_uses(dev, "bwd", stream)      # (still ours at `backward` if nobody asked for the buffer since)
            pref_reported = bool(sync and rep is not None)
        elif st != 6:
            _lib.check(st, "torch_solve_box_qp_grad (prefactor)")
    return _Prepared(lib=lib, head=head, tail=tail, keep=keep, grads=(dQ, dp, dA, db, dlb, dub, None), fail=fail, report=report,
                     dev=dev, dty=dty, B=B, sync=sync, linsolve=int(linsolve), rep=rep, pref=pref, stream=stream,
                     late_dQ=(B, n) if late_dQ else None, pref_reported=pref_reported)


def _fp_backward_run(prep, dl_dz):
    _lib.require_gpu(dl_dz)
    gc = _lib.norm(dl_dz, prep['dty'])
    dev, report, B, sync = prep['dev'], prep['report'], prep['B'], prep['sync']
    linsolve = prep['linsolve']
    if prep['pref'] is not None and prep['pref'] == _lib.workspace_uses(dev, "bwd", prep['stream']):
        linsolve |= _BWD_PREFACTORED                       # the factorisation made behind the forward is still in the workspace
        if prep.get('pref_reported'):
            linsolve |= _BWD_REPORTED                      # ... and its info words are in (or on their way into) the report buffer
    # Every run writes the (shared, cached) backward workspace: a full run rebuilds free set, factor and info words in it, the
    # solve phase its right-hand sides.  Whoever prefactored into the same buffer earlier (forward A, forward B, backward A,
    # backward B: B's factor is overwritten by A's full run) must see the count move and run in full too.
    _lib.workspace_touch(dev, "bwd", prep['stream'])
    tail = prep['tail']
    if prep.get('late_dQ'):
        Bq, nq = prep['late_dQ']
        dQ = torch.empty((Bq, nq, nq), dtype=prep['dty'], device=dev)
        prep['grads'] = (dQ,) + tuple(prep['grads'][1:])
        tail = tail[:11] + (_lib.ptr(dQ),) + tail[12:]                 # (the slot of dQ in lqp_boxqp_backward_fp's argument list)
    prep['ran'] = True                                                 # (the report buffer is this run's to hand back from here on)
    try:
        with _lib.on_device(dev):
            st = prep['lib'].lqp_boxqp_backward_fp(*prep['head'], _lib.ptr(gc), *tail, linsolve, prep['rep'])
        if st == 3:
            raise RuntimeError(f"lqp_py_amd.torch_solve_box_qp_grad: the input matrix is singular (batch index {prep['fail'].value})")
        _lib.check(st, "torch_solve_box_qp_grad")
    except Exception:
        _lib.pinned_release(report)     # (a failed call: its pinned report goes back -- quarantined while a queued kernel may still write it)
        raise
    try:
        _lib.poll_errors()              # (errors of earlier calls; this call's own report is queued behind the poll)
    finally:
        if not sync:
            _lib.defer_check("SolveBoxQP.backward", dev, report, B, False)
        else:
            _lib._pinned_free.setdefault(report.numel(), []).append(report)
    return prep['grads']


def _fp_backward(dl_dz, x, u, lams, nus, Q, A, lb, ub, rho, want, sync=True, linsolve=1):
    return _fp_backward_run(_fp_backward_prepare(x, u, lams, nus, Q, A, lb, ub, rho, want, sync=sync, linsolve=linsolve), dl_dz)
