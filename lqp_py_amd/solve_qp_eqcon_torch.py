"""Equality-constrained QP = one KKT solve on MI355X.

    [[Q, A^T], [A, 0]] [x; nu] = [-p; b]

Drop-in for ``lqp_py/solve_qp_eqcon_torch.py`` (``torch_solve_qp_eqcon`` :6-34,
``torch_solve_qp_eqcon_grad`` :37-70); the solve is ``lqp_kkt_solve`` (batched
LU + cached triangular solves), the rank-2 gradient epilogue ``lqp_qp_outer_grads``.
"""
import ctypes

import torch

from . import _lib
from .utils import get_ncon


def _kkt_solve(Q, p, A, b):
    _lib.require_gpu(Q, p, A, b)
    lib = _lib.load()
    B, n = Q.shape[0], p.shape[1]
    m = get_ncon(A, dim=1)
    dt = _lib.dtype_code(Q)
    dev, dty = Q.device, Q.dtype
    Qc, pc, Ac, bc = (None if t is None else _lib.c(t).to(dty) for t in (Q, p, A, b))
    x = torch.empty((B, n, 1), dtype=dty, device=dev)
    nus = torch.empty((B, m, 1), dtype=dty, device=dev) if m > 0 else None
    ws = _lib.workspace(dev, lib.lqp_kkt_solve_workspace_bytes(dt, B, n, m), "kkt")
    fail = ctypes.c_int32(-1)
    with torch.cuda.device(dev):
        st = lib.lqp_kkt_solve(_lib.stream_ptr(dev), dt, B, n, m, _lib.ptr(Qc), _lib.ptr(pc), _lib.ptr(Ac),
                               _lib.ptr(bc), _lib.ptr(x), _lib.ptr(nus), ctypes.byref(fail), _lib.ptr(ws), ws.numel())
    if st == 3:
        raise RuntimeError(f"lqp_py_amd: linalg.solve: (Batch element {fail.value}): The solver failed because "
                           "the input matrix is singular.")
    _lib.check(st, "kkt_solve")
    return x, nus


def _outer_grads(dx, x, dnu, nus):
    lib = _lib.load()
    B, n = x.shape[0], x.shape[1]
    m = 0 if dnu is None else dnu.shape[1]
    dty, dev = x.dtype, x.device
    dxc, xc, dnc, nc = (None if t is None else _lib.c(t).to(dty) for t in (dx, x, dnu, nus))
    dQ = torch.empty((B, n, n), dtype=dty, device=dev)
    dA = torch.empty((B, m, n), dtype=dty, device=dev) if m > 0 else None
    with torch.cuda.device(dev):
        _lib.check(lib.lqp_qp_outer_grads(_lib.stream_ptr(dev), _lib.dtype_code(x), B, n, m, _lib.ptr(dxc),
                                          _lib.ptr(xc), _lib.ptr(dnc), _lib.ptr(nc), _lib.ptr(dQ), _lib.ptr(dA)),
                   "qp_outer_grads")
    return dQ, dA


def torch_solve_qp_eqcon(Q, p, A, b):
    if get_ncon(A, dim=1) == 0:
        x, _ = _kkt_solve(Q, p, None, None)          # falls back to the unconstrained solve (:31-32)
        return {'x': x}
    x, nus = _kkt_solve(Q, p, A, b)
    return {"x": x, "nus": nus}


def torch_solve_qp_eqcon_grad(dl_dz, x, nus, Q, A):
    B = Q.shape[0]
    m = get_ncon(A, dim=1)
    zeros = torch.zeros((B, m, 1), dtype=Q.dtype, device=Q.device)
    sol = torch_solve_qp_eqcon(Q=Q, p=dl_dz, A=A, b=zeros)
    dx, dnu = sol.get('x'), sol.get('nus')
    dl_dQ, dl_dA = _outer_grads(dx, x, dnu if m > 0 else None, nus if m > 0 else None)
    dl_db = -dnu if m > 0 else None
    return (dl_dQ, dx, dl_dA, dl_db)
