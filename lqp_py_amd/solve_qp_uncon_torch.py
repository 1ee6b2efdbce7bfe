"""Unconstrained QP  x* = argmin 0.5 x^T Q x + p^T x  =  solve(Q, -p) on MI355X.
Drop-in for ``lqp_py/solve_qp_uncon_torch.py`` (:4-35)."""
from .solve_qp_eqcon_torch import _kkt_solve, _outer_grads


def torch_solve_qp_uncon(Q, p):
    x, _ = _kkt_solve(Q, p, None, None)
    return {'x': x}


def torch_solve_qp_uncon_grad(dl_dz, x, Q):
    dx = torch_solve_qp_uncon(Q=Q, p=dl_dz).get('x')
    dl_dQ, _ = _outer_grads(dx, x, None, None)
    return (dl_dQ, dx)
