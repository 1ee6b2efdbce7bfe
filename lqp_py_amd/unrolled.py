"""``unroll=True``: the ADMM loop as ordinary differentiable torch ops on the GPU, so that
autograd tapes every iteration (reference: lqp_py/solve_box_qp_admm_torch.py:14-15 routes the
module here, :216-219/:255-256/:264-265 swap the plain LU solve for the ``TorchLU`` layer).

Only the linear algebra leaves torch: every x-update is ``TorchLU`` (lqp_py_amd/lu_layer.py), i.e. the
HIP batched LU factor + streaming cached solve with the analytic backward of lu_layer.py:41-58.
The element-wise algebra stays in torch on purpose -- it is what autograd differentiates through.
This is the slow, memory-hungry mode the reference benchmarks as "ADMM Unroll"; the fast path is the
fixed-point backward of ``SolveBoxQPLayer``.
"""
import torch

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


def unrolled_solve_box_qp(Q, p, A, b, lb, ub, r, has_lb, has_ub):
    """``r`` is the resolved control (solve_box_qp_admm_torch.resolve_control). Returns x only,
    as the reference does in unroll mode (:328-329)."""
    dev, dt = p.device, p.dtype
    B, n = Q.shape[0], p.shape[1]
    m = get_ncon(A, dim=1)
    has_box = has_lb or has_ub
    p_inf = _inf_norm(p)
    rho = r['rho']
    if not has_box:
        rho = 0

    D = E = 1.0
    if r['scale']:
        d = torch.sqrt(1 / _floor_nonpositive(torch.linalg.norm(Q, ord=_INF, dim=1)))
        beta = r['beta']
        if beta is None:
            q = torch.quantile(d, q=torch.tensor([0.10, 0.90], dtype=d.dtype, device=dev), dim=1)
            beta = (1 - q[[0]] / q[[1]]).T
        d = (1 - beta) * d + beta * d.mean(dim=1, keepdim=True)
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
        rho = torch.clamp(torch.linalg.matrix_norm(Q, keepdim=True) / n ** 0.5, min=r['rho_min'], max=r['rho_max'])

    eye = torch.eye(n, dtype=dt, device=dev).unsqueeze(0)

    def kkt(rho_now):
        M = Q + rho_now * eye
        if m > 0:
            corner = torch.zeros(B, m, m, dtype=dt, device=dev)
            M = torch.cat((torch.cat((M, A.transpose(1, 2)), 2), torch.cat((A, corner), 2)), 1)
        return M

    M = kkt(rho)
    solver = TorchLU(A=M)                         # HIP factorisation, no_grad inside

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
                solver = TorchLU(A=M)
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
