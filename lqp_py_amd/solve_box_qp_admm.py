"""NumPy twin of the box-QP ADMM solver: ``solve_box_qp`` / ``BoxQP`` of
lqp_py/solve_box_qp_admm.py:8-91 for ONE problem given as NumPy arrays, computed by the same HIP kernels as the
torch layer (a batch of one).  The reference's NumPy code is the torch algorithm line by line -- its scaling takes
row maxima where the torch code takes column maxima (solve_box_qp_admm.py:127 vs solve_box_qp_admm_torch.py:163),
the same thing for the symmetric Q of a QP -- so the adapter only converts types and key names
(``lam``/``nu`` instead of ``lams``/``nus``, plus the primal / dual error of the last check).
"""
import numpy as np
import torch

from .solve_box_qp_admm_torch import _forward_solve
from .solve_qp_uncon import solve_qp_uncon
from .utils import make_matrix


class BoxQP:
    """Stateful holder (lqp_py/solve_box_qp_admm.py:8-43), including its quirk that ``update(lb=..)`` /
    ``update(ub=..)`` RESET the bound to None (:38-41)."""

    def __init__(self, Q, p, A, b, lb, ub, control):
        self.Q, self.p, self.A, self.b, self.lb, self.ub, self.control = Q, p, A, b, lb, ub, control
        self.sol = {}

    def solve(self):
        sol = solve_box_qp(Q=self.Q, p=self.p, A=self.A, b=self.b, lb=self.lb, ub=self.ub, control=self.control)
        self.sol = sol
        return sol.get('x')

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


def _bound_vector(value, n_x, fallback):
    """A bound given as None, a scalar or an array -> float vector of length n_x (the reference broadcasts a short
    bound over all variables, lqp_py/solve_box_qp_admm.py:270-277)."""
    arr = np.asarray(fallback if value is None else value, dtype=float).reshape(-1)
    if arr.size < n_x:
        arr = np.tile(arr, n_x)
    return arr


def solve_box_qp(Q, p, A=None, b=None, lb=-float("inf"), ub=float("inf"), control=None):
    """argmin 0.5 x'Qx + p'x  s.t.  A x = b, lb <= x <= ub for one problem (lqp_py/solve_box_qp_admm.py:45-91).
    Returns {"x","z","u","lam","nu","rho","primal_error","dual_error","iter"} with 1-D arrays, like the reference."""
    Q = make_matrix(Q)
    p = make_matrix(p)[:, 0]
    if A is not None:
        A = make_matrix(A)
    if b is not None:
        b = make_matrix(b)[:, 0]
    n_x = p.shape[0]
    lb = _bound_vector(lb, n_x, -float("inf"))
    ub = _bound_vector(ub, n_x, float("inf"))
    any_eq = A is not None
    any_ineq = (lb.max() > -float("inf")) or (ub.max() < float("inf"))        # (:72-74, ub.max() as there)
    if not any_ineq:
        control['rho'] = 0                                # (:77-78) written into the caller's dict
    if not any_eq and not any_ineq:
        return solve_qp_uncon(Q=Q, p=p)                   # (:81-82)

    dt = torch.float32 if Q.dtype == np.float32 else torch.float64
    dev = torch.device("cuda", torch.cuda.current_device())
    t3 = lambda a, *shape: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev).reshape(*shape)
    m = A.shape[0] if any_eq else 0
    sol = _forward_solve(t3(Q, 1, n_x, n_x), t3(p, 1, n_x, 1), t3(A, 1, m, n_x) if any_eq else None,
                         t3(b, 1, m, 1) if any_eq else None, t3(lb, 1, n_x, 1), t3(ub, 1, n_x, 1), control,
                         sync=True, residuals=True)
    npy = lambda t: t.reshape(-1).cpu().numpy()
    rho = sol["rho"]
    return {"x": npy(sol["x"]), "z": npy(sol["z"]), "u": npy(sol["u"]), "lam": npy(sol["lams"]),
            "nu": npy(sol["nus"]) if any_eq else None,
            "rho": float(rho.reshape(-1)[0]) if torch.is_tensor(rho) else rho,
            "primal_error": float(sol["primal_error"][0]), "dual_error": float(sol["dual_error"][0]),
            "iter": sol["iter"]}
