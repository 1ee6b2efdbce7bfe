"""NumPy twin of the unconstrained solve (lqp_py/solve_qp_uncon.py:4-15), on the GPU KKT-solve kernel."""
import numpy as np
import torch

from .solve_qp_uncon_torch import torch_solve_qp_uncon


def solve_qp_uncon(Q, p):
    """x = solve(Q, -p) for one problem given as NumPy arrays; returns {"x": ndarray} with p's shape."""
    Q = np.asarray(Q)
    p = np.asarray(p)
    dt = torch.float32 if Q.dtype == np.float32 else torch.float64
    dev = torch.device("cuda", torch.cuda.current_device())
    Qt = torch.as_tensor(Q, dtype=dt, device=dev).reshape(1, Q.shape[0], Q.shape[1])
    pt = torch.as_tensor(p, dtype=dt, device=dev).reshape(1, -1, 1)
    x = torch_solve_qp_uncon(Qt, pt)["x"]
    return {"x": x.reshape(p.shape).cpu().numpy()}
