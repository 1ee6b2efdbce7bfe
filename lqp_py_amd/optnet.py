"""``OptNet`` with equality constraints only.

The reference's ``OptNet`` layer (lqp_py/optnet.py:8-54) is an interior-point solver for
``A x = b, G x <= h``; with no inequality rows it is one KKT solve forward (``torch_solve_qp_eqcon``,
optnet.py:49 -> :91) and one more backward (``torch_solve_qp_eqcon_grad``, optnet.py:49-51).  That branch
is what this module provides, on the HIP KKT-solve kernels; the interior-point iteration for ``G, h`` is outside
the scope of this package and raises.
"""
import torch
import torch.nn as nn

from .control import optnet_control
from .solve_qp_eqcon_torch import torch_solve_qp_eqcon, torch_solve_qp_eqcon_grad
from .utils import get_ncon


class OptNet(nn.Module):
    """lqp_py/optnet.py:8-15."""

    def __init__(self, control):
        super().__init__()
        self.control = control

    def forward(self, Q, p, A, b, G, h):
        return OptNetLayer.apply(Q, p, A, b, G, h, self.control)


class OptNetLayer(torch.autograd.Function):
    """lqp_py/optnet.py:18-54, equality-only branch."""

    @staticmethod
    def forward(ctx, Q, p, A, b, G, h, control=None):
        sol = torch_solve_qp_optnet(Q=Q, p=p, A=A, b=b, G=G, h=h, control=control or optnet_control())
        x, nus = sol.get('x'), sol.get('nus')
        ctx.save_for_backward(x, nus, Q, A)
        return x

    @staticmethod
    def backward(ctx, dl_dz):
        x, nus, Q, A = ctx.saved_tensors
        grads = torch_solve_qp_eqcon_grad(dl_dz=dl_dz, x=x, nus=nus, Q=Q, A=A)
        return tuple(grads) + (None, None, None)          # G, h and control (optnet.py:50)


def torch_solve_qp_optnet(Q, p, A, b, G, h, control=None):
    """lqp_py/optnet.py:57-92: without inequality rows this IS ``torch_solve_qp_eqcon``."""
    if get_ncon(G, dim=1) > 0:
        raise NotImplementedError("lqp_py_amd.optnet: only the equality-constrained branch of OptNet (G = None) is "
                                  "provided; the interior-point solver for G x <= h is outside this package")
    return torch_solve_qp_eqcon(Q=Q, p=p, A=A, b=b)
