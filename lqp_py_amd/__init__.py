"""MI355X-native batched box-QP ADMM layer (drop-in for the box-QP path of
ipo-lab/lqp_py).  The compute path is the HIP library built from ``csrc/``;
there is no CPU fallback: CPU tensors or a missing library raise."""
from .control import box_qp_control, optnet_control
from .utils import get_ncon, make_matrix, torch_qp_eqcon_mat
from .solve_box_qp_admm_torch import (SolveBoxQP, SolveBoxQPLayer, BoxQPTH, torch_solve_box_qp,
                                      torch_solve_box_qp_grad, torch_solve_box_qp_grad_kkt,
                                      torch_qp_int_grads, torch_qp_int_grads_admm)
from .lu_layer import TorchLU, TorchLULayer
from .solve_qp_eqcon_torch import torch_solve_qp_eqcon, torch_solve_qp_eqcon_grad
from .solve_qp_uncon_torch import torch_solve_qp_uncon, torch_solve_qp_uncon_grad
from .optnet import OptNet, OptNetLayer, torch_solve_qp_optnet
from .solve_box_qp_admm import BoxQP, solve_box_qp
from .solve_qp_uncon import solve_qp_uncon



def release_workspaces(device=None):
    """Free the cached per-(device, stream) scratch buffers (see ``_lib.release_workspaces``)."""
    from . import _lib
    _lib.release_workspaces(device)


def synchronize():
    """Wait for every un-synchronised layer call (``control['sync'] = False``) and raise any error it reported (singular
    KKT matrix, matrix outside the symmetric x-update, barrier timeout, a batch whose bound flags differ from what the call
    was enqueued for).  By default ``SolveBoxQP`` waits for the GPU and raises at the call like the reference."""
    from . import _lib
    _lib.poll_errors(block=True)


__all__ = [
    "synchronize", "release_workspaces", "box_qp_control", "get_ncon", "torch_qp_eqcon_mat", "SolveBoxQP", "SolveBoxQPLayer", "BoxQPTH",
    "torch_solve_box_qp", "torch_solve_box_qp_grad", "torch_solve_box_qp_grad_kkt", "torch_qp_int_grads",
    "torch_qp_int_grads_admm", "TorchLU", "TorchLULayer",
    "torch_solve_qp_eqcon", "torch_solve_qp_eqcon_grad", "torch_solve_qp_uncon", "torch_solve_qp_uncon_grad",
    "optnet_control", "make_matrix", "OptNet", "OptNetLayer", "torch_solve_qp_optnet", "BoxQP", "solve_box_qp",
    "solve_qp_uncon",
]
