"""Small helpers shared by the QP layers (reference: lqp_py/utils.py:5-32)."""
import numpy as np
import torch


def make_matrix(x):
    """1-D input -> column matrix (lqp_py/utils.py:5-11)."""
    x = np.asarray(x)
    if len(x.shape) < 2:
        x = x.reshape(-1, 1)
    return x


def get_ncon(x, dim=0):
    """None-safe size of ``x`` along ``dim`` (lqp_py/utils.py:14-20)."""
    return 0 if x is None else x.shape[dim]


def torch_qp_eqcon_mat(Q, A, bottom_right=None):
    """KKT block [[Q, A^T], [A, bottom_right]] (lqp_py/utils.py:23-32); temporaries follow Q's device/dtype."""
    if bottom_right is None:
        bottom_right = torch.zeros((A.shape[0], A.shape[1], A.shape[1]), dtype=Q.dtype, device=Q.device)
    upper = torch.cat((Q, A.transpose(1, 2)), 2)
    lower = torch.cat((A, bottom_right), 2)
    return torch.cat((upper, lower), 1)
