import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lqp_py_amd import SolveBoxQP, box_qp_control
B, n = 128, 500
L = torch.randn(B, 2 * n, n); Q = (L.transpose(1, 2) @ L / (2 * n)).cuda().requires_grad_(True)
p = torch.randn(B, n, 1).cuda().requires_grad_(True)
A, b = torch.ones(B, 1, n).cuda(), torch.ones(B, 1, 1).cuda()
lb, ub = -(torch.rand(B, n, 1) + 1).cuda(), (torch.rand(B, n, 1) + 1).cuda()
layer = SolveBoxQP(control=box_qp_control(eps_abs=1e-5, eps_rel=1e-5))
x = layer(Q, p, A, b, lb, ub)
x.sum().backward()
import lqp_py_amd; lqp_py_amd.synchronize()
print(x.shape, float(x.abs().max()), Q.grad.shape, float(p.grad.abs().max()))
