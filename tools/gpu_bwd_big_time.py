#!/usr/bin/env python3
"""Backward at config-4 size (B=128, n=1000, m=1): Cholesky form against the LU form (device events)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
import lqp_py_amd.solve_box_qp_admm_torch as SB
dev = torch.device("cuda:0")
B, n = 128, 1000
gen = torch.Generator(device=dev).manual_seed(5)
Lm = torch.randn(B, 2 * n, n, device=dev, generator=gen)
Q = Lm.transpose(1, 2) @ Lm / (2 * n); del Lm
p = torch.randn(B, n, 1, device=dev, generator=gen)
A, b = torch.ones(B, 1, n, device=dev), torch.ones(B, 1, 1, device=dev)
lb, ub = -(torch.rand(B, n, 1, device=dev, generator=gen) + 1), torch.rand(B, n, 1, device=dev, generator=gen) + 1
sol = L.torch_solve_box_qp(Q, p, A, b, lb, ub, L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5))
cot = torch.ones(B, n, 1, device=dev)
want = dict(dQ=True, dp=True, dA=True, db=True, dlb=True, dub=True)
for ls, name in ((1, "LU + refinement"), (2, "Cholesky")):
    f = lambda: SB._fp_backward(cot, sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"], want, sync=False, linsolve=ls)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize(); L.synchronize()
    print(f"backward B={B} n={n} m=1, {name}: {e0.elapsed_time(e1) / 5:.3f} ms")
