#!/usr/bin/env python3
"""bench.py's step_linsolve_lu workload (batch=128 dz=500 m=1, control['linsolve']='lu', forward + backward): per-kernel-class times."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
from tools.profile_workload import device_batch
dev = torch.device("cuda:0")
B = int(os.environ.get("LU_B", "128")); n = int(os.environ.get("LU_N", "500"))
data = device_batch(dev, B, n, 0)
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False, linsolve="lu"))
cot = torch.ones_like(data[1])
def step():
    Q = data[0].detach().requires_grad_(True); p = data[1].detach().requires_grad_(True)
    layer(Q, p, *data[2:]).backward(cot)
    return Q.grad, p.grad
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g = step()
torch.cuda.synchronize(); L.synchronize()
dt = (time.perf_counter() - t0) / 10
st = last_forward_status(dev)
_lib.profile(enable=True, reset=True)
for _ in range(5): step()
torch.cuda.synchronize()
pr = {k: round(v[0] / 5, 4) for k, v in _lib.profile().items() if v[1]}
_lib.profile(enable=False)
print(f"lu B={B} n={n}: {dt*1e3:.3f} ms/step = {B/dt:.0f} QPs/s  iters {st['iters']} linsolve {st['linsolve_used']}  {pr}  |dQ| {float(g[0].abs().sum()):.6e} |dp| {float(g[1].abs().sum()):.6e}")
