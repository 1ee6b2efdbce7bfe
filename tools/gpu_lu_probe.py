#!/usr/bin/env python3
"""The benched step with control['linsolve'] = 'lu' (the reference's algorithm: pivoted LU + cached triangular solves; B = 128, n = 500,
m = 1): per-kernel-class times.  Usage: gpu_lu_probe.py [B] [n]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False, linsolve='lu'))
cot = torch.ones_like(inp[1])
def step():
    Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
    layer(Q, p, *inp[2:]).backward(cot)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); L.synchronize()
dt = (time.perf_counter() - t0) / 10
st = last_forward_status(dev)
_lib.profile(enable=True, reset=True)
for _ in range(5): step()
torch.cuda.synchronize()
pr = {k: (round(v[0] / 5, 4), v[1] // 5) for k, v in _lib.profile().items() if v[1]}
_lib.profile(enable=False)
print(f"B={B} n={n}: {dt*1e3:.3f} ms/step  iters {st['iters']} linsolve {st['linsolve_used']} mode {st['mode_used']} launches {st.get('n_launch')}  {pr}")
