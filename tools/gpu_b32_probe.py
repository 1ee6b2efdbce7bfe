#!/usr/bin/env python3
"""B = 32, n = 500 forward + backward (pipelined): per-kernel-class times and the schedules the library chose."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
for B in (int(a) for a in (sys.argv[1:] or ["32", "64", "128"])):
    inp = [t.to(dev) for t in create_qp_data(500, B, seed=0)]
    ones = torch.ones(B, 500, 1, device=dev)
    layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
    def step():
        Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
        layer(Q, p, *inp[2:]).backward(ones)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); L.synchronize()
    dt = (time.perf_counter() - t0) / 20
    st = last_forward_status(dev)
    _lib.profile(enable=True, reset=True)
    for _ in range(10): step()
    torch.cuda.synchronize()
    pr = {k: round(v[0] / 10, 4) for k, v in _lib.profile().items() if v[1]}
    _lib.profile(enable=False)
    print(f"B={B}: {dt*1e3:.3f} ms/step  loop_wg {st['loop_workgroups_per_qp']} factor_launches {st['factor_launches']} iters {st['iters']}  {pr}")
